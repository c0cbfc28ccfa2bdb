#!/bin/bash
# round 5: the multi-stream codec stress test with NaN-poisoned workspaces (a consumer that runs before its producer shows as NaN)
mkdir -p gpurun_out/race
for k in 1 2 3 4 5 6; do
  SSRHIP_POISON_ALLOC=1 timeout 300 python -m pytest tests/test_gpu_codec.py -x -q -m gpu -k "concurrent_streams" 2>&1 | grep -E "passed|failed|AssertionError: round" | cut -c1-500
done 2>&1 | tee gpurun_out/race/race_poison.log
