#!/bin/bash
# round 6, call 27: 36 more stand-alone runs of the multi-stream codec test (12 concurrent rounds each) on the last commit
O=gpurun_out/r6c27; mkdir -p $O
for i in $(seq 1 36); do timeout 120 python -m pytest tests/test_gpu_codec.py -q -p no:cacheprovider -k "concurrent_streams" 2>&1 | grep -E "passed|failed|AssertionError: round" | cut -c1-500; done | tee $O/multistream_x36.log
echo "passed: $(grep -c passed $O/multistream_x36.log) failed: $(grep -c failed $O/multistream_x36.log)"
