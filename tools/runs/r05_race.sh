#!/bin/bash
# round 5: the multi-stream codec stress test, repeated under the codec's kernel switches (one full-suite run saw it fail once)
mkdir -p gpurun_out/race
for knob in "X=0" "X=0" "X=0" "SSRHIP_EPILOGUE_TM=0" "SSRHIP_EPILOGUE_TM=0" "SSRHIP_GEMM_SPLIT_DMA=0" "SSRHIP_GEMM_SPLIT_DMA=0" "SSRHIP_RESBLOCK_DMA=0" "SSRHIP_RESBLOCK_DMA=0" "SSRHIP_NO_RECORD_STREAM=1"; do
  echo "== $knob"
  env $knob timeout 300 python -m pytest tests/test_gpu_codec.py -x -q -m gpu -k "concurrent_streams" 2>&1 | grep -E "passed|failed|AssertionError: round" | cut -c1-400
done 2>&1 | tee gpurun_out/race/race.log
