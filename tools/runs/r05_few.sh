#!/bin/bash
# round 5: the 64 -> 1 output convolution on the matrix core — codec fixtures + oracle comparisons, then config 5 with both forms
mkdir -p gpurun_out/few
timeout 900 python -m pytest tests/test_gpu_codec.py -x -q -m gpu -k "matches_reference or every_seanet_layer or match_the_oracle_directly or roundtrip" > gpurun_out/few/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/few/tests.log
timeout 300 python tools/codec_bench.py 256 30 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/few/codec256_mfma.log
SSRHIP_CONV_FEW_MFMA=0 timeout 300 python tools/codec_bench.py 256 30 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/few/codec256_lds.log
