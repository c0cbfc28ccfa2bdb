#!/bin/bash
# round 5: after the host-side pair-buffer rule moved behind ssrhip_pair_buffer (ABI 105): kernel test, sampled 830M run against the oracle, A/B
mkdir -p gpurun_out/sanity
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q -m gpu -k "pair_launch or config2" 2>&1 | tail -3 | tee gpurun_out/sanity/tests.log
timeout 600 python tools/decode_ab.py --greedy --steps 300 --reps 2 pair2: nopair:SSRHIP_GEMV_PAIR=0 2>&1 | tail -3 | tee gpurun_out/sanity/decode_ab.log
