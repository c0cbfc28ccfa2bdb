#!/bin/bash
# round 5, last seconds of GPU time: the pair kernels after the "no long wait once a launch has given up" change — bit compare + a short A/B
mkdir -p gpurun_out/sanity
timeout 200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pair_launch" 2>&1 | tail -2 | tee gpurun_out/sanity/tests2.log
timeout 200 python tools/decode_ab.py --greedy --steps 200 --reps 1 pair2: nopair:SSRHIP_GEMV_PAIR=0 2>&1 | tail -2 | tee gpurun_out/sanity/decode_ab2.log
