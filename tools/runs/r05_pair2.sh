#!/bin/bash
# round 5: both pair forms — kernel-level bit compare, then the same-process A/B of the decode step (both pairs / FFN2 pairs only / none)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pair_launch" > gpurun_out/pair_kernel_test.log 2>&1; echo "kernel test rc=$?"
tail -15 gpurun_out/pair_kernel_test.log
timeout 900 python tools/decode_ab.py --greedy --steps 300 --reps 3 pair2: pair1:SSRHIP_GEMV_PAIR=1 nopair:SSRHIP_GEMV_PAIR=0 > gpurun_out/decode_ab_pair2.log 2>&1; echo "ab rc=$?"
cat gpurun_out/decode_ab_pair2.log | tail -5
