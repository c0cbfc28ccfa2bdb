#!/bin/bash
# round 5: first runs of the paired GEMV launch — kernel-level bit compare, then the same-process A/B of the decode step (pair on / off)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pair_launch or segu" > gpurun_out/pair_kernel_test.log 2>&1; echo "kernel test rc=$?"
tail -5 gpurun_out/pair_kernel_test.log
timeout 900 python tools/decode_ab.py --greedy --steps 300 --reps 3 pair: nopair:SSRHIP_GEMV_PAIR=0 > gpurun_out/decode_ab_pair.log 2>&1; echo "ab rc=$?"
cat gpurun_out/decode_ab_pair.log | tail -5
