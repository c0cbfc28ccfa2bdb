#!/bin/bash
# round 6, call 19: merge pair launch, early units and NOTHING requested between barrier (1b) and the gather
O=gpurun_out/r6c19; mkdir -p $O
SSRHIP_GEMV_PAIR_EARLY=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "pair_launch" 2>&1 | tail -2
timeout 900 python tools/decode_ab.py --steps 300 --warmup 20 --reps 4 early3: early2:SSRHIP_GEMV_PAIR_EARLY=3 noearly:SSRHIP_GEMV_PAIR_EARLY=0 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/decode_ab_pair_early_pf2.log
