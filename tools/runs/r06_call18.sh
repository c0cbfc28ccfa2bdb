#!/bin/bash
# round 6, call 18: merge pair launch, early units + 3 vs 4 units in flight behind barrier (1b)
O=gpurun_out/r6c18; mkdir -p $O
SSRHIP_GEMV_PAIR_EARLY=2 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q -k "pair_launch or config2" 2>&1 | tail -2
timeout 900 python tools/decode_ab.py --steps 300 --warmup 20 --reps 4 early3: early4:SSRHIP_GEMV_PAIR_EARLY=2 noearly:SSRHIP_GEMV_PAIR_EARLY=0 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/decode_ab_pair_early_pf.log
