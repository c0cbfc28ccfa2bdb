#!/bin/bash
# round 6, call 21: the attention launch pre-touching the next launch's W_o slice (SSRHIP_ATTN_PREFETCH, default on) — parity tests, then A/B
O=gpurun_out/r6c21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_lm.py -x -q -k "attn or pair_launch or config2 or config1 or 830m_greedy or tokens_match_reference" 2>&1 | tail -3 | tee $O/pytest_prefetch.log
timeout 900 python tools/decode_ab.py --steps 300 --warmup 20 --reps 4 prefetch: noprefetch:SSRHIP_ATTN_PREFETCH=0 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/decode_ab_attn_prefetch.log
timeout 900 python tools/decode_ab.py --steps 20 --warmup 5 --reps 6 prefetch: noprefetch:SSRHIP_ATTN_PREFETCH=0 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/decode_ab_attn_prefetch.log
