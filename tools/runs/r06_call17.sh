#!/bin/bash
# round 6, call 17: the merge pair launch with B's first two units requested at entry (SSRHIP_GEMV_PAIR_EARLY, default on) — bit-identity
# tests, then the same-process alternating A/B of the 2-row step
O=gpurun_out/r6c17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_lm.py -x -q -k "pair_launch or config2 or config1 or 830m_greedy or tokens_match_reference" 2>&1 | tail -4 | tee $O/pytest_pair_early.log
timeout 900 python tools/decode_ab.py --steps 300 --warmup 20 --reps 4 early: noearly:SSRHIP_GEMV_PAIR_EARLY=0 2>&1 | grep -v amdgpu.ids | tail -20 | tee $O/decode_ab_pair_early.log
