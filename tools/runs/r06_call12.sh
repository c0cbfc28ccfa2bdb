#!/bin/bash
# round 6, call 12: 16-row LayerNorm launches with the first weight requests in front of the x requests (SSRHIP_GEMVM_WFIRST=1) against the shipped order
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c12; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "edge_kernel" 2>&1 | tail -3
for e in 1 0 1 0; do SSRHIP_GEMVM_WFIRST=$e tools/bin/gemvm_bench 16 1 1 2>&1 | grep -v amdgpu.ids | sed "s/^/wfirst=$e /"; done | tee $O/gemvm_bench_16_wfirst.log
for e in 1 0 1 0; do SSRHIP_GEMVM_WFIRST=$e timeout 600 python bench.py --utts 8 --no-extras --no-cpu-baseline --no-ctx700 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wfirst=$e', d['ms_per_step'], d['ms_per_step_passes'], d['value'], d['roofline']['us_per_launch'])"; done | tee $O/bench8_wfirst.log
