#!/bin/bash
# round 6, call 4: (a) tests of what changed since call 3 (codec without record_stream + sizing passes, 200-workgroup squatter, N-tile
# groups of the XCD order), (b) bench.py once (ctx700 pass, demo prompt, pairing info), (c) config-5 codec timing + FETCH_SIZE with the
# grouped order, (d) more per-process trials of the codec fix.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c4; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_codec.py -x -q -k "sized or concurrent_streams" 2>&1 | tail -12 | tee $O/pytest_codec_sized.log
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_codec.py -x -q -k "concurrent_streams" 2>&1 | tail -2; done | tee -a $O/pytest_codec_sized.log
timeout 900 python -m pytest tests/test_gpu_pair_guard.py -x -q 2>&1 | tail -8 | tee $O/pytest_guard.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "xcd_tile or edge_kernel or gemm_split" 2>&1 | tail -3 | tee $O/pytest_kernels.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench_n1.json | head -c 400; echo
python - <<PY
import json; d=json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","ms_per_step","ms_per_step_passes","ms_per_step_ctx700","ctx700","attention_share_of_step")}); print(d["config"]); print(d["roofline"]["frac"], d["roofline"]["us_per_launch"]); print(d.get("cpu_baseline")); print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in list(v.items())[:8]}) for k, v in d.get("extras", {}).items()})
PY
for x in 1 0 1 0; do SSRHIP_GEMM_XCD=$x timeout 600 python tools/codec_bench.py 256 30 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/xcd=$x /"; done | tee $O/codec256_xcd_groups.log
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fs_codec -- python $R/tools/codec_bench.py 256 30 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $(ls $O/fs_codec/*/*counter_collection.csv | head -1) $O/r06_codec_b256_pmc_fetch_size.md > /dev/null; rm -rf $O/fs_codec
grep gemm_split $O/r06_codec_b256_pmc_fetch_size.md
timeout 1300 python tools/race_trials.py ${1:-90} on:SSRHIP_POISON_ALLOC=1,rounds=3 2>&1 | grep -v amdgpu.ids | tee $O/race_trials_on.log | grep -v "^            item\|^    FAIL" | tail -6
timeout 800 python tools/race_trials.py ${2:-25} off:SSRHIP_POISON_ALLOC=1,SSRHIP_CODEC_PRESIZE=0,rounds=3 on:SSRHIP_POISON_ALLOC=1,rounds=3 2>&1 | grep -v amdgpu.ids | tee $O/race_trials_ab2.log | grep -v "^            item\|^    FAIL" | tail -8
