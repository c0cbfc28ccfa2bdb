#!/bin/bash
# round 6, call 3: first GPU run of (a) the new kernel paths' tests, (b) 16-row GEMV with / without the edge-wave LayerNorm kernel,
# (c) config-5 codec with / without the XCD-aware tile order (+ FETCH_SIZE), (d) the codec fix A/B: sizing passes off vs on, 100 trials each.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c3; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "edge_kernel or xcd_tile or mfma or gemm_split" 2>&1 | tail -8 | tee $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_gpu_pair_guard.py tests/test_gpu_pipeline.py -x -q -k "pair or squat or engine or process or mask_spans" 2>&1 | tail -8 | tee $O/pytest_guard_cli.log
timeout 900 python -m pytest tests/test_gpu_codec.py -x -q -k "sized or concurrent_streams" 2>&1 | tail -8 | tee $O/pytest_codec_sized.log
for e in 1 0 1 0; do SSRHIP_GEMVM_EDGE=$e tools/bin/gemvm_bench 16 1 1 2>&1 | grep -v amdgpu.ids | sed "s/^/edge=$e /"; done | tee $O/gemvm_bench_16_edge.log
for e in 1 0; do SSRHIP_GEMVM_EDGE=$e timeout 600 python bench.py --utts 8 --no-extras --no-cpu-baseline 2> $O/bench8_$e.err > $O/bench_n1_8utts_edge$e.json; python - <<PY
import json; d=json.loads(open("$O/bench_n1_8utts_edge$e.json").read().strip().splitlines()[-1]); print("edge=$e", d["ms_per_step"], d["value"], d["roofline"]["us_per_launch"], d["roofline"]["event_timed_us_per_launch"])
PY
done | tee $O/bench8_edge.log
for x in 1 0 1 0; do SSRHIP_GEMM_XCD=$x timeout 600 python tools/codec_bench.py 256 30 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/xcd=$x /"; done | tee $O/codec256_xcd.log
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fs_codec -- python $R/tools/codec_bench.py 256 30 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $(ls $O/fs_codec/*/*counter_collection.csv | head -1) $O/r06_codec_b256_pmc_fetch_size.md > /dev/null; rm -rf $O/fs_codec
head -40 $O/r06_codec_b256_pmc_fetch_size.md
T=${1:-100}
timeout 1500 python tools/race_trials.py $T \
  off:SSRHIP_POISON_ALLOC=1,SSRHIP_CODEC_PRESIZE=0,rounds=3 \
  on:SSRHIP_POISON_ALLOC=1,rounds=3 2>&1 | grep -v amdgpu.ids | tee $O/race_trials_fix_ab.log | grep -v "^            item\|^    FAIL" | tail -30
