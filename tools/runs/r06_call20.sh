#!/bin/bash
# round 6, call 20: final bench lines, decode traces and PMC passes again (the merge pair launch changed), driver command line, whole suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6z; mkdir -p $O
cd $R
python bench.py > $O/r06_bench_n1.json 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_n1_driver_cmd.json 2>/dev/null; echo "driver-cmd bench rc=$?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python $R/bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1_$c -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline --no-ctx700 > /dev/null 2>&1
done
cd $R
python tools/prof_summary.py $(ls $O/kt1/*/*kernel_trace.csv | head -1) $O/r06_decode_kernel_trace_summary.md | tail -2
cp $(ls $O/kt1/*/*kernel_stats.csv | head -1) $O/r06_rocprofv3_kernel_stats.csv
python tools/pmc_summary.py $(ls $O/pmc1_FETCH_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_fetch_size.md | grep -i "gemv" | head -6
python tools/pmc_summary.py $(ls $O/pmc1_WRITE_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_write_size.md | grep -i "gemv" | head -6
rm -rf $O/kt1 $O/pmc1_*
python - <<PY
import json
for f in ("r06_bench_n1.json","r06_bench_n1_driver_cmd.json"):
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["ms_per_step_passes"], d["ms_per_step_ctx700"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["roofline"]["step_level"]["frac"], d.get("speedup_vs_cpu"))
    for k in ("rtf_10s_tts","codec256"):
        if k in d: print("  ", k, {kk: vv for kk, vv in d[k].items() if kk in ("rtf","wall_ms","time_to_first_16_frames_ms","encode_ms","decode_ms","wmdecode_ms","wmdecode_with_detector_ms")})
    if "dp64" in d: print("  dp64", d["dp64"]["codec_tokens_per_s_per_gpu"], d["dp64"]["wall_ms_with_codec"], "ragged", d["dp64_ragged"]["speedup"])
PY
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=15 -rs > $O/pytest_gpu_full.log 2>&1; echo "suite rc=$?"; grep -n "passed\|failed" $O/pytest_gpu_full.log | tail -2; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok"
