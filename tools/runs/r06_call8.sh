#!/bin/bash
# round 6, call 8: decode PMC passes again (bench.py no longer empties the allocator cache under the profiler), the decode kernel
# traces again (tools/prof_summary.py picks whole-step windows now), then the whole GPU suite on the final build + smoke
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c8; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1_$c -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/pmc_$c.out 2> $O/pmc_$c.err
  echo "pmc $c rc=$?"
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python $R/bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt8 -- python $R/bench.py --steps 200 --warmup 20 --utts 8 --no-extras --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $(ls $O/pmc1_FETCH_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_fetch_size.md | grep -i "gemv\|attn\|sample" | head -12
python tools/pmc_summary.py $(ls $O/pmc1_WRITE_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_write_size.md | grep -i "gemv\|attn\|sample" | head -12
python tools/prof_summary.py $(ls $O/kt1/*/*kernel_trace.csv | head -1) $O/r06_decode_kernel_trace_summary.md | tail -3
cp $(ls $O/kt1/*/*kernel_stats.csv | head -1) $O/r06_rocprofv3_kernel_stats.csv
python tools/prof_summary.py $(ls $O/kt8/*/*kernel_trace.csv | head -1) $O/r06_decode16rows_kernel_trace_summary.md | tail -3
cp $(ls $O/kt8/*/*kernel_stats.csv | head -1) $O/r06_rocprofv3_kernel_stats_16rows.csv
rm -rf $O/pmc1_* $O/kt1 $O/kt8
bash tools/runs/r06_suite.sh
