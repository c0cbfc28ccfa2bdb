#!/bin/bash
# round 6, call 10: decode PMC passes without the context-700 region (the only thing bench.py does under the profiler that round 5's did not:
# ~400 graph replays queued behind one synchronisation)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c10; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1_$c -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline --no-ctx700 > $O/pmc_$c.out 2> $O/pmc_$c.err
  echo "pmc $c rc=$? files $(ls $O/pmc1_$c/*/*counter_collection.csv 2>/dev/null | wc -l)"
done
cd $R
python tools/pmc_summary.py $(ls $O/pmc1_FETCH_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_fetch_size.md | grep -i "gemv\|attn_decode\|sample" | head -10
python tools/pmc_summary.py $(ls $O/pmc1_WRITE_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_write_size.md | grep -i "gemv\|attn_decode\|sample" | head -10
rm -rf $O/pmc1_*
