#!/bin/bash
# round 6, call 7: the decode PMC passes of the final build (they produced no counter file in call 6: keep the profiler's output this time),
# then more per-process trials of the codec fix
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c7; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1_$c -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/pmc_$c.out 2> $O/pmc_$c.err
  echo "pmc $c rc=$?"; tail -5 $O/pmc_$c.err | cut -c1-300; ls $O/pmc1_$c/* | head -5
done
cd $R
python tools/pmc_summary.py $(ls $O/pmc1_FETCH_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_fetch_size.md | tail -12
python tools/pmc_summary.py $(ls $O/pmc1_WRITE_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_write_size.md | tail -12
rm -rf $O/pmc1_*
timeout 1500 python tools/race_trials.py ${1:-40} off:SSRHIP_POISON_ALLOC=1,SSRHIP_CODEC_PRESIZE=0,SSRHIP_RECORD_STREAM=1,rounds=3 off_nors:SSRHIP_POISON_ALLOC=1,SSRHIP_CODEC_PRESIZE=0,rounds=3 on:SSRHIP_POISON_ALLOC=1,rounds=3 2>&1 | grep -v amdgpu.ids | tee $O/race_trials_ab3.log | grep -v "^            item\|^    FAIL" | tail -12
