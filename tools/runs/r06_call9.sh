#!/bin/bash
# round 6, call 9: why does `rocprofv3 --pmc` of bench.py crash since this round? bisect: (a) as is, (b) random prompt (no codec in the
# process), (c) no pair launches, (d) both off; keep the first counter file that appears
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c9; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_$name -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/$name.out 2> $O/$name.err
  echo "$name rc=$? counter files: $(ls $O/p_$name/*/*counter_collection.csv 2>/dev/null | wc -l)"
}
run random_prompt BENCH_RANDOM_PROMPT=1
run nopair SSRHIP_GEMV_PAIR=0
run both_off BENCH_RANDOM_PROMPT=1 SSRHIP_GEMV_PAIR=0
run presize_off SSRHIP_CODEC_PRESIZE=0
cd $R
for n in random_prompt nopair both_off presize_off; do
  f=$(ls $O/p_$n/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py $f $O/fetch_$n.md | grep -i "gemv\|attn_decode\|sample" | head -8; fi
done
# WRITE_SIZE with whichever configuration worked first
for n in random_prompt presize_off nopair both_off; do
  if [ -f $O/fetch_$n.md ]; then
    case $n in random_prompt) E="BENCH_RANDOM_PROMPT=1";; presize_off) E="SSRHIP_CODEC_PRESIZE=0";; nopair) E="SSRHIP_GEMV_PAIR=0";; both_off) E="BENCH_RANDOM_PROMPT=1 SSRHIP_GEMV_PAIR=0";; esac
    cd /tmp; env $E timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w_$n -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1; cd $R
    python tools/pmc_summary.py $(ls $O/w_$n/*/*counter_collection.csv | head -1) $O/write_$n.md | grep -i "gemv\|attn_decode\|sample" | head -8
    break
  fi
done
rm -rf $O/p_* $O/w_*
