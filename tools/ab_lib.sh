# A/B of two builds of libssrhip.so on the SAME box: tools/bin/libssrhip_prev.so vs the in-tree one (bench.py batch-1 step, alternating)
O=gpurun_out/r2n; mkdir -p $O
L=ssr-speech_amd/csrc/libssrhip.so
cp $L /tmp/new.so
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], r['us_per_launch'], r['other_kernels_us_per_launch'], r['event_timed_us_per_launch'])
" $1 $2; }
for rep in 1 2; do
  cp tools/bin/libssrhip_prev.so $L; python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_prev$rep.json 2>/dev/null; show $O/ab_prev$rep.json prev
  cp /tmp/new.so $L;                 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_new$rep.json 2>/dev/null; show $O/ab_new$rep.json new
done
cp /tmp/new.so $L
