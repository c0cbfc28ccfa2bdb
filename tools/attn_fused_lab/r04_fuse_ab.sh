# round 4: fused attention + out-projection launch (csrc/attn_fused.hip) vs the two launches, same box, alternating (SSRHIP_FUSE_ATTN)
O=gpurun_out/r4a; mkdir -p $O
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], r['us_per_launch'], r['other_kernels_us_per_launch'], r['event_timed_us_per_launch'])
" $1 $2; }
for rep in 1 2; do
  SSRHIP_FUSE_ATTN=0 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_two$rep.json 2>$O/ab_two$rep.err; show $O/ab_two$rep.json two
  SSRHIP_FUSE_ATTN=1 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_fused$rep.json 2>$O/ab_fused$rep.err; show $O/ab_fused$rep.json fused
done
