# round 4: segment GEMV with x fetched once per workgroup through LDS (SSRHIP_GEMV_SEG_XLDS) vs every wave its own slice, same box
O=gpurun_out/r4x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv_step_shapes or seg_combine or gemv_matches or grouped_heads" 2>&1 | tail -3
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], r['us_per_launch'], r['other_kernels_us_per_launch'], r['event_timed_us_per_launch'])
" $1 $2; }
for rep in 1 2; do
  SSRHIP_GEMV_SEG_XLDS=0 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_off$rep.json 2>/dev/null; show $O/ab_off$rep.json perwave
  SSRHIP_GEMV_SEG_XLDS=1 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_on$rep.json 2>/dev/null; show $O/ab_on$rep.json xlds
done
