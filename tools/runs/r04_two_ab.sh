# round 4: segment GEMV with both units of a wave requested at entry (out-projection, head-MLP1) vs the in-place re-request, same box
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv_step_shapes or seg_combine or gemv_matches or grouped_heads" 2>&1 | tail -3 | tee $O/pytest_kernels.log
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], r['us_per_launch'], r['other_kernels_us_per_launch'], r['event_timed_us_per_launch'])
" $1 $2; }
for rep in 1 2; do
  SSRHIP_GEMV_SEG_TWO=0 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_inplace$rep.json 2>$O/ab_inplace$rep.err; show $O/ab_inplace$rep.json inplace
  SSRHIP_GEMV_SEG_TWO=1 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_two$rep.json 2>$O/ab_two$rep.err; show $O/ab_two$rep.json twounits
done
