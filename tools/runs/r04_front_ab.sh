# round 4: "front" GEMV (all units requested at entry, one 8-wave workgroup per CU; since removed from csrc/gemv.hip — kept for the record of how profiles/r04_microbench/decode_ab.log was taken)
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv_step_shapes or seg_combine or grouped_heads" 2>&1 | tail -3 | tee $O/pytest_kernels.log
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], r['us_per_launch'], r['other_kernels_us_per_launch'], r['event_timed_us_per_launch'])
" $1 $2; }
for rep in 1 2; do
  SSRHIP_GEMV_FRONT=0 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_seg$rep.json 2>$O/ab_seg$rep.err; show $O/ab_seg$rep.json seg
  SSRHIP_GEMV_FRONT=1 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/ab_front$rep.json 2>$O/ab_front$rep.err; show $O/ab_front$rep.json front
done
