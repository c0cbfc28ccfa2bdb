# round 5, GPU call 11: two-phase admission — refill parity tests, then the ragged queue with the knob off / on (same process: bench leg twice)
O=gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lm.py -x -q -k "refill or run_queue or queue" 2>&1 | tail -4 | tee $O/pytest_refill.log
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "queue" 2>&1 | tail -4 | tee -a $O/pytest_refill.log
for v in 0 1; do echo "SSRHIP_ADMIT_TWO_PHASE=$v" | tee -a $O/dp64_ragged_ab.log; SSRHIP_ADMIT_TWO_PHASE=$v timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --legs dp64_ragged 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d.get('dp64_ragged')))" | tee -a $O/dp64_ragged_ab.log; done
