#!/bin/bash
# round 6, call 16: the driver's own bench command line on the final build
O=gpurun_out/r6c16; mkdir -p $O
T0=$(date +%s); python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "rc=$?"
echo "wall $(( $(date +%s) - T0 )) s"
wc -l $O/bench_driver_cmd.json
python - <<PY
import json; d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","ms_per_step_passes","higher_is_better","scaling","vs_baseline","dtype","data","ms_per_step_ctx700","attention_share_of_step","speedup_vs_cpu")})
print(d["config"]); print(d["roofline"]["frac"], d["roofline"]["achieved"], d["roofline"]["us_per_launch"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
PY
