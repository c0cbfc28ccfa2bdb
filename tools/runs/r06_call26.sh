#!/bin/bash
# round 6, call 26: the driver's bench command line on the last commit of the round
O=gpurun_out/r6c26; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_last_commit.json 2>/dev/null; echo "rc=$?"
python - <<PY
import json; d=json.loads(open("$O/bench_driver_cmd_last_commit.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_passes"], d["ms_per_step_ctx700"], d["roofline"]["frac"], d["rtf_10s_tts"]["rtf"], d["rtf_10s_tts"]["wall_ms"], d["codec256"]["encode_ms"], d["codec256"]["decode_ms"], d["speedup_vs_cpu"])
PY
