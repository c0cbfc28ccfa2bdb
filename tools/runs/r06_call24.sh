#!/bin/bash
# round 6, call 24: the LSTM pipeline from 8 items on — codec / pipeline / ragged test files, then the API leg of the bench
O=gpurun_out/r6c24; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_codec.py tests/test_gpu_pipeline.py tests/test_gpu_ragged.py -q -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest_codec_files.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs rtf_10s_tts 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('rtf_10s_tts'))" | tee $O/bench_rtf.log
