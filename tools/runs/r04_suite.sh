# round 4: the whole -m gpu suite + prefill timing (split vs chain) + the batch-1 bench line
O=gpurun_out/r4e; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.log
python tools/prefill_time.py 2>&1 | grep -v amdgpu.ids | tee $O/prefill_split.log
SSRHIP_PREFILL_SPLIT=0 python tools/prefill_time.py 2>&1 | grep -v amdgpu.ids | tee $O/prefill_chain.log
python bench.py --no-cpu-baseline --steps 300 --warmup 20 --legs rtf_10s_tts > $O/bench_rtf.json 2>$O/bench_rtf.err; tail -c 1500 $O/bench_rtf.json
