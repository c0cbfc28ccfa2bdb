# round 5: the whole GPU suite on the current build (the log is copied to profiles/r05_pytest_gpu.log)
O=gpurun_out/r5s; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu --durations=15 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -40 $O/pytest_gpu_full.log | tee $O/pytest_gpu.log
