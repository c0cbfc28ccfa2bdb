O=gpurun_out/r5s; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.log
