#!/bin/bash
# round 6: the whole GPU suite on the current build (the log is copied to profiles/r06_pytest_gpu.log), then smoke()
O=gpurun_out/r6s; mkdir -p $O
timeout 3000 python -m pytest tests/ -q -m gpu --durations=15 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -45 $O/pytest_gpu_full.log | cut -c1-250 | tee $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.log
