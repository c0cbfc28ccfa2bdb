#!/bin/bash
# round 6, call 14: the squatter test now waits until all 200 squatters are resident — five runs of the file, then the whole suite once more
O=gpurun_out/r6c14; mkdir -p $O
for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_pair_guard.py -q -p no:cacheprovider 2>&1 | tail -1; done | tee $O/pair_guard_x5.log
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=15 > $O/pytest_gpu_full.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok"
