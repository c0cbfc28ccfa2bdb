#!/bin/bash
# round 6, call 15: the squatter on a high-priority stream (its own hardware queue); the whole suite once more
O=gpurun_out/r6c15; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=15 -rs > $O/pytest_gpu_full.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_gpu_full.log; grep -n "SKIPPED.*squat\|squat" $O/pytest_gpu_full.log | head -5
