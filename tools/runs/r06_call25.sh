#!/bin/bash
# round 6, call 25: the whole GPU suite on the final build (after the LSTM pipeline threshold), smoke
O=gpurun_out/r6c25; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=15 -rs > $O/pytest_gpu_full.log 2>&1; echo "suite rc=$?"; grep -n "passed\|failed\|AssertionError: round" $O/pytest_gpu_full.log | tail -3 | cut -c1-500
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok"
