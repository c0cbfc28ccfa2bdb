#!/bin/bash
# round 6, call 13: the whole GPU suite twice more on the final build (round 5: one of three full runs was red in the multi-stream codec test)
O=gpurun_out/r6c13; mkdir -p $O
for i in 2 3; do
  timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $O/pytest_gpu_run$i.log 2>&1; echo "run $i rc=$?"; tail -2 $O/pytest_gpu_run$i.log
done
