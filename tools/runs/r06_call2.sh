#!/bin/bash
# round 6, call 2: (a) preempt_lab with THREE checked streams (the codec trials said: several queues busy + fresh hipMallocs);
# (b) codec trials: base / warm allocator / warm allocator + a thread that keeps mapping fresh memory; (c) the pairing-slot tests.
O=gpurun_out/r6c2; mkdir -p $O
export TMPDIR=/tmp
for rep in $(seq 1 8); do
  for arm in none malloc malloctouch freshout; do
    timeout 120 tools/bin/preempt_lab $arm 100 40 3 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/preempt_lab_3streams.log
grep -c CORRUPTED $O/preempt_lab_3streams.log
T=${1:-40}
timeout 1500 python tools/race_trials.py $T \
  base:SSRHIP_POISON_ALLOC=1 \
  plain: \
  warm:SSRHIP_POISON_ALLOC=1,warm-alloc \
  warmbg:SSRHIP_POISON_ALLOC=1,warm-alloc,bg-malloc 2>&1 | grep -v amdgpu.ids | tee $O/race_trials.log | grep -v "^            item\|^        caller\|^    FAIL" | tail -25
timeout 900 python -m pytest tests/test_gpu_pair_guard.py -x -q 2>&1 | tail -15 | tee $O/pytest_pair_guard.log
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -k "config2" 2>&1 | tail -5 | tee -a $O/pytest_pair_guard.log
