#!/bin/bash
# round 6, call 1: what stops-and-resumes our waves? (a) tools/preempt_lab: torch-free, one host activity per arm, 10 processes per arm;
# (b) tools/race_trials.py: the codec's first concurrent round, one fresh process per trial, arms interleaved.
O=gpurun_out/r6c1; mkdir -p $O
export TMPDIR=/tmp
for rep in $(seq 1 10); do
  for arm in none streams malloc hostfree d2hfree hostreg events; do
    timeout 120 tools/bin/preempt_lab $arm 100 100 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/preempt_lab.log
echo "--- with 16 hardware queues allowed"
for rep in $(seq 1 6); do GPU_MAX_HW_QUEUES=16 timeout 120 tools/bin/preempt_lab streams 100 100 2>&1 | grep -v amdgpu.ids; done | tee -a $O/preempt_lab.log
grep -c CORRUPTED $O/preempt_lab.log
T=${1:-25}
timeout 2400 python tools/race_trials.py $T \
  base:SSRHIP_POISON_ALLOC=1 \
  queues:SSRHIP_POISON_ALLOC=1,prewarm-queues=8 \
  early:SSRHIP_POISON_ALLOC=1,early-streams \
  warm:SSRHIP_POISON_ALLOC=1,warm-alloc \
  nocache:SSRHIP_POISON_ALLOC=1,PYTORCH_NO_CUDA_MEMORY_CACHING=1 \
  hwq1:SSRHIP_POISON_ALLOC=1,GPU_MAX_HW_QUEUES=1 2>&1 | grep -v amdgpu.ids | tee $O/race_trials.log | tail -40
