#!/bin/bash
mkdir -p gpurun_out/race
for k in $(seq 1 14); do timeout 120 python tools/race_first_round.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/race/race_first_round.log
