#!/bin/bash
mkdir -p gpurun_out/race
for k in $(seq 1 20); do SSRHIP_POISON_ALLOC=1 timeout 120 python tools/race_first_round.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500; done | tee gpurun_out/race/race_first_round_poison.log
