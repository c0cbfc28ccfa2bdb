#!/bin/bash
mkdir -p gpurun_out/race
for k in $(seq 1 9); do SSRHIP_POISON_ALLOC=1 timeout 120 python tools/race_first_round.py 2>&1 | grep -v amdgpu.ids | grep -v "per-frame max" | cut -c1-700; done | tee gpurun_out/race/race_first_round_rows.log
