#!/bin/bash
mkdir -p gpurun_out/race
for k in 1 2; do timeout 600 python tools/race_probe.py 16 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/race/race_probe.log
