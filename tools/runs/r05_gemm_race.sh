#!/bin/bash
mkdir -p gpurun_out/race
for k in 1 2 3; do timeout 300 python tools/gemm_race_lab.py 8 12 2>&1 | grep -v amdgpu.ids | cut -c1-400; done | tee gpurun_out/race/gemm_race_lab.log
