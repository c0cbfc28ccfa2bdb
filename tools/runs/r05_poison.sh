#!/bin/bash
mkdir -p gpurun_out/race
timeout 600 python tools/poison_check.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/race/poison_check.log
