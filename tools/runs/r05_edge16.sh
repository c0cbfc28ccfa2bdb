#!/bin/bash
mkdir -p gpurun_out
timeout 120 tools/bin/edge16_lab 2>&1 | tee gpurun_out/edge16_lab.log
