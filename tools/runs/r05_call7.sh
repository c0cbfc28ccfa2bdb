# round 5, GPU call 7: (a) the cost of every CU reading the same block behind a kernel boundary; (b) 2-row segment kernels with the x-first barrier
O=gpurun_out/r5g; mkdir -p $O
timeout 120 tools/bin/xbcast_lab 2>&1 | tee $O/xbcast_lab.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "segu or seg_combine or gemv_matches or grouped_heads" 2>&1 | tail -3 | tee $O/pytest_kernels.log
timeout 400 python tools/decode_ab.py --reps 3 r4order:SSRHIP_GEMV_XFIRST=0 xfirst: 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_ab_xfirst.log
