# round 5, GPU call 8: the weights requested only when x has ARRIVED — 16-row kernels (lab stamps, bench, step) and 2-row segment kernels
O=gpurun_out/r5h; mkdir -p $O
for v in 0 1; do echo "SSRHIP_GEMVM_XFIRST=$v" | tee -a $O/gemvm_lab_xwait.log; SSRHIP_GEMVM_XFIRST=$v timeout 120 tools/bin/gemvm_lab 16 2>&1 | tee -a $O/gemvm_lab_xwait.log; done
for v in 0 1; do echo "SSRHIP_GEMVM_XFIRST=$v" | tee -a $O/gemvm_bench_16.log; SSRHIP_GEMVM_XFIRST=$v timeout 60 tools/bin/gemvm_bench 16 1 1 2>&1 | tee -a $O/gemvm_bench_16.log; done
for v in 0 1 2 3; do echo "SSRHIP_GEMV_XFIRST=$v" | tee -a $O/gemvm_bench_2.log; SSRHIP_GEMV_XFIRST=$v timeout 60 tools/bin/gemvm_bench 2 0 0 2>&1 | tee -a $O/gemvm_bench_2.log; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "mfma or rows or tiled or sixteen or gemvm or segu or seg_combine or gemv_matches" 2>&1 | tail -3 | tee $O/pytest_gemv.log
timeout 400 python tools/decode_ab.py --utts 8 --steps 200 --reps 3 r4order:SSRHIP_GEMVM_XFIRST=0 xwait:SSRHIP_GEMVM_XFIRST=1 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_16rows_xwait.log
timeout 400 python tools/decode_ab.py --reps 3 r4order:SSRHIP_GEMV_XFIRST=0 xwait_x:SSRHIP_GEMV_XFIRST=1 xwait_merge:SSRHIP_GEMV_XFIRST=2 xwait_both:SSRHIP_GEMV_XFIRST=3 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_ab_xwait.log
