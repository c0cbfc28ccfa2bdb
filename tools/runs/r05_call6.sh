# round 5, GPU call 6: 16-row GEMV with all x requests of a workgroup in front of its weight requests (SSRHIP_GEMVM_XFIRST)
O=gpurun_out/r5f; mkdir -p $O
for v in 0 1; do echo "SSRHIP_GEMVM_XFIRST=$v" | tee -a $O/gemvm_lab_xfirst.log; SSRHIP_GEMVM_XFIRST=$v timeout 120 tools/bin/gemvm_lab 16 2>&1 | tee -a $O/gemvm_lab_xfirst.log; done
for v in 0 1; do echo "SSRHIP_GEMVM_XFIRST=$v" | tee -a $O/gemvm_bench_16.log; SSRHIP_GEMVM_XFIRST=$v timeout 60 tools/bin/gemvm_bench 16 1 1 2>&1 | tee -a $O/gemvm_bench_16.log; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "mfma or rows or tiled or sixteen or gemvm" 2>&1 | tail -3 | tee $O/pytest_gemvm.log
timeout 400 python tools/decode_ab.py --utts 8 --steps 200 --reps 3 r4order:SSRHIP_GEMVM_XFIRST=0 xfirst: 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_16rows_xfirst.log
timeout 400 python tools/decode_ab.py --utts 4 --steps 200 --reps 2 r4order:SSRHIP_GEMVM_XFIRST=0 xfirst: 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/decode_16rows_xfirst.log
