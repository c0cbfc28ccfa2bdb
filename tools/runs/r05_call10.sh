O=gpurun_out/r5j; mkdir -p $O
for v in 0 1; do echo "SSRHIP_GEMVM_XFIRST=$v" | tee -a $O/gemvm_lab_ln_first.log; SSRHIP_GEMVM_XFIRST=$v timeout 120 tools/bin/gemvm_lab 16 2>&1 | tee -a $O/gemvm_lab_ln_first.log; done
for v in 0 1; do echo "SSRHIP_GEMVM_XFIRST=$v" | tee -a $O/gemvm_bench_16.log; SSRHIP_GEMVM_XFIRST=$v timeout 60 tools/bin/gemvm_bench 16 1 1 2>&1 | tee -a $O/gemvm_bench_16.log; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "mfma or rows or tiled or sixteen or gemvm" 2>&1 | tail -3 | tee $O/pytest_gemv.log
timeout 400 python tools/decode_ab.py --utts 8 --steps 200 --reps 3 r4order:SSRHIP_GEMVM_XFIRST=0 ln_first:SSRHIP_GEMVM_XFIRST=1 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_16rows_ln_first.log
