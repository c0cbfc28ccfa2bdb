O=gpurun_out/r5p; mkdir -p $O
for v in 0 1; do echo "SSRHIP_GEMV_LN_LOCAL=$v" | tee -a $O/gemvm_bench_2_lnlocal.log; SSRHIP_GEMV_LN_LOCAL=$v timeout 60 tools/bin/gemvm_bench 2 0 0 2>&1 | tee -a $O/gemvm_bench_2_lnlocal.log; done
timeout 400 python tools/decode_ab.py --reps 4 lds_exchange: ln_local:SSRHIP_GEMV_LN_LOCAL=1 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_ab_lnlocal.log
