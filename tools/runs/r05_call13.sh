O=gpurun_out/r5m; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "segu" 2>&1 | tail -3 | tee $O/pytest_segu.log
for v in 0 1; do echo "SSRHIP_GEMV_SEGU_RAMP=$v" | tee -a $O/gemvm_bench_2_ramp.log; SSRHIP_GEMV_SEGU_RAMP=$v timeout 60 tools/bin/gemvm_bench 2 0 0 2>&1 | tee -a $O/gemvm_bench_2_ramp.log; done
timeout 400 python tools/decode_ab.py --reps 4 all_at_entry: ramp:SSRHIP_GEMV_SEGU_RAMP=1 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_ab_ramp.log
