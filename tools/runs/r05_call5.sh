# round 5, GPU call 5: 16-row GEMV — parity after the staged kv chain, where a launch's time goes (stamps), and the step
O=gpurun_out/r5e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "mfma or rows or tiled or sixteen or gemvm" 2>&1 | tail -4 | tee $O/pytest_gemvm.log
timeout 120 tools/bin/gemvm_lab 16 2>&1 | tee $O/gemvm_lab.log
timeout 60 tools/bin/gemvm_bench 16 1 1 2>&1 | tee $O/gemvm_bench_16.log
timeout 400 python tools/decode_ab.py --utts 8 --steps 200 --reps 2 staged_kv: 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_16rows.log
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -k "config4 or sixteen" 2>&1 | tail -4 | tee $O/pytest_config4.log
