O=gpurun_out/r5n; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "segu or seg_combine or gemv_matches or grouped_heads or mfma or rows" 2>&1 | tail -3 | tee $O/pytest.log
timeout 60 tools/bin/gemvm_bench 2 0 0 2>&1 | tee $O/gemvm_bench_2.log
timeout 400 python tools/decode_ab.py --reps 3 final: 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_ab_final.log
timeout 400 python tools/decode_ab.py --utts 8 --steps 200 --reps 2 final16: 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/decode_ab_final.log
