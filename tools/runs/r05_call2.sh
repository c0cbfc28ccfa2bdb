# round 5, GPU call 2: the decode-step changes (profiling stamps compiled out, scalar kv chain, epilogue operands first, merge prologue without
# the drain, attention K/V requested together, gemv_segu_kernel) — kernel + LM parity first, then same-box A/Bs.
O=gpurun_out/r5b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q --durations=8 -k "segu or seg_combine or attn_decode or gemv_matches or grouped_heads or attn_matches or sampler" 2>&1 | tail -14 | tee $O/pytest_kernels.log
timeout 600 python -m pytest tests/test_gpu_lm.py -x -q --durations=5 2>&1 | tail -10 | tee $O/pytest_lm.log
for v in 0 2 4; do echo "SSRHIP_GEMV_SEGU=$v" | tee -a $O/gemvm_bench_b2.log; SSRHIP_GEMV_SEGU=$v timeout 60 tools/bin/gemvm_bench 2 0 0 2>&1 | tee -a $O/gemvm_bench_b2.log; done
timeout 400 python tools/decode_ab.py --reps 3 r4sched:SSRHIP_GEMV_SEGU=0,SSRHIP_ATTN_PIN=0 pin:SSRHIP_GEMV_SEGU=0 segu2:SSRHIP_GEMV_SEGU=2 segu4: 2>&1 | grep -v Warning | tee $O/decode_ab.log
# the round-4 library on the same box (ABI 102 as well): swapped in for one run
L=ssr-speech_amd/csrc/libssrhip.so; cp $L /tmp/new.so; cp tools/bin/libssrhip_r04.so $L
timeout 300 python tools/decode_ab.py --reps 3 r04lib: 2>&1 | grep -v Warning | tee $O/decode_ab_r04lib.log
cp /tmp/new.so $L
timeout 300 python tools/decode_ab.py --reps 2 segu4_again: 2>&1 | grep -v Warning | tee -a $O/decode_ab.log
timeout 60 tools/bin/sampler_bench 40 0.8 2>&1 | tee $O/sampler_bench.log
for b in 32 64 128; do timeout 60 tools/bin/lstm_split_lab $b 1024 200 2>&1 | tee -a $O/lstm_split_lab_b.log; done
