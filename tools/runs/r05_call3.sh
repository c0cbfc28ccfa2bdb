# round 5, GPU call 3: (a) lab: does prefetching the head of the NEXT launch's matrix into the Infinity Cache shorten the chain? (b) decode
# attention with the head as the fastest grid dimension (live items spread over all 8 XCDs) vs round 4's page-fastest order.
O=gpurun_out/r5c; mkdir -p $O
timeout 120 tools/bin/gemv_floor_lab 2>&1 | tee $O/gemv_floor_lab_prefetch.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn" 2>&1 | tail -4 | tee $O/pytest_attn.log
timeout 400 python tools/decode_ab.py --reps 3 pagefast:SSRHIP_ATTN_HEAD_FASTEST=0 headfast: 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_ab_attn_grid.log
timeout 400 python tools/decode_ab.py --utts 2 --reps 2 pagefast:SSRHIP_ATTN_HEAD_FASTEST=0 headfast: 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/decode_ab_attn_grid.log
timeout 400 python tools/decode_ab.py --utts 8 --steps 200 --reps 2 base16: 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_16rows.log
