# round 5, GPU call 4: decode attention — when to request the V rows (SSRHIP_ATTN_VAT) — on the head-fastest grid
O=gpurun_out/r5d; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn" 2>&1 | tail -3 | tee $O/pytest_attn.log
timeout 400 python tools/decode_ab.py --reps 3 hipcc: vat4:SSRHIP_ATTN_VAT=4 vat8:SSRHIP_ATTN_VAT=8 vat12:SSRHIP_ATTN_VAT=12 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/decode_ab_attn_vat.log
