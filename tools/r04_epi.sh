# round 4: batched epilogues (resblock_split_dma_kernel, gemm_split_dma_kernel): lab timings, the kernels' parity tests, config-5 codec
O=gpurun_out/r4g; mkdir -p $O
timeout 100 tools/bin/resblock_lab 32 5 > $O/resblock_lab_after.log 2>&1; grep -v "tile \|ELU(x)" $O/resblock_lab_after.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "resblock or gemm" 2>&1 | tail -2
python tools/codec_bench.py 256 30 2>&1 | grep "B=" | tee $O/codec256.log
timeout 900 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -2
