# round 4: 16-byte epilogues (resblock_split_dma_kernel, gemm_split_dma_kernel): lab timings + bit-identity of the two forms, the kernels'
# parity tests in both forms, the whole -m gpu suite, config-5 codec timings
O=gpurun_out/r4h; mkdir -p $O
timeout 100 tools/bin/resblock_lab 32 5 > $O/resblock_lab_wide.log 2>&1; grep -v "tile \|ELU(x)" $O/resblock_lab_wide.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "resblock or gemm" 2>&1 | tail -2
SSRHIP_EPILOGUE_WIDE=0 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "resblock or gemm" 2>&1 | tail -2
python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
python tools/codec_bench.py 256 30 2>&1 | grep "B=" | tee $O/codec256.log
python bench.py --steps 200 --no-cpu-baseline --legs codec256 > $O/bench_codec256.json 2> $O/bench.err; echo "bench rc=$?"
SSRHIP_EPILOGUE_WIDE=0 python tools/codec_bench.py 256 30 2>&1 | grep "B=" | tee $O/codec256_dword.log
