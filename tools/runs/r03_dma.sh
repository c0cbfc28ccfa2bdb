# round 3: the 8-wave DMA split kernel in the product: GEMM / codec parity tests, then codec throughput at 256 and 32 clips x 30 s with the
# 4-wave kernel (SSRHIP_GEMM_SPLIT_DMA=0) and the new one alternating on the same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3x; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gemm or split" 2>&1 | tail -3 > $O/tests.log
timeout 700 python -m pytest tests/test_gpu_codec.py tests/test_gpu_ragged.py -q -m gpu -x 2>&1 | tail -3 >> $O/tests.log
for i in 1 2; do
  SSRHIP_GEMM_SPLIT_DMA=0 python tools/codec_bench.py 256 30 2>&1 | tail -1 | sed 's/^/4-wave  /' >> $O/codec.log
  python tools/codec_bench.py 256 30 2>&1 | tail -1 | sed 's/^/dma     /' >> $O/codec.log
done
SSRHIP_GEMM_SPLIT_DMA=0 python tools/codec_bench.py 32 30 2>&1 | tail -1 | sed 's/^/4-wave  /' >> $O/codec.log
python tools/codec_bench.py 32 30 2>&1 | tail -1 | sed 's/^/dma     /' >> $O/codec.log
cat $O/tests.log $O/codec.log
