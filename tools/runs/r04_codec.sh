# round 4: residual blocks at C = 64 / 128 on resblock_split_dma_kernel (SSRHIP_RESBLOCK_DMA) vs the round-3 kernels: tests, then A/B on one box
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "resblock_split" 2>&1 | tail -4 | tee $O/pytest_resblock.log
timeout 900 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -4 | tee $O/pytest_codec.log
for rep in 1 2; do
  for v in 0 1; do
    echo "SSRHIP_RESBLOCK_DMA=$v" | tee -a $O/codec_ab.log
    SSRHIP_RESBLOCK_DMA=$v python tools/codec_bench.py 256 30 2>&1 | grep "B=" | tee -a $O/codec_ab.log
    SSRHIP_RESBLOCK_DMA=$v python tools/codec_bench.py 32 30 2>&1 | grep "B=" | tee -a $O/codec_ab.log
  done
done
