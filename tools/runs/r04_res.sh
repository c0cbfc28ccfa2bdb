O=gpurun_out/r4f; mkdir -p $O
python tools/resblock_bench.py 32 | grep C= | tee $O/resblock_bench_final.log
python tools/resblock_bench.py 32 0 | grep C= | tee -a $O/resblock_bench_final.log
timeout 1200 python -m pytest tests/test_gpu_codec.py tests/test_gpu_ragged.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "resblock or gemm" 2>&1 | tail -2
rm -f $O/codec_ab3.log
for v in 0 1; do
    echo "SSRHIP_RESBLOCK_DMA=$v" | tee -a $O/codec_ab3.log
    SSRHIP_RESBLOCK_DMA=$v python tools/codec_bench.py 256 30 wm 4 2>&1 | grep "B=" | tee -a $O/codec_ab3.log
    SSRHIP_RESBLOCK_DMA=$v python tools/codec_bench.py 32 30 2>&1 | grep "B=" | tee -a $O/codec_ab3.log
done
