# round 5, first GPU session: the two labs on the shipped codec kernels (built in-tree: tools/bin travels with the snapshot)
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I ssr-speech_amd/csrc -I include tools/gemm_dma_lab.hip -o tools/bin/gemm_dma_lab
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I ssr-speech_amd/csrc -I include tools/resblock_lab.hip -o tools/bin/resblock_lab
O=gpurun_out/r5a; mkdir -p $O
timeout 150 tools/bin/gemm_dma_lab 32 3 > $O/gemm_dma_lab.log 2>&1; cat $O/gemm_dma_lab.log
timeout 60 tools/bin/resblock_lab 32 5 > $O/resblock_lab.log 2>&1; grep -v "tile \|ELU(x)" $O/resblock_lab.log
# the split LSTM step (csrc/lstm_split.hip, never run before): its kernel test, then the codec with the knob
SSRHIP_RUN_UNVALIDATED=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "lstm_split" 2>&1 | tail -3
SSRHIP_LSTM_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -3
for v in 0 1; do echo "SSRHIP_LSTM_SPLIT=$v"; SSRHIP_LSTM_SPLIT=$v python tools/codec_bench.py 256 30 2>&1 | grep "B=" | tee -a $O/codec256_lstm_split.log; done
# the transposed convolutions' time mask as a row predicate of the 16-byte epilogue (never run before): GEMM tests + codec fixtures + timing with the knob
SSRHIP_EPILOGUE_TM=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -3
SSRHIP_EPILOGUE_TM=1 timeout 600 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -3
for v in 0 1; do echo "SSRHIP_EPILOGUE_TM=$v"; SSRHIP_EPILOGUE_TM=$v python tools/codec_bench.py 256 30 2>&1 | grep "B=" | tee -a $O/codec256_tm.log; done
