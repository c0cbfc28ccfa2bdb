# round 5, first GPU sessions. Built in-tree beforehand (tools/bin travels with the snapshot):
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I ssr-speech_amd/csrc -I include tools/gemm_dma_lab.hip -o tools/bin/gemm_dma_lab
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I ssr-speech_amd/csrc -I include tools/resblock_lab.hip -o tools/bin/resblock_lab
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/lstm_split_lab.hip -Lssr-speech_amd/csrc -lssrhip -Wl,-rpath,'$ORIGIN/../../ssr-speech_amd/csrc' -o tools/bin/lstm_split_lab
# One part per gpurun call (a never-run kernel that faults must not take the labs' output with it):
#   bash tools/runs/r05_labs.sh labs    the two labs on the shipped codec kernels                                    (~1.5 min)
#   bash tools/runs/r05_labs.sh lstm    csrc/lstm_split.hip: its kernel test alone, under a short timeout            (~1.5 min)
#   bash tools/runs/r05_labs.sh lstm2   ... then the codec fixtures and config-5 timings with SSRHIP_LSTM_SPLIT=1    (~6 min)
#   bash tools/runs/r05_labs.sh tm      SSRHIP_EPILOGUE_TM=1: GEMM tests, codec fixtures, config-5 timings           (~6 min)
O=gpurun_out/r5a; mkdir -p $O
case "$1" in
labs)
  timeout 150 tools/bin/gemm_dma_lab 32 3 > $O/gemm_dma_lab.log 2>&1; cat $O/gemm_dma_lab.log
  timeout 60 tools/bin/resblock_lab 32 5 > $O/resblock_lab.log 2>&1; grep -v "tile \|ELU(x)" $O/resblock_lab.log ;;
lstm)
  timeout 60 tools/bin/lstm_split_lab 256 1024 200 2>&1 | tee $O/lstm_split_lab.log          # no torch: both paths timed + compared in seconds
  SSRHIP_RUN_UNVALIDATED=1 timeout 120 python -m pytest tests/test_gpu_kernels.py -x -q -k "lstm_split" 2>&1 | tail -15 | tee $O/lstm_split_test.log ;;
lstm2)
  SSRHIP_LSTM_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -3 | tee $O/lstm_split_codec.log
  for v in 0 1; do echo "SSRHIP_LSTM_SPLIT=$v" | tee -a $O/codec256_lstm_split.log; SSRHIP_LSTM_SPLIT=$v python tools/codec_bench.py 256 30 2>&1 | grep "B=" | tee -a $O/codec256_lstm_split.log; done ;;
tm)
  SSRHIP_EPILOGUE_TM=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -3 | tee $O/tm_gemm_test.log
  SSRHIP_EPILOGUE_TM=1 timeout 600 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -3 | tee $O/tm_codec.log
  for v in 0 1; do echo "SSRHIP_EPILOGUE_TM=$v" | tee -a $O/codec256_tm.log; SSRHIP_EPILOGUE_TM=$v python tools/codec_bench.py 256 30 2>&1 | grep "B=" | tee -a $O/codec256_tm.log; done ;;
*) echo "usage: bash tools/runs/r05_labs.sh labs|lstm|lstm2|tm" ;;
esac
