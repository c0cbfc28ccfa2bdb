# round 5, GPU call 1: the prepared validations of code that has never executed + the two codec labs.
# Every part under its own timeout, logs written as they go (gpurun_out/r5a survives a later part's fault).
O=gpurun_out/r5a; mkdir -p $O
bash tools/runs/r05_labs.sh labs > $O/part_labs.log 2>&1
bash tools/runs/r05_labs.sh lstm > $O/part_lstm.log 2>&1
SSRHIP_EPILOGUE_TM=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -3 | tee $O/tm_gemm_test.log
SSRHIP_EPILOGUE_TM=1 timeout 600 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -3 | tee $O/tm_codec.log
SSRHIP_LSTM_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -3 | tee $O/lstm_split_codec.log
for v in "0 0" "1 0" "0 1" "1 1" "0 0"; do set -- $v
  echo "SSRHIP_EPILOGUE_TM=$1 SSRHIP_LSTM_SPLIT=$2" | tee -a $O/codec256_ab.log
  SSRHIP_EPILOGUE_TM=$1 SSRHIP_LSTM_SPLIT=$2 timeout 300 python tools/codec_bench.py 256 30 2>&1 | grep "B=" | tee -a $O/codec256_ab.log
done
tail -5 $O/part_lstm.log
