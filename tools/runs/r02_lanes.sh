# A/B of the codec's batch lanes and LSTM stream priority (tools/codec_bench.py), results in gpurun_out/r2m/lanes2.log
mkdir -p gpurun_out/r2m
run() { env "$@" timeout 300 python tools/codec_bench.py $B 30 2>&1 | tail -1 | sed "s/^/$* /" >> gpurun_out/r2m/lanes2.log; }
B=32
run SSRHIP_CODEC_LANES=1 SSRHIP_LSTM_PRIO=0
run SSRHIP_CODEC_LANES=1 SSRHIP_LSTM_PRIO=1
run SSRHIP_CODEC_LANES=2 SSRHIP_LSTM_PRIO=1
run SSRHIP_CODEC_LANES=4 SSRHIP_LSTM_PRIO=1
B=64
run SSRHIP_CODEC_LANES=2 SSRHIP_LSTM_PRIO=1
run SSRHIP_CODEC_LANES=4 SSRHIP_LSTM_PRIO=1
B=16
run SSRHIP_CODEC_LANES=2 SSRHIP_LSTM_PRIO=1
cat gpurun_out/r2m/lanes2.log
