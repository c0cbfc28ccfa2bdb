# A/B of two builds of libssrhip.so on one box with tools/codec_bench.py (tools/bin/libssrhip_prev.so vs the in-tree build)
L=ssr-speech_amd/csrc/libssrhip.so
B=${1:-32}
cp $L /tmp/new.so
python tools/codec_bench.py $B 30 2>&1 | tail -1 | sed 's/^/new  /'
cp tools/bin/libssrhip_prev.so $L; python tools/codec_bench.py $B 30 2>&1 | tail -1 | sed 's/^/prev /'
cp /tmp/new.so $L
python tools/codec_bench.py $B 30 2>&1 | tail -1 | sed 's/^/new  /'
