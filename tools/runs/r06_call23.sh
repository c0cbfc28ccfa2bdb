#!/bin/bash
# round 6, call 23: the multi-stream codec test failed in round 10 of the seventh full run (one item's last LSTM frames, 1e-3): how often?
# 24 stand-alone runs (12 rounds each) with the codec's counters in the message, then the whole suite once more
O=gpurun_out/r6c23; mkdir -p $O
for i in $(seq 1 24); do timeout 300 python -m pytest tests/test_gpu_codec.py -q -p no:cacheprovider -k "concurrent_streams" 2>&1 | grep -E "passed|failed|AssertionError" | cut -c1-400; done | tee $O/multistream_x24.log
grep -c passed $O/multistream_x24.log
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=15 -rs > $O/pytest_gpu_full.log 2>&1; echo "suite rc=$?"; grep -n "passed\|failed\|AssertionError: round" $O/pytest_gpu_full.log | tail -3 | cut -c1-400
