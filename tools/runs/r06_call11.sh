#!/bin/bash
# round 6, call 11: the A/B of the codec fix up to the >= 200 vs >= 200 processes VERDICT r5 asked for (round 5's code path: 151 so far; fix: 205)
O=gpurun_out/r6c11; mkdir -p $O
timeout 2400 python tools/race_trials.py ${1:-55} off:SSRHIP_POISON_ALLOC=1,SSRHIP_CODEC_PRESIZE=0,SSRHIP_RECORD_STREAM=1,rounds=3 on:SSRHIP_POISON_ALLOC=1,rounds=3 2>&1 | grep -v amdgpu.ids | tee $O/race_trials_ab4.log | grep -v "^            item\|^    FAIL\|^        " | tail -14
timeout 600 python -m pytest tests/test_gpu_configs.py -q -k "config1 or config2" 2>&1 | tail -3 | tee $O/pytest_config12.log
