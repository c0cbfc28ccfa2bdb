#!/bin/bash
# round 6, call 6: the three tests that were red in the first full run, then the final bench lines / traces / PMC passes
O=gpurun_out/r6c6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair_guard.py "tests/test_gpu_configs.py::test_config2_830m_sampled_top_k40_top_p08_matches_oracle" tests/test_gpu_codec.py -q -k "pair or squat or engine or process or config2 or sized or concurrent_streams" 2>&1 | tail -15 | cut -c1-220 | tee $O/pytest_three.log
bash tools/runs/r06_final.sh all 2>&1 | tail -30
