#!/bin/bash
# round 5, last GPU session: kernel trace of config 5 on the final build, then the whole GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -f $O/gemm.log
SSRHIP_GEMM_LOG=$O/gemm.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktc -- python $R/tools/codec_bench.py 256 30 > $O/codec256.txt 2>&1
cd $R
python tools/prof_summary.py $(ls $O/ktc/*/*kernel_trace.csv | head -1) $O/r05_codec_b256_kernel_trace_summary.md --gemm-log $O/gemm.log > /dev/null
rm -rf $O/ktc
tail -3 $O/codec256.txt
timeout 2400 python -m pytest tests/ -q -m gpu --durations=15 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest_gpu_full.log
