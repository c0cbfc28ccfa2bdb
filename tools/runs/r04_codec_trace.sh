# round 4: kernel trace of wmencodec encode + decode at 256 clips x 30 s (config 5), FINAL build, with the problem size and a per-launch
# TFLOP/s column for every GEMM launch (SSRHIP_GEMM_LOG joined by tools/prof_summary.py)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O; rm -f $O/gemm.log
SSRHIP_GEMM_LOG=$O/gemm.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/codec_bench.py 256 30 > $O/codec256.txt 2>&1
cd $R
python tools/prof_summary.py $(ls $O/kt/*/*kernel_trace.csv | head -1) $O/r04_codec_b256_kernel_trace_summary.md --gemm-log $O/gemm.log | head -3
rm -rf $O/kt
tail -1 $O/codec256.txt; head -34 $O/r04_codec_b256_kernel_trace_summary.md
