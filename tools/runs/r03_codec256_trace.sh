# round 3: kernel trace of wmencodec encode + decode at 256 clips x 30 s (config 5) with the split GEMM
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/codec_bench.py 256 30 > $O/codec256.txt 2>&1
cd $R
python tools/prof_summary.py $(ls $O/kt/*/*kernel_trace.csv | head -1) $O/r03_codec_b256_kernel_trace_summary.md > /dev/null
rm -rf $O/kt
cat $O/codec256.txt | tail -1; head -30 $O/r03_codec_b256_kernel_trace_summary.md
