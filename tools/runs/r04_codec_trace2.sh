# round 4, after the epilogue rework: kernel trace of config 5 (256 clips x 30 s) with per-launch GEMM shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4j; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -f $O/gemm.log
SSRHIP_GEMM_LOG=$O/gemm.log timeout 110 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktc -- python $R/tools/codec_bench.py 256 30 > $O/codec256.txt 2>&1
cd $R
python tools/prof_summary.py $(ls $O/ktc/*/*kernel_trace.csv | head -1) $O/r04_codec_b256_kernel_trace_summary.md --gemm-log $O/gemm.log > /dev/null
rm -rf $O/ktc; grep "B=" $O/codec256.txt; head -12 $O/r04_codec_b256_kernel_trace_summary.md | cut -c1-200
