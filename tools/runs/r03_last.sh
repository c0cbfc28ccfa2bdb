# round 3, last refresh: kernel traces of the codec with the final kernels (split GEMM + split residual chain + ELU on store) at 256 and
# 32 clips x 30 s, wmdecode at 32 clips, and the matrix-core utilisation of the same kernels (own --pmc pass)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3z; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt256 -- python $R/tools/codec_bench.py 256 30 > $O/codec256.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktwm -- python $R/tools/codec_bench.py 32 30 wm > $O/wm32.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mu -- python $R/tools/codec_bench.py 32 30 > $O/mu32.txt 2>&1
cd $R
python tools/prof_summary.py $(ls $O/kt256/*/*kernel_trace.csv | head -1) $O/r03_codec_b256_kernel_trace_summary.md > /dev/null
python tools/prof_summary.py $(ls $O/ktwm/*/*kernel_trace.csv | head -1) $O/r03_wmdecode_b32_kernel_trace_summary.md > /dev/null
python tools/pmc_summary.py $(ls $O/mu/*/*counter_collection.csv | head -1) $O/r03_mfma_util_codec_b32.md
rm -rf $O/kt256 $O/ktwm $O/mu
tail -1 $O/codec256.txt; tail -3 $O/wm32.txt; head -14 $O/r03_codec_b256_kernel_trace_summary.md; head -16 $O/r03_mfma_util_codec_b32.md
