# HBM traffic of the codec kernels (separate --pmc passes), same command as profiles/r02_codec_b32_kernel_trace_summary.md
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pc_$c -- python $R/tools/codec_bench.py 32 30 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/codec_bench.py 32 30 > /dev/null 2>&1
cd $R
for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py $(ls $O/pc_$c/*/*counter_collection.csv | head -1) $O/r02_codec_b32_pmc_$c.md > /dev/null; done
python tools/prof_summary.py $(ls $O/kt/*/*kernel_trace.csv | head -1) $O/r02_codec_b32_kernel_trace_summary.md > /dev/null
rm -rf $O/pc_* $O/kt
