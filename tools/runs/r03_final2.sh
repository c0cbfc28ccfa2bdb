# round 3, last GPU call: the whole -m gpu suite with the 8-wave DMA split kernels, the codec kernel trace at 256 clips x 30 s, and the
# bench line's codec leg
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O
cd $R
timeout 440 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|rc=" $O/pytest.log | tail -4
cd /tmp; export TMPDIR=/tmp
timeout 110 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt256 -- python $R/tools/codec_bench.py 256 30 > $O/codec256.txt 2>&1
cd $R
python tools/prof_summary.py $(ls $O/kt256/*/*kernel_trace.csv | head -1) $O/r03_codec_b256_kernel_trace_summary.md > /dev/null
rm -rf $O/kt256
tail -1 $O/codec256.txt; head -16 $O/r03_codec_b256_kernel_trace_summary.md
timeout 170 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --legs codec256 > $O/bench_codec.json 2> $O/bench_codec.err; echo "bench rc=$?"; tail -c 1500 $O/bench_codec.json
