set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O
cd $R
python -m pytest tests -q -m gpu --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --utts 8 --no-extras --no-cpu-baseline > $O/bench8.json 2> $O/bench8.err; echo "bench8 rc=$?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python $R/bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt8 -- python $R/bench.py --steps 200 --warmup 20 --utts 8 --no-extras --no-cpu-baseline > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1_$c -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc8_$c -- python $R/bench.py --steps 40 --warmup 5 --utts 8 --no-extras --no-cpu-baseline > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktc -- python $R/tools/codec_bench.py 32 30 > $O/codec32.txt 2>&1
cd $R
python tools/prof_summary.py $(ls $O/kt1/*/*kernel_trace.csv | head -1) $O/r02_decode_kernel_trace_summary.md > /dev/null
python tools/prof_summary.py $(ls $O/kt8/*/*kernel_trace.csv | head -1) $O/r02_decode16rows_kernel_trace_summary.md > /dev/null
python tools/prof_summary.py $(ls $O/ktc/*/*kernel_trace.csv | head -1) $O/r02_codec_b32_kernel_trace_summary.md > /dev/null
for t in 1 8; do for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py $(ls $O/pmc${t}_$c/*/*counter_collection.csv | head -1) $O/r02_pmc${t}_$c.md > /dev/null; done; done
cp $(ls $O/kt1/*/*kernel_stats.csv | head -1) $O/r02_rocprofv3_kernel_stats.csv; cp $(ls $O/kt8/*/*kernel_stats.csv | head -1) $O/r02_rocprofv3_kernel_stats_16rows.csv
rm -rf $O/kt1 $O/kt8 $O/ktc $O/pmc1_* $O/pmc8_*
ls -la $O
