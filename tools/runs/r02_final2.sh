set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2z2; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lm.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -q -m gpu -k "not config5" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --utts 2 --no-extras --no-cpu-baseline --steps 200 > $O/bench2.json 2> $O/bench2.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python $R/bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1_$c -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python tools/prof_summary.py $(ls $O/kt1/*/*kernel_trace.csv | head -1) $O/r02_decode_kernel_trace_summary.md > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py $(ls $O/pmc1_$c/*/*counter_collection.csv | head -1) $O/r02_pmc1_$c.md > /dev/null; done
cp $(ls $O/kt1/*/*kernel_stats.csv | head -1) $O/r02_rocprofv3_kernel_stats.csv
rm -rf $O/kt1 $O/pmc1_*
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['us_per_launch'], r['other_kernels_us_per_launch'], d['rtf_10s_tts']['rtf'], d['dp64']['codec_tokens_per_s_per_gpu'])
"
