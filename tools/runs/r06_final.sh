# round 6, GPU box: bench lines (1 and 8 utterances per GPU), rocprofv3 kernel traces and PMC passes of the FINAL build. `bash tools/runs/r06_final.sh [bench|prof1|prof16|codec]`
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6z; mkdir -p $O
cd $R
part=${1:-all}
if [ "$part" = all ] || [ "$part" = bench ]; then
python bench.py > $O/r06_bench_n1.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --utts 8 --no-extras --no-cpu-baseline > $O/r06_bench_n1_8utts.json 2> $O/bench8.err; echo "bench8 rc=$?"
fi
cd /tmp; export TMPDIR=/tmp
if [ "$part" = all ] || [ "$part" = prof1 ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python $R/bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1_$c -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline --no-ctx700 > /dev/null 2>&1
done
cd $R
python tools/prof_summary.py $(ls $O/kt1/*/*kernel_trace.csv | head -1) $O/r06_decode_kernel_trace_summary.md > /dev/null
cp $(ls $O/kt1/*/*kernel_stats.csv | head -1) $O/r06_rocprofv3_kernel_stats.csv
python tools/pmc_summary.py $(ls $O/pmc1_FETCH_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_fetch_size.md > /dev/null
python tools/pmc_summary.py $(ls $O/pmc1_WRITE_SIZE/*/*counter_collection.csv | head -1) $O/r06_pmc_write_size.md > /dev/null
rm -rf $O/kt1 $O/pmc1_*
cd /tmp
fi
if [ "$part" = all ] || [ "$part" = prof16 ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt8 -- python $R/bench.py --steps 200 --warmup 20 --utts 8 --no-extras --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/prof_summary.py $(ls $O/kt8/*/*kernel_trace.csv | head -1) $O/r06_decode16rows_kernel_trace_summary.md > /dev/null
cp $(ls $O/kt8/*/*kernel_stats.csv | head -1) $O/r06_rocprofv3_kernel_stats_16rows.csv
rm -rf $O/kt8
cd /tmp
fi
if [ "$part" = all ] || [ "$part" = codec ]; then
rm -f $O/gemm.log
SSRHIP_GEMM_LOG=$O/gemm.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktc -- python $R/tools/codec_bench.py 256 30 > $O/codec256.txt 2>&1
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mu_codec -- python $R/tools/codec_bench.py 256 30 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fs_codec -- python $R/tools/codec_bench.py 256 30 > /dev/null 2>&1
cd $R
python tools/prof_summary.py $(ls $O/ktc/*/*kernel_trace.csv | head -1) $O/r06_codec_b256_kernel_trace_summary.md --gemm-log $O/gemm.log > /dev/null
python tools/pmc_summary.py $(ls $O/mu_codec/*/*counter_collection.csv | head -1) $O/r06_mfma_util_codec_b256.md > /dev/null
python tools/pmc_summary.py $(ls $O/fs_codec/*/*counter_collection.csv | head -1) $O/r06_codec_b256_pmc_fetch_size.md > /dev/null
rm -rf $O/ktc $O/mu_codec $O/fs_codec
fi
ls -la $O
