# round 3, GPU box: RCCL 1-rank test, wmdecode timings / memory at 32 and 64 clips x 30 s, wmdecode kernel trace, the bench line
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_ragged.py -q -m gpu -k rccl 2>&1 | tail -3 > $O/rccl.log
python tools/codec_bench.py 32 30 wm 1 > $O/codec32_wm.txt 2>&1
python tools/codec_bench.py 64 30 wm 1 > $O/codec64_wm.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktw -- python $R/tools/codec_bench.py 32 30 wm 1 > /dev/null 2>&1
cd $R
python tools/prof_summary.py $(ls $O/ktw/*/*kernel_trace.csv | head -1) $O/r03_wmdecode_b32_kernel_trace_summary.md > /dev/null
rm -rf $O/ktw
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rccl.log
cat $O/rccl.log $O/codec32_wm.txt $O/codec64_wm.txt; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
