R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5z; mkdir -p $O
python bench.py --utts 8 --no-extras --no-cpu-baseline > $O/r05_bench_n1_8utts.json 2> $O/bench8.err; echo "bench8 rc=$?"
bash tools/runs/r05_final.sh prof16
bash tools/runs/r05_final.sh codec
