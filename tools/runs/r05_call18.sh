R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5z; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "mfma or rows or tiled or sixteen or gemvm" 2>&1 | tail -3
python bench.py --utts 8 --no-extras --no-cpu-baseline > $O/r05_bench_n1_8utts.json 2> $O/bench8.err; echo "bench8 rc=$?"
bash tools/runs/r05_final.sh prof16
