# MFMA utilisation (rocprofv3 derived counter MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs)), own --pmc pass
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2j; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mu_codec -- python $R/tools/codec_bench.py 32 30 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mu_dec8 -- python $R/bench.py --steps 40 --warmup 5 --utts 8 --no-extras --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $(ls $O/mu_codec/*/*counter_collection.csv | head -1) $O/r02_mfma_util_codec_b32.md
python tools/pmc_summary.py $(ls $O/mu_dec8/*/*counter_collection.csv | head -1) $O/r02_mfma_util_decode16rows.md
rm -rf $O/mu_codec $O/mu_dec8
