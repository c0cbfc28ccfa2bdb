# round 4: where the cycles of resblock_split_dma_kernel go (tools/resblock_bench.py 32: C = 128 / 64 at the config-5 shapes), ring of 4 vs 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O; rm -f $O/resblock_pmc.txt
cd $R
for ring in 2; do echo "ring $ring" | tee -a $O/resblock_pmc.txt; SSRHIP_RESBLOCK_RING=$ring python tools/resblock_bench.py 32 2>&1 | grep "C=" | tee -a $O/resblock_pmc.txt; done
cd /tmp; export TMPDIR=/tmp
for ring in 2; do
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  SSRHIP_RESBLOCK_RING=$ring timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python $R/tools/resblock_bench.py 32 > $O/pmc$i.txt 2>&1
  f=$(ls $O/pmc$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 - "$f" $ring >> $O/resblock_pmc.txt <<'P'
import csv, sys
from collections import defaultdict
agg = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "resblock_split_dma_kernel" in n:
        k = "C128" if "<128" in n else "C64"
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"ring{sys.argv[2]} {k:5s} {c:36s} {sum(v)/len(v):16.1f}  ({len(v)} launches)")
P
  else tail -3 $O/pmc$i.txt >> $O/resblock_pmc.txt; fi
  rm -rf $O/pmc$i
done
done
grep -v rocprofv3 $O/resblock_pmc.txt
