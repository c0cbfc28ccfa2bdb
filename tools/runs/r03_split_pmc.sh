# round 3: where the split GEMM's cycles go (tools/gemm_split_lab prof: 4096^3, the product-shaped kernel and the 8-wave double-buffered
# one), rocprofv3 --pmc in separate passes (counters only, no trace domains besides the kernel trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y$1; mkdir -p $O; rm -f $O/split_pmc.txt
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- $R/tools/bin/gemm_split_lab prof $1 > $O/pmc$i.txt 2>&1
  f=$(ls $O/pmc$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> $O/split_pmc.txt <<'P'
import csv, sys
from collections import defaultdict
agg = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = "db" if "db_kernel" in n else ("dma" if "w8g_kernel" in n else ("split" if "gemm_split_kernel" in n else None))
    if k: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:6s} {c:36s} {sum(v)/len(v):16.1f}  ({len(v)} launches)")
P
  else tail -3 $O/pmc$i.txt >> $O/split_pmc.txt; fi
  rm -rf $O/pmc$i
done
cat $O/split_pmc.txt
