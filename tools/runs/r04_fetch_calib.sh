# round 4: FETCH_SIZE of a kernel with KNOWN traffic in the residual blocks' access pattern
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fc -- $R/tools/bin/fetch_calib > $O/fetch_calib.txt 2>&1
python3 - $(ls $O/fc/*/*counter_collection.csv | head -1) >> $O/fetch_calib.txt <<'P'
import csv, sys
from collections import defaultdict
agg = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg["tile_read_twice" if "Lb1" in r["Kernel_Name"] or "<true>" in r["Kernel_Name"] else "tile_read"].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:16s} FETCH_SIZE raw {sum(v) / len(v) * 1024 / 1e6:10.1f} MB per launch ({len(v)} launches)")
P
rm -rf $O/fc; grep -v rocprofv3 $O/fetch_calib.txt
