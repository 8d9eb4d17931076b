#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_w$i -- python $R/tools/pmc_wino.py 93 305 688 320 32 > /tmp/pmc_w$i.log 2>&1
  g=$(ls /tmp/pmc_w$i/*/*counter_collection.csv | head -1)
  grep "Counter_Name\|wino3x3" $g > $O/pmc_wino_$i.csv
done
python - <<PY
import csv, collections
for i in (1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("$O/pmc_wino_%d.csv" % i)):
        k = r["Kernel_Name"].split("wino3x3_kernel")[1][:12]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, c in sorted(acc.items()):
        print(k, {n: round(sum(v) / len(v) / 1e6, 2) for n, v in c.items()})
PY
