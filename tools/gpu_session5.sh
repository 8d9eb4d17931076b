#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s5
mkdir -p $O
cd $R
python tools/pmc_lift.py 0 1 2 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcl_$c -- python $R/tools/pmc_lift.py 0 1 2 > /tmp/pmcl_$c.log 2>&1
  g=$(ls /tmp/pmcl_$c/*/*counter_collection.csv | head -1)
  grep "Counter_Name\|lift_" $g > $O/pmc_lift_$c.csv
done
python - <<PY
import csv
v = [float(r["Counter_Value"]) for r in csv.DictReader(open("$O/pmc_lift_FETCH_SIZE.csv")) if r["Counter_Name"] == "FETCH_SIZE"]
for m in range(3):
    seg = v[m * 24: (m + 1) * 24]
    print("xcd_mode %d: FETCH %.1f MB per launch (x2-corrected)" % (m, sum(seg) / len(seg) * 2048.0 / 1e6))
PY
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "sfa or lift or small" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --no-parity > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['stages_ms'], d['lift'])"
