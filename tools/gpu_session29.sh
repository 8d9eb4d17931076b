#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s29; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_winograd2d.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -2 $O/pytest_a.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_1 -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity > /tmp/b_1.json 2> /tmp/prof_1.err
f=$(ls /tmp/prof_1/*/*kernel_trace.csv | head -1)
python $R/tools/summarize_trace.py $f $O/stats_fuse1.csv 5 > /dev/null; grep "upconv\|wino3x3\|GPU busy" $O/stats_fuse1.csv | cut -c1-150
python -c "
import json; d=json.loads(open('/tmp/b_1.json').read().strip().splitlines()[-1]); print(d['value'])"
