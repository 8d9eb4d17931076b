#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s25
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_winograd2d.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -2 $O/pytest_a.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json,sys; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['stages_ms'], d['parity_rel_err']['ssc_logit'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_under_rocprof.json 2> /tmp/prof_bench.err
f=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
python $R/tools/summarize_trace.py $f $O/steady_state_kernel_stats.csv 5 > /dev/null; grep "upconv\|GPU busy" $O/steady_state_kernel_stats.csv | cut -c1-150

