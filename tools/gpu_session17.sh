#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s17
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; python -c "
import json,sys; d=json.loads(open('$O/bench_new.json').read().strip().splitlines()[-1]); print('new', d['value'], d['stages_ms'], d['parity_rel_err'], d['roofline']['frac'], d['stack3d'], d['lift'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_under_rocprof.json 2> /tmp/prof_bench.err
f=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
python $R/tools/summarize_trace.py $f $O/steady_state_kernel_stats.csv 5 > /dev/null; head -24 $O/steady_state_kernel_stats.csv | cut -c1-150
