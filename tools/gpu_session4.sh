#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_pw_gemm.py -q -m gpu -x > $O/pytest_new.txt 2>&1; tail -3 $O/pytest_new.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -s -k "config2_benched or small" > $O/pytest_parity.txt 2>&1; grep "relative errors\|HIP vs f\|passed\|failed\|Error" $O/pytest_parity.txt | tail -8
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['stages_ms'], d.get('parity_rel_err'))"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity > /tmp/b.json 2> /tmp/prof_bench.err
f=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
python $R/tools/summarize_trace.py $f $O/steady_state_kernel_stats.csv 5 > /dev/null; head -40 $O/steady_state_kernel_stats.csv | cut -c1-150
