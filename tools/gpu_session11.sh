#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s11
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_pw_gemm.py tests/test_train_step.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 200 python tools/bench_dw.py > $O/dw_layers.txt 2>&1; tail -15 $O/dw_layers.txt
timeout 300 python tools/bench_pw.py > $O/pw_layers.txt 2>&1; tail -29 $O/pw_layers.txt | cut -c1-130
timeout 300 python bench.py --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; python -c "
import json,sys; d=json.loads(open('$O/bench_new.json').read().strip().splitlines()[-1]); print('new', d['value'], d['stages_ms'], d['parity_rel_err'], d['roofline']['frac'], d['stack3d'])"
