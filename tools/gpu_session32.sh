#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s32; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json,sys; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['stages_ms'], d['parity_rel_err']['ssc_logit'])"
