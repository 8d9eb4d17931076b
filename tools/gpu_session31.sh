#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s31; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_pw_gemm.py tests/test_parity_gpu.py -q -m gpu -x -k "softmax or flosp or small" > $O/pytest_a.txt 2>&1; tail -2 $O/pytest_a.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json,sys; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['stages_ms'], d['parity_rel_err']['ssc_logit'])"
