#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s18
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winograd2d.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
for fold in 0 4000 20000 100000; do
OCCDEPTH_UPCONV_FOLD_BELOW=$fold timeout 300 python bench.py --no-cpu-baseline --no-parity > $O/bench_$fold.json 2> $O/bench_$fold.err; python -c "
import json,sys; d=json.loads(open('$O/bench_$fold.json').read().strip().splitlines()[-1]); print('fold below $fold:', d['value'], d['stages_ms'])"
done
