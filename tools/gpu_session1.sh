#!/bin/bash
# dev session: full GPU test pass, Winograd microbench, bench with / without the fused kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x -s > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 400 python tools/bench_wino.py > $O/bench_wino.txt 2>&1; tail -14 $O/bench_wino.txt
timeout 600 python bench.py > $O/bench_fused.json 2> $O/bench_fused.err; tail -c 1500 $O/bench_fused.json
OCCDEPTH_WINO_FUSED=0 timeout 400 python bench.py --no-cpu-baseline --no-parity > $O/bench_nofused.json 2> $O/bench_nofused.err; tail -c 700 $O/bench_nofused.json
