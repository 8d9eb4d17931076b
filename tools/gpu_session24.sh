#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s24
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_pw_gemm.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -2 $O/pytest_a.txt
timeout 300 python tools/bench_pw.py > $O/pw_layers.txt 2>&1; grep -- "->" $O/pw_layers.txt | awk '{print}' | cut -c1-60,105-300
