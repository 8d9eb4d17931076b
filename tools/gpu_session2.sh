#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2
mkdir -p $O
cd $R
timeout 400 python tools/exp_wino.py > $O/exp_wino.txt 2>&1; tail -13 $O/exp_wino.txt
