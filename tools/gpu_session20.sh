#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s20
mkdir -p $O
cd $R
timeout 900 python tools/debug_train_graph.py > $O/dbg.txt 2>&1; grep "SYNC\|eager step\|capture ok\|graphed step\|Error\|error" $O/dbg.txt | head -40
