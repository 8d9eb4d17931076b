#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s16
mkdir -p $O
cd $R
timeout 300 python tools/debug_upconv.py nyu_small kitti_small 2>&1 | grep "level" | tee $O/dbg.txt
OCCD_UPCONV_DIRECT=1 timeout 300 python tools/debug_upconv.py nyu_small 2>&1 | grep "level" | tee $O/dbg_direct.txt
