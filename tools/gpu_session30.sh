#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s30; mkdir -p $O; cd $R
timeout 300 python tools/bench_crp_hints.py 2>&1 | grep -- "->" | tee $O/crp_hints.txt
