#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s27; mkdir -p $O; cd $R
timeout 300 python tools/bench_decoder_levels.py 2>&1 | grep "1/\|totals" | tee $O/decoder_levels.txt
