#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s26
mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-parity > $O/b_$tag.json 2> $O/b_$tag.err; python -c "
import json,sys; d=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value'],2), {k: round(v,2) for k,v in d['stages_ms'].items()})"; }
run base A=1
run k10_from_10000px OCCDEPTH_WINO_FUSED_MIN_PIXELS=10000
run k10_everywhere OCCDEPTH_WINO_FUSED_MIN_PIXELS=0
run expand_lib_below_4000 OCCDEPTH_PW_EXPAND_LIB_BELOW=4000
run expand_lib_below_60000 OCCDEPTH_PW_EXPAND_LIB_BELOW=60000
run base2 A=1
