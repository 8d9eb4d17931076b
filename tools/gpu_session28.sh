#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s28; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_winograd2d.py tests/test_parity_gpu.py -q -m gpu -x -k "upconv or upsample_block or small or decoder or benched" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json,sys; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['stages_ms'], d['parity_rel_err']['ssc_logit'], d['parity_rel_err']['occ_logit'])"
OCCDEPTH_UPCONV_FUSE_SKIP=0 timeout 300 python bench.py --no-cpu-baseline --no-parity > $O/bench_nofuse.json 2> $O/bench_nofuse.err; python -c "
import json,sys; d=json.loads(open('$O/bench_nofuse.json').read().strip().splitlines()[-1]); print('no fuse', d['value'], d['stages_ms'])"
