#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s22
mkdir -p $O
cd $R
timeout 300 python tools/debug_eval_syncs.py 2>&1 | grep "SYNC\|scan done" | tee $O/eval_syncs.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json,sys; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['parity_rel_err'])"
timeout 900 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/bench_train_bf16.json 2> $O/bench_train_bf16.err; python -c "
import json,sys; d=json.loads(open('$O/bench_train_bf16.json').read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'], d['train_graph'], d['train_graph_error'], d['loss'])"
