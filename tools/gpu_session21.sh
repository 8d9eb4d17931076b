#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s21
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_train_step.py tests/test_stack3d_backward.py tests/test_conv_grad.py -q -m gpu -x -s > $O/pytest_a.txt 2>&1; grep "eager\|passed\|failed\|Error" $O/pytest_a.txt | tail -5
timeout 900 python bench.py --train --steps 5 --warmup 2 > $O/bench_train.json 2> $O/bench_train.err; tail -c 900 $O/bench_train.json
OCCDEPTH_TRAIN_GRAPH=0 timeout 900 python bench.py --train --steps 5 --warmup 2 > $O/bench_train_eager.json 2> $O/bench_train_eager.err; tail -c 500 $O/bench_train_eager.json
