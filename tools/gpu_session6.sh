#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s6
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_lift_autograd.py tests/test_train_step.py tests/test_pw_gemm.py -q -m gpu -x > $O/pytest_lift.txt 2>&1; tail -12 $O/pytest_lift.txt
python tools/pmc_lift.py 2 2>&1 | tail -2
timeout 900 python tools/bench_train.py 3 kitti_a100 > $O/train_fp32.txt 2>&1; grep "train step\|step [123]:" $O/train_fp32.txt
