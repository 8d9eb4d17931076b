#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s14
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winograd2d.py tests/test_train_step.py tests/test_pw_gemm.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 900 python tools/bench_train.py 3 kitti_a100 > $O/train_fp32.txt 2>&1; grep "train step" $O/train_fp32.txt
OCCDEPTH_TRAIN_K10=0 timeout 900 python tools/bench_train.py 3 kitti_a100 > $O/train_fp32_miopen.txt 2>&1; grep "train step" $O/train_fp32_miopen.txt
timeout 900 python tools/bench_train.py 3 kitti_a100 bf16 > $O/train_bf16.txt 2>&1; grep "train step" $O/train_bf16.txt
