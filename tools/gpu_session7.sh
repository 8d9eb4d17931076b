#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s7
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_pw_gemm.py tests/test_train_step.py -q -m gpu -x -k "depthwise or train_step" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python tools/bench_train.py 3 kitti_a100 > $O/train_fp32.txt 2>&1; grep "train step" $O/train_fp32.txt
timeout 900 python tools/bench_train.py 3 kitti_a100 bf16 > $O/train_bf16.txt 2>&1; grep "train step" $O/train_bf16.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python $R/tools/bench_train.py 2 kitti_a100 bf16 > /tmp/t.log 2>&1
f=$(ls /tmp/prof_train/*/*kernel_trace.csv | head -1)
python $R/tools/summarize_trace.py $f $O/train_bf16_kernels.csv 1 > /dev/null; head -45 $O/train_bf16_kernels.csv | cut -c1-130
