#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s19
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winograd2d.py tests/test_train_step.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 900 python tools/bench_train.py 3 kitti_a100 > $O/train_fp32.txt 2>&1; grep "train step" $O/train_fp32.txt
OCCDEPTH_TRAIN_FUSED_ACT=0 timeout 900 python tools/bench_train.py 3 kitti_a100 > $O/train_fp32_aten_act.txt 2>&1; grep "train step" $O/train_fp32_aten_act.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python $R/tools/bench_train.py 2 kitti_a100 > /tmp/t.log 2>&1
f=$(ls /tmp/prof_train/*/*kernel_trace.csv | head -1)
python $R/tools/summarize_trace.py $f $O/train_fp32_kernels.csv train > /dev/null; head -40 $O/train_fp32_kernels.csv | cut -c1-150
