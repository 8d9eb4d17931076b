#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s9
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_pw_gemm.py tests/test_winograd2d.py -q -m gpu -x > $O/pytest_pw.txt 2>&1; tail -3 $O/pytest_pw.txt
timeout 200 python tools/bench_dw.py > $O/dw_layers.txt 2>&1; tail -16 $O/dw_layers.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; python -c "
import json,sys; d=json.loads(open('$O/bench_new.json').read().strip().splitlines()[-1]); print('new', d['value'], d['stages_ms'], d['parity_rel_err'])"
OCCDEPTH_MERGE_HEAD=0 OCCDEPTH_PW_EXPAND_LIB_BELOW=0 timeout 300 python bench.py --no-cpu-baseline --no-parity > $O/bench_allk11.json 2> $O/bench_allk11.err; python -c "
import json,sys; d=json.loads(open('$O/bench_allk11.json').read().strip().splitlines()[-1]); print('all-K11, unmerged', d['value'], d['stages_ms'])"
timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_under_rocprof.json 2> /tmp/prof_bench.err
f=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
python $R/tools/summarize_trace.py $f $O/steady_state_kernel_stats.csv 5 > /dev/null; head -45 $O/steady_state_kernel_stats.csv | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python $R/tools/bench_train.py 2 kitti_a100 > /tmp/t.log 2>&1
f=$(ls /tmp/prof_train/*/*kernel_trace.csv | head -1)
python $R/tools/summarize_trace.py $f $O/train_fp32_kernels.csv train > /dev/null; head -60 $O/train_fp32_kernels.csv | cut -c1-150
grep "train step" /tmp/t.log
