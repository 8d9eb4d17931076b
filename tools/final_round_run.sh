#!/bin/bash
# Round-end evidence run on the GPU box (everything lands under gpurun_out/final/; copy the summaries to profiles/).
#   gpurun --timeout 3000 -- 'bash tools/final_round_run.sh [tests] [bench] [prof] [pmc] [train] [micro]'     (default: all)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
WHAT="${*:-tests bench prof pmc train micro}"
has() { [[ " $WHAT " == *" $1 "* ]]; }

if has tests; then
  timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
fi
if has bench; then
  timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
  timeout 400 python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_config5.err
  OCCDEPTH_BF16X3=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err
  OCCDEPTH_LIFT_PROJ=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_lift_tables.json 2> $O/bench_lift_tables.err
fi
if has train; then
  timeout 400 python bench.py --train --steps 5 --warmup 2 > $O/train_fp32.json 2> $O/train_fp32.err
  timeout 400 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16.json 2> $O/train_bf16.err
  OCCDEPTH_FORCE_DIST=1 OCCDEPTH_TRAIN_GRAPH_DDP=1 timeout 400 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16_forced_rccl.json 2> $O/train_bf16_forced_rccl.err
  OCCDEPTH_FORCE_DIST=1 OCCDEPTH_TRAIN_GRAPH_DDP=1 timeout 400 python bench.py --train --steps 5 --warmup 2 > $O/train_fp32_forced_rccl.json 2> $O/train_fp32_forced_rccl.err
  python - <<PY
import json
for f in ("train_fp32", "train_bf16", "train_bf16_forced_rccl", "train_fp32_forced_rccl"):
    try:
        t = json.loads([l for l in open("$O/%s.json" % f).read().splitlines() if l.startswith('{"metric"')][0])
        print(f, round(t["ms_per_step"], 1), "ms/step,", "graph" if t.get("train_graph") else "eager", t.get("train_graph_error"))
    except Exception as e:
        print(f, "ERR", e)
PY
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_under_rocprof.json 2> /tmp/prof_bench.err
  f=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
  python $R/tools/summarize_trace.py $f $O/steady_state_kernel_stats.csv 5 > /dev/null; head -12 $O/steady_state_kernel_stats.csv | cut -c1-140
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python $R/bench.py --train --bf16 --steps 3 --warmup 2 > $O/train_bf16_under_rocprof.json 2> /tmp/prof_train.err
  f=$(ls /tmp/prof_train/*/*kernel_trace.csv | head -1)
  python $R/tools/summarize_trace.py $f $O/train_step_bf16_kernels.csv train > /dev/null 2>&1 || python $R/tools/agg_trace.py $f $O/train_step_bf16_kernels.csv 5 > /dev/null 2>&1
  head -14 $O/train_step_bf16_kernels.csv 2>/dev/null | cut -c1-140
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/tools/pmc_head.py > /tmp/pmc_$c.log 2>&1
    g=$(ls /tmp/pmc_$c/*/*counter_collection.csv | head -1)
    grep "Counter_Name\|conv3d_c32" $g > $O/pmc_head_$c.csv
  done
  python $R/tools/pmc_to_json.py $O/pmc_head_FETCH_SIZE.csv $O/pmc_head_WRITE_SIZE.csv $O/head_conv_hbm_bytes.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcl_$c -- python $R/tools/pmc_lift.py fused 2 > /tmp/pmcl_$c.log 2>&1
    g=$(ls /tmp/pmcl_$c/*/*counter_collection.csv | head -1)
    grep "Counter_Name\|lift_" $g > $O/pmc_lift_proj_$c.csv
  done
  python - <<PY
import csv
for c, mul in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open("$O/pmc_lift_proj_%s.csv" % c)) if r["Counter_Name"] == c]
    print("fused lift %s: %.1f MB per launch (KB counter%s)" % (c, sum(v) / len(v) * mul / 1e6, ", doubled per the gfx950 correction" if mul > 1024 else ""))
PY
  # in-frame family table (eager frame: every launch is its own dispatch)
  for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" FETCH_SIZE WRITE_SIZE; do
    n=$(echo $c | cut -d' ' -f1)
    OCCDEPTH_GRAPH_ALL=0 OCCDEPTH_GRAPH_2D=0 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcf_$n -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > /tmp/pmcf_$n.log 2>&1
    cp $(ls /tmp/pmcf_$n/*/*counter_collection.csv | head -1) /tmp/pmcf_$n.csv
  done
  python $R/tools/pmc_frame.py /tmp/pmcf_GRBM_GUI_ACTIVE.csv /tmp/pmcf_FETCH_SIZE.csv /tmp/pmcf_WRITE_SIZE.csv > $O/pmc_frame.txt 2>&1; cat $O/pmc_frame.txt | cut -c1-130
fi
cd $R
if has micro; then
  timeout 300 python tools/bench_kernels.py head stack lift > $O/bench_kernels.txt 2>&1; grep "^conv\|UNet3D\|sfa_lift\|fused\|tables" $O/bench_kernels.txt
  timeout 400 python tools/bench_bf16.py fwd wgrad > $O/bench_bf16.txt 2>&1; grep -c . $O/bench_bf16.txt
  timeout 400 python tools/bf16_layer_errors.py > $O/bf16_layer_errors.txt 2>&1; tail -2 $O/bf16_layer_errors.txt
  timeout 600 python -m pytest tests/test_stack3d_backward.py tests/test_net2d_backward.py tests/test_bf16_conv.py -q -m gpu -s -k "bf16 or net2d" > $O/backward_chain_tests.txt 2>&1; grep -E "worst|flips|passed|failed|^bf16x3" $O/backward_chain_tests.txt | cut -c1-250
fi
