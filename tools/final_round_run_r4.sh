#!/bin/bash
# Round-4 evidence run (everything lands under gpurun_out/r4ev/; summaries are copied to profiles/ by hand)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4ev
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
WHAT="${*:-bench frame prof pmc exact tests}"
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has bench; then
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
fi
if has frame; then
  timeout 300 python tools/frame_table.py > $O/frame_per_launch.txt 2> $O/frame_per_launch.err; tail -2 $O/frame_per_launch.txt | cut -c1-300
fi
if has exact; then
  OCCDEPTH_BF16X3=0 OCCDEPTH_GEMM_X3=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_exact_fp32.json 2> $O/bench_exact_fp32.err
  OCCDEPTH_BF16X3=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_exact_head.json 2> $O/bench_exact_head.err
  OCCDEPTH_FORCE_DIST=1 OCCDEPTH_TRAIN_GRAPH_DDP=1 timeout 400 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16_forced_rccl.json 2> $O/train_bf16_forced_rccl.err
  python - <<PY
import json
for f in ("bench_exact_fp32", "bench_exact_head", "train_bf16_forced_rccl"):
    try:
        t = json.loads([l for l in open("$O/%s.json" % f).read().splitlines() if l.startswith('{"metric"')][0])
        print(f, round(t["ms_per_step"], 2), t.get("stages_ms"), t.get("roofline", {}).get("frac"), t.get("train_graph"))
    except Exception as e:
        print(f, "ERR", e)
PY
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-extras > $O/bench_under_rocprof.json 2> /tmp/prof_bench.err
  f=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
  python $R/tools/summarize_trace.py $f $O/steady_state_kernel_stats.csv 5 > /dev/null; head -8 $O/steady_state_kernel_stats.csv | cut -c1-140
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python $R/bench.py --train --bf16 --steps 3 --warmup 2 > $O/train_bf16_under_rocprof.json 2> /tmp/prof_train.err
  f=$(ls /tmp/prof_train/*/*kernel_trace.csv | head -1)
  python $R/tools/summarize_trace.py $f $O/train_step_bf16_kernels.csv train > /dev/null 2>&1; head -6 $O/train_step_bf16_kernels.csv | cut -c1-140
fi
if has pmc; then
  for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" FETCH_SIZE WRITE_SIZE; do
    n=$(echo $c | cut -d' ' -f1)
    OCCDEPTH_GRAPH_ALL=0 OCCDEPTH_GRAPH_2D=0 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcf_$n -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras > /tmp/pmcf_$n.log 2>&1
    cp $(ls /tmp/pmcf_$n/*/*counter_collection.csv | head -1) /tmp/pmcf_$n.csv
  done
  python $R/tools/pmc_frame.py /tmp/pmcf_GRBM_GUI_ACTIVE.csv /tmp/pmcf_FETCH_SIZE.csv /tmp/pmcf_WRITE_SIZE.csv > $O/pmc_frame.txt 2>&1; cat $O/pmc_frame.txt | cut -c1-130
fi
cd $R
if has tests; then
  timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
fi
