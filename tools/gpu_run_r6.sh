#!/bin/bash
# Round-6 GPU session driver (round 5's sections + att, direct3x3, splitk, forceparts with bf16 buckets) (everything lands under gpurun_out/<tag>/; summaries are copied to profiles/ by hand).
#   gpurun -- 'bash tools/gpu_run_r6.sh <tag> <section> [<section> ...]'
# sections: panel newtests alltests smoke bench benchq config5 train trainpw ipc exact frame prof proftrain pmc
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
WHAT="$*"
has() { [[ " $WHAT " == *" $1 "* ]]; }
line() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        t = json.loads([l for l in open(f).read().splitlines() if l.startswith('{"metric"')][-1])
        ex = t.get("extras") or {}
        print(f.split("/")[-1], "ms/step", round(t["ms_per_step"], 3), "stages", t.get("stages_ms"), "roof", round((t.get("roofline") or {}).get("frac") or 0, 4),
              "head_ms", (t.get("roofline") or {}).get("avg_launch_ms"), "graph", t.get("train_graph"), "parity", (t.get("parity_rel_err") or {}).get("worst_of_all_outputs"),
              "extras", {k: (v.get("ms_per_step"), v.get("error")) for k, v in ex.items()})
    except Exception as e:
        print(f, "ERR", e)
PY
}
if has newtests; then
  timeout 1500 python -m pytest -q -m gpu -x tests/test_loss_kernels_gpu.py tests/test_losses_gpu.py tests/test_bf16_conv.py tests/test_gemm_x3.py tests/test_train_step.py ${EXTRA_TESTS:-} > $O/pytest_new.txt 2>&1; tail -15 $O/pytest_new.txt
fi
if has panel; then
  timeout 900 python -m pytest -q -m gpu -x tests/test_gemm_x3.py > $O/pytest_gemm.txt 2>&1; tail -8 $O/pytest_gemm.txt
  timeout 300 python tools/bench_gemm_panel.py > $O/gemm_panel.txt 2>&1; cat $O/gemm_panel.txt
  OCCDEPTH_GEMM_X3_PANEL=0 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/benchq_panel0.json 2> $O/benchq_panel0.err; line $O/benchq_panel0.json
fi
if has benchq; then
  timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/benchq.json 2> $O/benchq.err; line $O/benchq.json
fi
if has config5; then
  timeout 400 python bench.py --config 5 --steps 5 --warmup 2 > $O/config5.json 2> $O/config5.err; line $O/config5.json
  python - <<PY
import json
t = json.loads([l for l in open("$O/config5.json").read().splitlines() if l.startswith('{"metric"')][-1])
print(json.dumps(t["roofline"], indent=None)[:900]); print(t["top_conv_launches_ms"])
PY
fi
if has train; then
  timeout 500 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16.json 2> $O/train_bf16.err; line $O/train_bf16.json
  timeout 500 python bench.py --train --steps 5 --warmup 2 > $O/train_fp32.json 2> $O/train_fp32.err; line $O/train_fp32.json
  OCCDEPTH_LOSS_KERNELS=0 timeout 500 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16_aten_losses.json 2> $O/train_bf16_aten_losses.err; line $O/train_bf16_aten_losses.json
fi
if has trainpw; then
  OCCDEPTH_TRAIN_PW_GEMM=0 timeout 500 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16_pw0.json 2> $O/train_bf16_pw0.err; line $O/train_bf16_pw0.json
fi
if has ipc; then
  timeout 600 python -m pytest -q -m gpu -x tests/test_ipc_allreduce_gpu.py -s > $O/pytest_ipc.txt 2>&1; tail -12 $O/pytest_ipc.txt | cut -c1-400
  OCCDEPTH_FORCE_DIST=1 OCCDEPTH_TRAIN_GRAPH_DDP=1 timeout 500 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16_forced_ipc.json 2> $O/train_bf16_forced_ipc.err; line $O/train_bf16_forced_ipc.json
  OCCDEPTH_SYNCBN_IPC=0 OCCDEPTH_FORCE_DIST=1 OCCDEPTH_TRAIN_GRAPH_DDP=1 timeout 500 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16_forced_rccl.json 2> $O/train_bf16_forced_rccl.err; line $O/train_bf16_forced_rccl.json
  grep -o '"parallelism": "[^"]*"' $O/train_bf16_forced_ipc.json $O/train_bf16_forced_rccl.json
fi
if has forceparts; then
  for parts in bn buckets; do for ipc in 1 0; do
    OCCDEPTH_FORCE_PARTS=$parts OCCDEPTH_SYNCBN_IPC=$ipc OCCDEPTH_FORCE_DIST=1 OCCDEPTH_TRAIN_GRAPH_DDP=1 timeout 500 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16_forced_${parts}_ipc$ipc.json 2> $O/train_bf16_forced_${parts}_ipc$ipc.err; line $O/train_bf16_forced_${parts}_ipc$ipc.json
  done; done
  timeout 500 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16_plain.json 2> $O/train_bf16_plain.err; line $O/train_bf16_plain.json
  python - > $O/forced_decomposition.txt <<PY
import json
def ms(f):
    try:
        return json.loads([l for l in open("$O/" + f).read().splitlines() if l.startswith('{"metric"')][-1])["ms_per_step"]
    except Exception as e:
        return float("nan")
plain = ms("train_bf16_plain.json")
print("# config-2 training step, bf16-MFMA mode, one MI355X, 5 timed steps each (bench.py --train --bf16; OCCDEPTH_FORCE_DIST=1 OCCDEPTH_TRAIN_GRAPH_DDP=1")
print("# drives SyncBatchNorm + gradient buckets through a single-rank RCCL group; OCCDEPTH_FORCE_PARTS picks which exchanges run)")
print("plain step (no exchanges)                                   %8.2f ms" % plain)
for parts, what in (("bn", "SyncBatchNorm exchanges only"), ("buckets", "gradient buckets only (600 MB all-reduced onto itself)")):
    for ipc, how in ((1, "peer-memory path (csrc/ipc_allreduce.hip, in-kernel exchange)"), (0, "process group (RCCL)")):
        v = ms("train_bf16_forced_%s_ipc%d.json" % (parts, ipc))
        print("%-34s %-62s %8.2f ms  %+5.1f %%" % (what[:34], how, v, 100.0 * (v / plain - 1.0)))
for f, how in (("train_bf16_forced_ipc.json", "both, peer-memory SyncBatchNorm"), ("train_bf16_forced_rccl.json", "both, RCCL SyncBatchNorm")):
    v = ms(f)
    print("%-34s %-62s %8.2f ms  %+5.1f %%" % ("SyncBatchNorm + buckets", how, v, 100.0 * (v / plain - 1.0)))
PY
  cat $O/forced_decomposition.txt
fi
if has direct3x3; then
  timeout 400 python tools/bench_direct3x3.py > $O/direct3x3.txt 2>&1; grep -v "^/opt" $O/direct3x3.txt | cut -c1-330
fi
if has splitk; then
  timeout 300 python tools/bench_splitk.py > $O/gemm_splitk.txt 2>&1; grep "best\|K16" $O/gemm_splitk.txt
fi
if has commbf16; then
  OCCDEPTH_GRAD_COMM=bf16 OCCDEPTH_FORCE_PARTS=buckets OCCDEPTH_FORCE_DIST=1 OCCDEPTH_TRAIN_GRAPH_DDP=1 timeout 500 python bench.py --train --bf16 --steps 5 --warmup 2 > $O/train_bf16_forced_buckets_commbf16.json 2> $O/train_bf16_forced_buckets_commbf16.err; line $O/train_bf16_forced_buckets_commbf16.json
fi
if has headab; then
  for e in 1 0; do OCCD_C32X3_RES_EARLY=$e timeout 300 python tools/bench_head_x3.py > $O/head_x3_res_early$e.txt 2>&1; grep "nres=[12] K2s3\|nres=0 K2s3" $O/head_x3_res_early$e.txt | cut -c1-120; done
fi
if has exact; then
  OCCDEPTH_BF16X3=0 OCCDEPTH_GEMM_X3=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_exact_fp32.json 2> $O/bench_exact_fp32.err; line $O/bench_exact_fp32.json
fi
if has bench; then
  timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; line $O/bench.json
fi
if has frame; then
  timeout 300 python tools/frame_table.py > $O/frame_per_launch.txt 2> $O/frame_per_launch.err; tail -2 $O/frame_per_launch.txt | cut -c1-400
fi
if has alltests; then
  timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
fi
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  OCCDEPTH_BENCH_ISOLATED=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-extras > $O/bench_under_rocprof.json 2> /tmp/prof_bench.err
  f=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
  python $R/tools/summarize_trace.py $f $O/steady_state_kernel_stats.csv 5 > /dev/null; head -12 $O/steady_state_kernel_stats.csv | cut -c1-140
fi
if has proftrain; then
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python $R/bench.py --train --bf16 --steps 3 --warmup 2 > $O/train_bf16_under_rocprof.json 2> /tmp/prof_train.err
  f=$(ls /tmp/prof_train/*/*kernel_trace.csv | head -1)
  python $R/tools/summarize_trace.py $f $O/train_step_bf16_kernels.csv train > /dev/null 2>&1; head -8 $O/train_step_bf16_kernels.csv | cut -c1-140
  cd $R && timeout 500 python tools/prof_train_aten.py bf16 > $O/train_step_bf16_aten_ops.txt 2>&1; cd /tmp
fi
if has att; then
  # the instruction-level trace DESIGN section 10 named as K16's next step; needs the trace decoder library (not in this image:
  # the run is kept as evidence of what the tool answers here)
  timeout 300 rocprofv3 --att --att-target-cu 1 --kernel-trace -d /tmp/att_gemm -- python $R/tools/att_gemm.py > $O/att_gemm.txt 2>&1
  echo "rc=$?" >> $O/att_gemm.txt; (ls -R /tmp/att_gemm 2>/dev/null | head -40) >> $O/att_gemm.txt; tail -15 $O/att_gemm.txt | cut -c1-200
fi
if has pmc; then
  for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" FETCH_SIZE WRITE_SIZE; do
    n=$(echo $c | cut -d' ' -f1)
    OCCDEPTH_BENCH_ISOLATED=0 OCCDEPTH_GRAPH_ALL=0 OCCDEPTH_GRAPH_2D=0 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcf_$n -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras > /tmp/pmcf_$n.log 2>&1
    cp $(ls /tmp/pmcf_$n/*/*counter_collection.csv | head -1) /tmp/pmcf_$n.csv
  done
  python $R/tools/pmc_frame.py /tmp/pmcf_GRBM_GUI_ACTIVE.csv /tmp/pmcf_FETCH_SIZE.csv /tmp/pmcf_WRITE_SIZE.csv > $O/pmc_frame.txt 2>&1; cat $O/pmc_frame.txt | cut -c1-130
fi
echo "[gpu_run_r6 $TAG done: $WHAT]"
