#!/bin/bash
# round 6: the five-step captured-training test in FRESH processes, memset rewrite on (N_ON runs) and off (N_OFF runs)
N_ON=${1:-26}; N_OFF=${2:-16}
mkdir -p gpurun_out/r6m
run() {  # $1 = label, $2 = count, $3 = OCCDEPTH_GRAPH_FIX_MEMSETS
  bad=0
  for i in $(seq 1 $2); do
    OCCDEPTH_GRAPH_FIX_MEMSETS=$3 timeout 120 python -m pytest tests/test_train_step.py -k whole_step_hipgraph -m gpu -x -q > gpurun_out/r6m/$1_$i.log 2>&1
    if [ $? -ne 0 ]; then bad=$((bad+1)); grep -h "nan\|assert\|Error" gpurun_out/r6m/$1_$i.log | head -3; else rm -f gpurun_out/r6m/$1_$i.log; fi
  done
  echo "== $1 (OCCDEPTH_GRAPH_FIX_MEMSETS=$3): $bad bad of $2 fresh processes" | tee -a gpurun_out/r6m/summary.txt
}
date +%s > gpurun_out/r6m/t0
run off $N_OFF 0
run on $N_ON 1
echo "elapsed $(( $(date +%s) - $(cat gpurun_out/r6m/t0) )) s" | tee -a gpurun_out/r6m/summary.txt
