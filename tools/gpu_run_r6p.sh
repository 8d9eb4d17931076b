#!/bin/bash
# round 6, last call: the -m gpu suite on 4 xdist workers (one file per worker at a time), the two-process tests serially,
# then fresh-process repetitions of the captured five-step training test
O=gpurun_out/r6p; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout ${1:-235} python -m pytest tests -q -m gpu -n 4 --dist loadfile -p no:cacheprovider \
  --deselect tests/test_ipc_allreduce_gpu.py --deselect tests/test_syncbn_lightning_gpu.py > $O/pytest_xdist.txt 2>&1
echo "xdist rc=$? after $(( $(date +%s) - T0 )) s"; tail -4 $O/pytest_xdist.txt | cut -c1-300
grep -h "^FAILED\|^ERROR" $O/pytest_xdist.txt | head -10 | cut -c1-300
timeout ${2:-75} python -m pytest tests/test_ipc_allreduce_gpu.py tests/test_syncbn_lightning_gpu.py -q -m gpu > $O/pytest_two_process.txt 2>&1
echo "two-process rc=$? after $(( $(date +%s) - T0 )) s"; tail -2 $O/pytest_two_process.txt | cut -c1-300
bad=0; n=0
while [ $(( $(date +%s) - T0 )) -lt ${3:-330} ]; do
  n=$((n+1))
  timeout 60 python -m pytest tests/test_train_step.py -k whole_step_hipgraph -m gpu -x -q > $O/loop_$n.log 2>&1 || { bad=$((bad+1)); grep -h "nan\|assert" $O/loop_$n.log | head -3; }
done
echo "fresh-process loop: $bad bad of $n; total $(( $(date +%s) - T0 )) s" | tee $O/loop_summary.txt
