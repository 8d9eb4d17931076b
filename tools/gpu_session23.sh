#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s23
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /tmp/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /tmp/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /tmp/pmc_w.log 2>&1
s=$(ls /tmp/pmc_sq/*/*counter_collection.csv | head -1); f=$(ls /tmp/pmc_f/*/*counter_collection.csv | head -1); w=$(ls /tmp/pmc_w/*/*counter_collection.csv | head -1)
python $R/tools/pmc_frame.py $s $f $w | tee $O/pmc_frame.txt
tail -2 /tmp/pmc_sq.log | cut -c1-300
