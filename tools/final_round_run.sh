#!/bin/bash
# Round-end evidence run on the GPU box (everything lands under gpurun_out/final/; copy the summaries to profiles/).
#   gpurun --timeout 2400 -- 'bash tools/final_round_run.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_under_rocprof.json 2> /tmp/prof_bench.err
f=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
python $R/tools/summarize_trace.py $f $O/steady_state_kernel_stats.csv 5 > /dev/null; head -12 $O/steady_state_kernel_stats.csv | cut -c1-140
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/tools/pmc_head.py > /tmp/pmc_$c.log 2>&1
  g=$(ls /tmp/pmc_$c/*/*counter_collection.csv | head -1)
  grep "Counter_Name\|conv3d_c32" $g > $O/pmc_head_$c.csv
done
python $R/tools/pmc_to_json.py $O/pmc_head_FETCH_SIZE.csv $O/pmc_head_WRITE_SIZE.csv $O/head_conv_hbm_bytes.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcl_$c -- python $R/tools/pmc_lift.py > /tmp/pmcl_$c.log 2>&1
  g=$(ls /tmp/pmcl_$c/*/*counter_collection.csv | head -1)
  grep "Counter_Name\|lift_" $g > $O/pmc_lift_$c.csv
done
python - <<PY
import csv
for c, mul in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open("$O/pmc_lift_%s.csv" % c)) if r["Counter_Name"] == c]
    print("lift %s: %.1f MB per launch (KB counter%s)" % (c, sum(v) / len(v) * mul / 1e6, ", doubled per the gfx950 correction" if mul > 1024 else ""))
PY
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_sq -- python $R/tools/pmc_head.py > /tmp/pmc_sq.log 2>&1
g=$(ls /tmp/pmc_sq/*/*counter_collection.csv | head -1)
grep "Counter_Name\|conv3d_c32" $g > $O/pmc_head_sq.csv
python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$O/pmc_head_sq.csv")):
    d = r["Kernel_Name"].split("<")[1][0]
    acc[d][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[d]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for d, c in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
    print("head conv d=%s: %.3f ms under the profiler, %.2f GHz, MFMA pipe busy %.1f %% of SIMD-cycles, waves: wait %.0f %% / issue-stall %.0f %% / active %.0f %%, LDS bank-conflict cycles %.2f %% of wave cycles"
          % (d, m["ns"] / 1e6, cyc / m["ns"], 100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
             100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
             100 * m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_LDS_BANK_CONFLICT"] / (4 * m["SQ_WAVE_CYCLES"])))
PY
cd $R
timeout 300 python tools/bench_kernels.py head stack wgrad loss > $O/bench_kernels.txt 2>&1; grep "^conv\|^wgrad\|^loss\|UNet3D\|sfa_lift" $O/bench_kernels.txt
