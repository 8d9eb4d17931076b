#!/bin/bash
# round 6, first frame measurement: new unit tests + the forward bench with K21 / fused SE on and off (same box)
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_stem_gate_gpu.py -m gpu -x -q -k "se_gate" 2>&1 | tail -15 > gpurun_out/r6a_se.log
timeout 560 python -m pytest tests/test_syncbn_lightning_gpu.py -m gpu -x -q -s 2>&1 | tail -30 > gpurun_out/r6a_syncbn.log
B="python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5"
timeout 400 $B > gpurun_out/r6a_bench_new.json 2> gpurun_out/r6a_bench_new.err
OCCDEPTH_PW_PROJECT_SPLITK=0 timeout 400 $B --no-parity > gpurun_out/r6a_bench_no_splitk.json 2> /dev/null
OCCD_SE_FUSED=0 timeout 400 $B --no-parity > gpurun_out/r6a_bench_no_sefused.json 2> /dev/null
OCCD_SE_FUSED=0 OCCDEPTH_PW_PROJECT_SPLITK=0 timeout 400 $B --no-parity > gpurun_out/r6a_bench_r5like.json 2> /dev/null
timeout 400 $B --no-parity > gpurun_out/r6a_bench_new2.json 2> /dev/null
for f in new no_splitk no_sefused r5like new2; do python - <<PY
import json
try:
    t = json.loads([l for l in open("gpurun_out/r6a_bench_$f.json") if l.startswith('{"metric"')][-1])
    print("$f", round(t["ms_per_step"], 3), "ms/frame; 2d", round(t["stages_ms"]["net_rgb_2d_ms"], 3), "launches", t.get("roofline_2d", {}).get("launches_per_frame_all_kinds"), "parity", (t.get("parity_rel_err") or {}).get("worst_of_all_outputs"))
except Exception as e:
    print("$f", "failed", e)
PY
done
