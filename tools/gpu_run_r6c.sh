#!/bin/bash
# round 6: K12 with shared source rows (COOP) against one row pair per wave, same box: unit tests, decoder goldens, frame A/B
mkdir -p gpurun_out/r6c
timeout 600 python -m pytest tests/test_winograd2d.py tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r6c/tests.log
B="python bench.py --no-extras --no-cpu-baseline --no-parity --steps 30 --warmup 5"
for v in 1 0 1 0; do
  OCCD_UPCONV_COOP=$v timeout 400 $B > gpurun_out/r6c/bench_coop${v}_$RANDOM.json 2> /dev/null
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r6c/bench_coop*.json")):
    try:
        t = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
        r = t["roofline_2d"]
        print(f.split("/")[-1], round(t["ms_per_step"], 3), "ms/frame; 2d", round(t["stages_ms"]["net_rgb_2d_ms"], 3), "K12", r.get("K12_upconv_gather"))
    except Exception as e:
        print(f, "failed", e)
PY
cat gpurun_out/r6c/tests.log
