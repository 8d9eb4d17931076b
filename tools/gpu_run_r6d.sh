#!/bin/bash
# round 6: DepthNet on a side stream beside the decoder's fine levels (OCCDEPTH_DEPTHNET_OVERLAP=1) against the serial frame, same box
mkdir -p gpurun_out/r6d
B="python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5"
for v in 1 0 1 0; do
  OCCDEPTH_DEPTHNET_OVERLAP=$v timeout 400 $B > gpurun_out/r6d/bench_ov${v}_$RANDOM.json 2> gpurun_out/r6d/err_$v.txt
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r6d/bench_ov*.json")):
    try:
        t = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
        print(f.split("/")[-1], round(t["ms_per_step"], 3), "ms/frame; stages", {k: round(v, 3) for k, v in t["stages_ms"].items()}, "parity", (t.get("parity_rel_err") or {}).get("worst_of_all_outputs"), t["config"].get("graph_all_error"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 gpurun_out/r6d/err_1.txt
OCCDEPTH_DEPTHNET_OVERLAP=1 timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_lightning_hooks.py -m gpu -x -q 2>&1 | tail -4
