#!/bin/bash
# scratch dev session (GPU box): targeted tests + A/B bench lines -> gpurun_out/$1/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-dev}
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_winograd2d.py tests/test_hip_vs_aten.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
line() { python - "$1" <<'PY'
import json, sys
try:
    t = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][0])
    print(sys.argv[1].split("/")[-1], round(t["ms_per_step"], 3), t.get("stages_ms"), t.get("parity_rel_err"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
timeout 300 $B > $O/b_default.json 2> $O/b_default.err; line $O/b_default.json
OCCDEPTH_PHASES_ONE_LAUNCH=0 timeout 300 $B > $O/b_phases8.json 2> $O/b_phases8.err; line $O/b_phases8.json
OCCDEPTH_PW_EXPAND_LIB_BELOW=60000 timeout 300 $B > $O/b_expand60k.json 2> $O/b_expand60k.err; line $O/b_expand60k.json
OCCDEPTH_PW_EXPAND_LIB_BELOW=15000 timeout 300 $B > $O/b_expand15k.json 2> $O/b_expand15k.err; line $O/b_expand15k.json
timeout 300 python tools/frame_table.py > $O/frame_per_launch.txt 2> $O/frame_per_launch.err; grep "upconv\|phases" $O/frame_per_launch.txt | cut -c1-125
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -x > $O/pytest_b.txt 2>&1; tail -3 $O/pytest_b.txt
