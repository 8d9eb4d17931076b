"""The committed bench lines (profiles/r02_final_bench.json, profiles/r03_bench.json, produced by `python bench.py` on an
MI355X) carry every field the driver contract names, and their derived numbers are self-consistent."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["r02_final_bench.json", "r03_bench.json"])
def test_committed_bench_line_matches_contract(name):
    d = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(d["n_gpus"] * 1e3 / d["ms_per_step"], rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == pytest.approx(157.3)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert r["achieved"] == pytest.approx(r["gflop_per_launch"] / r["avg_launch_ms"], rel=1e-6)      # GFLOP / ms = TFLOP/s
    assert r["gflop_per_launch"] == pytest.approx(2 * 256 * 256 * 32 * 27 * 32 * 32 / 1e9, rel=1e-9)
    assert r["traffic"] is None or r["traffic"] > 5.3e8                                              # >= algorithmic bytes
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "frames/s" and c["cores"] >= 1 and c["value"] > 0
    assert isinstance(c["sample"], str) and c["sample"]
