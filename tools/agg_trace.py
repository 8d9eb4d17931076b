"""Aggregate a rocprofv3 --kernel-trace CSV by kernel name:
    python tools/agg_trace.py <kernel_trace.csv> <out.csv> [divide_by] [marker_kernel markers_per_iter]
With a marker, only ONE steady-state iteration is kept: the launches between the first marker of the second-to-last
iteration and the first marker of the last one (e.g. marker ssc_stats_kernel, 2 per training step)."""
import csv
import re
import sys
from collections import defaultdict


def main(path, out, div=1.0, marker=None, per_iter=1):
    agg = defaultdict(lambda: [0, 0])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if marker:
        idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
        assert len(idx) >= 2 * per_iter, "not enough marker launches"
        rows = rows[idx[-2 * per_iter]:idx[-per_iter]]
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)[:100]
        a = agg[n]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values())
    with open(out, "w") as f:
        f.write(f"# total GPU busy {tot / 1e6 / div:.2f} ms per iteration (trace divided by {div:g})\n")
        f.write("kernel,calls_per_iter,ms_per_iter,avg_us,percent\n")
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{n}\",{c / div:.1f},{t / 1e6 / div:.3f},{t / 1e3 / c:.2f},{100 * t / tot:.2f}\n")
    print(open(out).read()[:4500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 1.0,
         sys.argv[4] if len(sys.argv) > 4 else None, int(sys.argv[5]) if len(sys.argv) > 5 else 1)
