"""Steady-state per-kernel summary from a rocprofv3 --kernel-trace CSV of bench.py.

The first forward of a process runs MIOpen's first-call fallbacks (naive convolutions, 25 ms each) and hipcc-free
JIT lookups; `--stats` mixes them into the averages.  This script keeps only the last N forwards, delimited by the
persistent head kernel `conv3d_c32_slide_kernel<1>` (4 launches per forward: conv0, conv1.0, conv2.0, wide cls).

    python tools/summarize_trace.py <kernel_trace.csv> <out.csv> [n_forwards=5]
"""
import csv
import re
import sys
from collections import defaultdict


def train_main(path, out):
    """Training trace (tools/bench_train.py): one steady-state step = the kernels between the last two optimizer
    updates (fused AdamW multi-tensor kernels mark the end of a step)."""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
    assert adam, "no optimizer kernels in the trace"
    ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 50]    # last kernel of each update
    assert len(ends) >= 2, "need two optimizer updates"
    sel = rows[ends[-2] + 1:ends[-1] + 1]
    agg = defaultdict(lambda: [0, 0.0])
    for r in sel:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:120]
        a = agg[n]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values())
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    with open(out, "w") as f:
        f.write(f"# one steady-state training step (between the last two AdamW updates): {len(sel)} launches, "
                f"GPU busy {tot / 1e6:.2f} ms, wall span {span / 1e6:.2f} ms\n")
        f.write("kernel,calls,ms,avg_us,percent\n")
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{n}\",{c},{t / 1e6:.3f},{t / 1e3 / c:.2f},{100 * t / tot:.2f}\n")
    print(open(out).read()[:6000])


def main(path, out, n_fwd=5):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # (the dilation-1 head launches: conv3d_c32_slide_kernel<1> in the exact-fp32 mode, conv3d_c32_slide_x3_kernel<1, NRES> in the
    # default split mode -- 4 per forward either way)
    marks = [i for i, r in enumerate(rows) if re.search(r"conv3d_c32_slide(_x3)?_kernel<1[,>]", r["Kernel_Name"])]
    assert len(marks) >= 4 * (n_fwd + 1), "not enough forwards in the trace"
    # a forward ends with its 4th <1> launch (+ the cascade tail right after); start after forward (F - n_fwd)'s end
    fwd_ends = marks[3::4]
    start = fwd_ends[-n_fwd - 1] + 2
    end = fwd_ends[-1] + 2
    sel = rows[start:end]
    agg = defaultdict(lambda: [0, 0.0])
    for r in sel:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\((ConvP|LiftP|FlospP|PersistP|SlideP)\)", "", n)[:110]
        a = agg[n]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values())
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    with open(out, "w") as f:
        f.write(f"# steady state: last {n_fwd} forwards of `rocprofv3 --kernel-trace -- python bench.py ...`; "
                f"GPU busy {tot / 1e6 / n_fwd:.2f} ms/forward, wall span {span / 1e6 / n_fwd:.2f} ms/forward\n")
        f.write("kernel,calls_per_forward,ms_per_forward,avg_us,percent\n")
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{n}\",{c / n_fwd:.1f},{t / 1e6 / n_fwd:.3f},{t / 1e3 / c:.2f},{100 * t / tot:.2f}\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "train":
        train_main(sys.argv[1], sys.argv[2])
    else:
        main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5)
