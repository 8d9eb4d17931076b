"""Per-kernel averages of every counter in one or more rocprofv3 counter-collection CSVs (separate --pmc passes):
    python tools/pmc_table.py <csv> [<csv> ...] [--match substring]
Derived columns: MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); FETCH_SIZE is in KB and
doubled on gfx950, WRITE_SIZE in KB (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:70]


def main(paths, match=None):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if match and match not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    counters = sorted({c for v in acc.values() for c in v if c != "_ns"})
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]["_ns"])):
        m = {c: sum(x) / len(x) for c, x in v.items()}
        n = max(len(x) for c, x in v.items() if c != "_ns")
        line = [f"{k:70s} launches {n:4d}  avg {m['_ns'] / 1e3:8.1f} us"]
        if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            line.append(f"MFMA busy {100.0 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0):5.1f} %")
        if "FETCH_SIZE" in m:
            line.append(f"HBM fetch {2048.0 * m['FETCH_SIZE'] / 1e6:8.1f} MB")
        if "WRITE_SIZE" in m:
            line.append(f"HBM write {1024.0 * m['WRITE_SIZE'] / 1e6:8.1f} MB")
        print("  ".join(line))
        print("      " + "  ".join(f"{c}={m[c]:.4g}" for c in counters if c in m))


if __name__ == "__main__":
    args = sys.argv[1:]
    match = None
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1]
        del args[i:i + 2]
    main(args, match)
