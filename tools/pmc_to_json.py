"""profiles/head_conv_hbm_bytes.json from two rocprofv3 --pmc passes of tools/pmc_head.py:
    python tools/pmc_to_json.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 counts its 128-byte requests as 64 bytes --
MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import json
import re
import sys
from collections import defaultdict

ALGORITHMIC = 2 * 256 * 256 * 32 * 32 * 4 + 27 * 32 * 32 * 4     # in + out + weights of one 32->32 head launch


def mean_by_dilation(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"conv3d_c32_(?:slide|persist)_kernel<(\d)>", r["Kernel_Name"])
        if m and r["Counter_Name"] == counter:
            acc[m.group(1)].append(float(r["Counter_Value"]))
    return {d: sum(v) / len(v) for d, v in acc.items()}


def main(fetch_csv, write_csv, out):
    f, w = mean_by_dilation(fetch_csv, "FETCH_SIZE"), mean_by_dilation(write_csv, "WRITE_SIZE")
    per = {}
    for d in sorted(f):
        fb, wb = f[d] * 1024.0, w[d] * 1024.0
        per[d] = {"FETCH_SIZE_bytes_raw": fb, "fetch_bytes_corrected_x2": 2 * fb, "WRITE_SIZE_bytes": wb,
                  "hbm_bytes": 2 * fb + wb, "algorithmic_bytes": ALGORITHMIC, "ratio": (2 * fb + wb) / ALGORITHMIC}
    mix = {"1": 3, "2": 2, "3": 2}                                 # the 7 launches 32->32 of one frame
    bpl = sum(per[d]["hbm_bytes"] * n for d, n in mix.items()) / sum(mix.values())
    res = {"kernel": "conv3d_c32_slide_kernel (head conv 32->32 3x3x3 @256x256x32)",
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_head.py); "
                     "counters are KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)",
           "per_dilation": per, "bytes_per_launch": bpl,
           "note": "bytes_per_launch = mean over the 7 launches of one frame (3 x d=1, 2 x d=2, 2 x d=3)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({d: round(v["ratio"], 3) for d, v in per.items()}), "bytes/launch", bpl)


if __name__ == "__main__":
    main(*sys.argv[1:4])
