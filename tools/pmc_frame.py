"""Per-kernel PMC table of one config-2 frame from rocprofv3 counter-collection CSVs of `bench.py` (separate --pmc passes):

    python tools/pmc_frame.py <sq counter_collection.csv> [<FETCH_SIZE csv> <WRITE_SIZE csv>] > profiles/r02_pmc_frame.txt

SQ pass counters: GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES.  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) as in tools/final_round_run.sh; FETCH_SIZE is in KB and doubled on gfx950
(MI355X_MICROARCH.md), WRITE_SIZE in KB.  Kernel families are grouped by name; the numbers are per launch averages over
every launch in the trace (the profiler serialises kernels, durations are a little longer than in the un-profiled frame)."""
import csv
import re
import sys
from collections import defaultdict

FAMILIES = [("conv3d_c32_slide_x3_kernel", "K2s3 head conv 32->32 3x3x3 (3x bf16 split)"), ("conv3d_c32_slide_kernel", "K2s head conv 32->32 3x3x3"),
            ("gemm_x3_panel_splitk_kernel", "K21 split-K panel GEMM (project convolutions of the 1/16, 1/32 stages; round 6)"),
            ("splitk_reduce_kernel", "K21 second launch (sum of the K chunks + shift + skip)"),
            ("gemm_x3_panel_kernel", "K16p panel-stationary GEMM (3x bf16 split, pre-split weights; late round 5)"), ("gemm_x3_pack", "K16 weight pre-split (once per weight)"), ("gemm_x3", "K16 fp32 GEMM (3x bf16 split)"), ("conv3d_igemm_kernel", "K2 generic 3-D igemm"), ("conv3d_bf16_kernel", "K2b 3-D conv (3x bf16 split: small volumes, transposed-conv phases)"), ("bneck_a_kernel", "K14 bottleneck A (conv1 + conv_z)"),
            ("bneck_b_kernel", "K14 bottleneck B (conv_y + conv_x + conv5)"),
            ("wino3x3_kernel", "K10 fused Winograd 3x3"), ("pw_gemm_splitk_kernel", "K11s split-K pointwise GEMM"),
            ("pw_gemm_kernel", "K11 streaming pointwise GEMM"), ("upconv_gather_kernel", "K12 upsample-shift-accumulate"),
            ("dwconv2d_kernel", "depthwise + SE pooling"), ("lift_p1_kernel", "K1b lift"), ("lift_proj_kernel", "K1b fused lift (projection + frustum)"), ("Cijk_", "library GEMM (hipBLASLt / rocBLAS)"),
            ("cascade_tail", "cascade tail"), ("se_reduce_kernel", "SE reduce"), ("se_expand", "SE expand"), ("wino_input_kernel", "K9 input transform"),
            ("stem_conv3x3_kernel", "encoder stem conv + BN + swish (round 5)"), ("depthnet_gate_kernel", "DepthNet gate (round 5)"),
            ("miopen", "MIOpen"), ("MIOpen", "MIOpen")]


def family(name):
    for key, label in FAMILIES:
        if key in name:
            return label
    return None


def load(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        f = family(r["Kernel_Name"])
        if f is None:
            continue
        acc[f][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
            acc[f]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc


def main(sq, fetch=None, write=None):
    a = load(sq)
    fb = load(fetch) if fetch else {}
    wb = load(write) if write else {}
    print("%-40s %7s %9s %10s %12s %12s" % ("kernel family", "launches", "avg us", "MFMA busy", "HBM fetch MB", "HBM write MB"))
    for _, label in FAMILIES:
        if label not in a:
            continue
        c = a[label]
        m = {k: sum(v) / len(v) for k, v in c.items()}
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        busy = 100.0 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0) if cyc else 0.0
        f = 2048.0 * sum(fb[label]["FETCH_SIZE"]) / len(fb[label]["FETCH_SIZE"]) / 1e6 if label in fb else float("nan")
        w = 1024.0 * sum(wb[label]["WRITE_SIZE"]) / len(wb[label]["WRITE_SIZE"]) / 1e6 if label in wb else float("nan")
        print("%-40s %7d %9.1f %9.1f%% %12.1f %12.1f" % (label, len(c["GRBM_GUI_ACTIVE"]), m["ns"] / 1e3, busy, f, w))


if __name__ == "__main__":
    main(*sys.argv[1:4])
