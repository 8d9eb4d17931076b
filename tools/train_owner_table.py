"""GPU time of one steady-state training step by owner of the kernel, from the per-kernel table tools/summarize_trace.py writes
(tools/gpu_run_r5.sh proftrain):  python tools/train_owner_table.py <train_step_kernels.csv>"""
import csv
import sys
from collections import defaultdict


def owner(name):
    n = name.strip('"')
    if "at::native" in n or "at::cuda" in n or n.startswith("void at::") or "c10::" in n:
        return "ATen"
    if "Cijk_" in n or "rocblas" in n.lower() or "hipblaslt" in n.lower():
        return "rocBLAS / hipBLASLt"
    if "miopen" in n.lower() or n.startswith(("igemm_", "gcnAsm", "Sp3Asm", "naive_conv", "batchnorm", "Im2", "Col2", "SubTensor", "transpose_", "batched_transpose", "wrw_", "MIOpen")) or "gtc" in n:
        return "MIOpen"
    if "rocclr" in n or "copyBuffer" in n or "fillBuffer" in n or "hip_" in n.lower():
        return "runtime copies"
    if "nccl" in n.lower() or "rccl" in n.lower():
        return "RCCL"
    return "in-repo (libocc_hip.so)"


def main(path):
    head = open(path).readline().rstrip("\n")
    rows = [r for r in csv.DictReader(l for l in open(path) if not l.startswith("#"))]
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        o = owner(r["kernel"])
        agg[o][0] += int(float(r["calls"]))
        agg[o][1] += float(r["ms"])
    total = sum(v[1] for v in agg.values())
    print(head)
    print("# GPU time of one steady-state bf16-mode training step by owner of the kernel (rocprofv3 --kernel-trace; tools/gpu_run_r5.sh proftrain; tools/train_owner_table.py)")
    print(f"{'owner':28s} {'launches':>9s} {'ms':>10s} {'share':>7s}")
    for o, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{o:28s} {n:9d} {ms:10.2f} {100.0 * ms / total:6.1f}%")
    print(f"{'total':28s} {'':9s} {total:10.2f}")
    print("# largest kernels that are not in-repo")
    ext = sorted((r for r in rows if owner(r["kernel"]) != "in-repo (libocc_hip.so)"), key=lambda r: -float(r["ms"]))[:8]
    for r in ext:
        print(f"{int(float(r['calls'])):5d} launches {float(r['ms']):8.3f} ms  {r['kernel'][:120]}")


if __name__ == "__main__":
    main(sys.argv[1])
