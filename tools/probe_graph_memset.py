"""Probe (GPU box): do hipMemsetAsync NODES of a captured hipGraph take effect on every replay, and do ATen's multi-block
reductions -- which zero their semaphores with such a node (ATen/native/cuda/Reduce.cuh) -- stay right under replay?

Background: DESIGN.md section 6 (round 4) records a captured memset of an odd-sized buffer that did not take effect on
replays; the captured training step still holds ~325 `aten::sum` launches per step, among them the convolutions' bias
gradients (`autograd3d._Conv3dFn.backward`).  Prints one JSON line; exit code 1 when any replay went wrong.
"""
import ctypes
import json
import sys

import torch


def memset_single(replays=6):
    """One graph per (size, offset): a lone memset node and its consumer."""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    out = {}
    for s in [1, 4, 8, 12, 16, 28, 64, 256, 4096]:
        for o in [0, 4, 1]:
            b = torch.ones(s + o + 64, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st = torch.cuda.current_stream().cuda_stream
                assert hip.hipMemsetAsync(ctypes.c_void_p(b.data_ptr() + o), 0, s, ctypes.c_void_p(st)) == 0
                b.add_(1)
            torch.cuda.synchronize()
            after_capture = int(b[o])                          # 1 = the call was captured, 0 = it ran eagerly
            vals = []
            for r in range(replays):
                g.replay()
                torch.cuda.synchronize()
                vals.append(int(b[o]))
            if vals != [1] * replays or after_capture != 1:
                out[f"{s}+{o}"] = {"after_capture": after_capture, "first_byte_after_replays": vals, "guard": int(b[o + s])}
    return out


def memset_nodes(replays=200):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    sizes = [1, 2, 3, 4, 8, 12, 16, 20, 28, 36, 60, 64, 100, 252, 256, 260, 1000, 1020, 1024, 4096, 4100, 65540]
    offs = [0, 4, 1]
    bufs = {(s, o): torch.zeros(s + o + 64, dtype=torch.uint8, device="cuda") for s in sizes for o in offs}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for b in bufs.values():
            b.add_(1)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        for (s, o), b in bufs.items():
            rc = hip.hipMemsetAsync(ctypes.c_void_p(b.data_ptr() + o), 0, s, ctypes.c_void_p(st))
            assert rc == 0, rc
            b.add_(1)                                   # the consumer right behind the memset node
    bad = {}
    for r in range(replays):
        g.replay()
        if r % 20 == 19 or r < 3:
            torch.cuda.synchronize()
            for (s, o), b in bufs.items():
                if not bool((b[o:o + s] == 1).all()):
                    bad.setdefault(f"{s}+{o}", []).append(r)
    return bad


def aten_column_sums(replays=60):
    bad = {}
    for rows, cs, c in [(4096, 8, 2), (32768, 8, 2), (262144, 8, 2), (2097152, 8, 2), (32768, 24, 20), (262144, 24, 20),
                        (2097152, 32, 32), (16384, 8, 2), (902800, 80, 80), (225700, 160, 160),
                        (56730, 320, 320), (4096, 512, 512), (65536, 96, 96), (65536, 224, 224)]:
        inp = torch.randn(rows, cs, device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            inp[:, :c].sum(0, dtype=torch.float32)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            junk = torch.full((c,), float("nan"), device="cuda")      # a previous tenant's leftovers in the pool
            del junk
            out = inp[:, :c].sum(0, dtype=torch.float32)
        for r in range(replays):
            inp.normal_()
            g.replay()
            want = inp[:, :c].double().sum(0)
            err = float((out.double() - want).abs().max() / (want.abs().max() + 1.0))
            if not err < 1e-3:
                bad.setdefault(f"{rows}x{cs}:{c}", []).append((r, err))
    return bad


if __name__ == "__main__":
    res = {"memset_single_bad": memset_single(), "memset_nodes_bad": sorted(memset_nodes()), "aten_sum_bad": {k: v[:4] for k, v in aten_column_sums().items()}}
    print(json.dumps(res))
    sys.exit(1 if res["memset_single_bad"] or res["memset_nodes_bad"] or res["aten_sum_bad"] else 0)
