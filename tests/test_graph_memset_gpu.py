"""Captured hipMemsetAsync nodes (csrc/graph_fix.hip, train_graph.new_graph / seal_graph).

On ROCm 7.2 / gfx950 a memset node of an instantiated hipGraph fills with its value on the first launch only; ATen's
multi-block reductions zero their semaphores with such a node, which is how the captured training step lost a bias gradient
now and then.  Every graph of the package has its memset nodes rewritten as fill kernels before instantiation; these tests
drive that rewrite on bare memset nodes (all alignments / sizes / element sizes), on a reduction, and record what the
un-rewritten node does (informative: the stack's behaviour, not ours).  No reference counterpart: scripts/train.py:176-206
launches eagerly.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _hiprt():
    rt = ctypes.CDLL("libamdhip64.so")
    rt.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    rt.hipMemsetAsync.restype = ctypes.c_int
    rt.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    rt.hipMemsetD32Async.restype = ctypes.c_int
    rt.hipMemsetD16Async.argtypes = [ctypes.c_void_p, ctypes.c_ushort, ctypes.c_size_t, ctypes.c_void_p]
    rt.hipMemsetD16Async.restype = ctypes.c_int
    return rt


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _capture_fill(rt, graph, b, o, v, s, esz=1):
    with torch.cuda.graph(graph):
        if esz == 1:
            rc = rt.hipMemsetAsync(ctypes.c_void_p(b.data_ptr() + o), v, s, _stream())
        elif esz == 2:
            rc = rt.hipMemsetD16Async(ctypes.c_void_p(b.data_ptr() + o), v, s // 2, _stream())
        else:
            rc = rt.hipMemsetD32Async(ctypes.c_void_p(b.data_ptr() + o), v, s // 4, _stream())
        assert rc == 0, rc


def _expected(v, s, esz):
    pat = [(v >> (8 * k)) & 0xFF for k in range(esz)]
    return torch.tensor([pat[i % esz] for i in range(s)], dtype=torch.uint8)


def test_rewritten_memset_nodes_fill_on_every_replay(hip_lib):
    from occdepth_amd import train_graph
    assert train_graph.FIX_MEMSETS
    rt = _hiprt()
    cases = [(o, v, s, 1) for o in (0, 1, 4, 7, 16, 250) for v in (0, 0x5A) for s in (1, 3, 4, 12, 15, 16, 17, 64, 100, 4099, 65540)]
    cases += [(o, 0x1234ABCD, s, 4) for o in (0, 4, 12, 256) for s in (4, 12, 64, 4100)]
    cases += [(o, 0xBEEF, s, 2) for o in (0, 2, 6, 254) for s in (2, 6, 30, 1026)]
    for o, v, s, esz in cases:
        b = torch.empty(s + o + 300, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        g = train_graph.new_graph()
        _capture_fill(rt, g, b, o, v, s, esz)
        assert train_graph.seal_graph(g) == 1, (o, v, s, esz)
        want = _expected(v, s, esz)
        for r in range(3):
            b.fill_(0xEE)
            g.replay()
            got = b.cpu()
            assert torch.equal(got[o:o + s], want), (o, v, s, esz, r, bytes(got[o:o + min(s, 24)].tolist()).hex())
            assert bool((got[:o] == 0xEE).all()) and bool((got[o + s:] == 0xEE).all()), (o, v, s, esz, r, "wrote outside")
        del g, b


def test_edges_survive_the_rewrite(hip_lib):
    """producer kernel -> memset -> consumer kernel: the fill kernel sits between the same neighbours."""
    from occdepth_amd import train_graph
    rt = _hiprt()
    b = torch.zeros(64, dtype=torch.int32, device="cuda")
    out = torch.zeros(64, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = train_graph.new_graph()
    with torch.cuda.graph(g):
        b.add_(7)                                                        # producer: must run BEFORE the fill
        assert rt.hipMemsetAsync(ctypes.c_void_p(b.data_ptr() + 16), 0, 64, _stream()) == 0
        out.copy_(b + 1)                                                 # consumer: must run AFTER it
    assert train_graph.seal_graph(g) == 1
    for r in range(4):
        g.replay()
        torch.cuda.synchronize()
        want = torch.full((64,), 7 * (r + 1) + 1, dtype=torch.int32)
        want[4:20] = 1
        assert torch.equal(out.cpu(), want), r
        with torch.no_grad():
            b[4:20] = 7 * (r + 1)                                        # keep the untouched lanes' arithmetic simple


def test_multi_block_reduction_in_a_sealed_graph(hip_lib):
    """The convolutions' bias gradient (autograd3d._Conv3dFn.backward): a column sum over 2M rows is a multi-block ATen
    reduction with semaphores; sealed, it is right on every replay and its memset node is gone."""
    from occdepth_amd import train_graph
    for rows, cs, c in [(2097152, 8, 2), (262144, 24, 20), (902800, 80, 80)]:
        inp = torch.randn(rows, cs, device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            inp[:, :c].sum(0, dtype=torch.float32)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = train_graph.new_graph()
        with torch.cuda.graph(g):
            out = inp[:, :c].sum(0, dtype=torch.float32)
        n = train_graph.seal_graph(g)
        assert n >= 1, "expected ATen's semaphore memset in the capture"
        for r in range(5):
            inp.normal_()
            g.replay()
            want = inp[:, :c].double().sum(0)
            assert float((out.double() - want).abs().max()) < 1e-3 * (float(want.abs().max()) + 1.0), (rows, cs, c, r)


def test_unrewritten_memset_node_behaviour_is_recorded(hip_lib, capsys):
    """Informative: what a plain captured memset does on replays on this stack (the reason for the rewrite).  Never fails on
    the stack's behaviour -- a fixed runtime simply reports zero bad replays."""
    rt = _hiprt()
    bad, total, seen = 0, 0, set()
    for v in (0, 0x5A):
        for s in (4, 12, 64):
            b = torch.empty(s + 300, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            _capture_fill(rt, g, b, 0, v, s)
            for r in range(3):
                b.fill_(0xEE)
                g.replay()
                total += 1
                if not bool((b[:s] == v).all()):
                    bad += 1
                    seen.add(bytes(b[:min(s, 16)].cpu().tolist()).hex())
    with capsys.disabled():
        # (process-dependent: tools/probe_graph_memset2.py, a bare interpreter, sees 90 of 90 lone memset nodes fill with a
        # stale pattern from the second launch on -- 0 of 90 with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 --, this pytest process
        # usually none: the replayed AQL packet's pattern argument points at recycled host memory)
        print(f"\n[plain hipGraph memset nodes: {bad} of {total} replays filled with a wrong pattern {sorted(seen)[:4]}]")
