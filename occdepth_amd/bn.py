"""Training-mode BatchNorm (+ activation + residual) on the K13 kernels (csrc/bn.hip), forward and backward.

    y = bn_act(bn_module, x, act="relu" | "leaky" | "swish" | None, slope=0.01, res=None, res_first=False)

is `act(bn(x) [+ res]) [+ res]` of a `torch.nn.BatchNorm{2,3}d` (or `shard.SyncBatchNorm` / `torch.nn.SyncBatchNorm`) in TRAINING mode -- the pattern of
every normalisation site of the reference's step (occdepth/models/DDR.py:111-139 `relu(bn(conv(x)))`, `relu(bn5(.) + skip)`;
modules.py:40-46 `y += bn2(conv2(relu(bn1(conv1(x)))))`; unet2d.py:24-46 conv-BN-LeakyReLU; the EfficientNet blocks'
BN-swish) -- as two passes over the activation per direction plus per-channel kernels, instead of the backend's
batch_norm and separate activation / add kernels.  The module keeps its parameters, buffers and state_dict; running
statistics and `num_batches_tracked` are updated on the device as nn.BatchNorm does.

Statistics that span ranks (`shard.SyncBatchNorm`, and `torch.nn.SyncBatchNorm` as Lightning's converter produces it for the
reference's `sync_batchnorm=True`, scripts/train.py:179 -- its `process_group`, None = the world): the same
kernels with ONE all-reduce of the packed (2C + 1)-element float64 vector in the forward and one of 2C floats in the
backward (see csrc/bn.hip); the parameter gradients stay per-rank sums, the gradient buckets average them.

Layouts: channels-last rows ((N, C, *spatial) tensors whose memory is (N, *spatial, cs >= C), fp32 or bf16) and NCHW /
NCDHW contiguous fp32.  On the CPU (the test-suite's host-logic runs) and for eval mode the call is plain torch.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import hip

ACT = {None: 0, "none": 0, "relu": 1, "swish": 2, "leaky": 3}


BnArgs = hip.BnArgs


def _rows_geometry(x):
    """(rows, cs) when x (N, C, *spatial) is laid out channels-last with dense rows of cs >= C elements, else None."""
    if x.dim() < 3:
        return None
    cl = x.permute(0, *range(2, x.dim()), 1)
    if cl.stride(-1) != 1 and cl.shape[-1] != 1:
        return None
    cs = cl.stride(-2)
    if cs < x.shape[1] or cs % 4 != 0:
        return None
    expect = cs
    for d in range(cl.dim() - 2, -1, -1):
        if cl.shape[d] != 1 and cl.stride(d) != expect:
            return None
        expect *= cl.shape[d]
    esz = x.element_size()
    if x.data_ptr() % (4 * esz) != 0:
        return None
    return x.numel() // x.shape[1], cs


class _Geom:
    """How a tensor of the call is addressed by the kernels (all tensors of a call share layout, dtype and extent)."""

    def __init__(self, x):
        self.C = x.shape[1]
        self.dtype = x.dtype
        rows = _rows_geometry(x) if not (x.is_contiguous() and x.dtype == torch.float32) or x.dim() == 2 else None
        if x.is_contiguous() and x.dtype == torch.float32 and x.dim() > 2:
            self.layout, self.batch, self.S, self.rows = 1, x.shape[0], x[0, 0].numel(), 0
        elif rows is not None:
            self.layout, self.batch, self.S, self.rows = 0, 0, 0, rows[0]
        else:
            raise RuntimeError("bn_act: unsupported tensor layout")

    @staticmethod
    def supported(x):
        if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16) or x.dim() < 3:
            return False
        if x.is_contiguous() and x.dtype == torch.float32:
            return x.shape[0] * x.shape[1] <= 65535
        return _rows_geometry(x) is not None

    def like(self, x, t, what):
        """cs of tensor t, which must have x's shape, dtype and layout kind."""
        if t.shape != x.shape or t.dtype != x.dtype:
            raise RuntimeError(f"bn_act: {what} must have the activation's shape and dtype")
        if self.layout == 1:
            if not t.is_contiguous():
                raise RuntimeError(f"bn_act: {what} must be contiguous like the activation")
            return 0
        g = _rows_geometry(t)
        if g is None:
            raise RuntimeError(f"bn_act: {what} must be channels-last like the activation")
        return g[1]

    def empty_like(self, x):
        """Output tensor in x's layout: NCHW contiguous, or channels-last rows padded to ceil8(C) (zero pads)."""
        if self.layout == 1:
            return torch.empty_like(x), 0
        cs = hip.round_up(self.C, 8)
        buf = torch.empty((x.shape[0],) + tuple(x.shape[2:]) + (cs,), device=x.device, dtype=x.dtype)
        return buf[..., :self.C].permute(0, x.dim() - 1, *range(1, x.dim() - 1)), cs

    def args(self):
        a = BnArgs()
        a.C, a.dtype, a.layout = self.C, 1 if self.dtype == torch.bfloat16 else 0, self.layout
        a.rows, a.S, a.batch = self.rows, self.S, self.batch
        return a


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _group_active(group):
    from . import shard
    return shard._group_active(group)


def _to_rows(t):
    """t (N, C, *spatial) as a channels-last rows tensor the kernels can address (a copy only when it is not one)."""
    if _rows_geometry(t) is not None:
        return t
    C = t.shape[1]
    cs = hip.round_up(C, 8)
    nd = t.dim()
    buf = torch.zeros((t.shape[0],) + tuple(t.shape[2:]) + (cs,), device=t.device, dtype=t.dtype)
    buf[..., :C].copy_(t.permute(0, *range(2, nd), 1))
    return buf[..., :C].permute(0, nd - 1, *range(1, nd - 1))


class _BNActFn(torch.autograd.Function):
    """Launch plan per direction: [sync] reduce, combine, all-reduce, finish, apply; [local] reduce, combine + finish, apply;
    [local, small NCHW tensor] ONE launch (occd_bn_fwd_small / occd_bn_bwd_small)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, running_mean, running_var, nbt, eps, momentum, act, slope, res_first, group,
                sync, ngroups, proto=None):
        geo = _Geom(x)
        C = geo.C
        dev = x.device
        G = int(ngroups)
        if x.shape[0] % G:
            raise RuntimeError("bn_act: the batch does not divide into the view groups")
        n = x.shape[0] // G
        exchange = bool(sync and _group_active(group))
        # synchronised layers follow the protocol the ranks AGREED on for this module (`proto`, see bn_act): the one-launch
        # in-kernel exchange or the packed all-reduce -- never a choice made from this rank's tensor shape alone
        one_launch = bool(proto["one_launch"]) if (exchange and proto is not None) else None
        packed = torch.empty(G, 2 * C + 1, device=dev, dtype=torch.float64)
        vec = torch.empty(G, 4, C, device=dev, dtype=torch.float32)       # per group: mean, invstd, a, b
        w = weight.detach().float() if weight is not None else None
        b = bias.detach().float() if bias is not None else None
        y, _ = geo.empty_like(x)
        small = True
        for g in range(G):                                                # one statistics group per stacked view, in call order
            sl = slice(g * n, (g + 1) * n)
            small = _BNActFn._forward_group(x[sl] if G > 1 else x, y[sl] if G > 1 else y,
                                            (res[sl] if G > 1 else res) if res is not None else None, w, b, running_mean,
                                            running_var, nbt, eps, momentum, act, slope, res_first, group, exchange, vec[g],
                                            packed[g], one_launch) and small
        # the pre-activation's sign comes from y when a residual entered before the activation (x a + b alone is not it)
        need_y = act != 0 and res is not None and res_first
        ctx.save_for_backward(x, vec, packed, weight, y if need_y else None)
        ctx.cfg = (act, float(slope), bool(res_first), res is not None, group, exchange, G, one_launch)
        return y

    @staticmethod
    def _forward_group(x, y, res, w, b, running_mean, running_var, nbt, eps, momentum, act, slope, res_first, group, exchange,
                       vec, packed, one_launch=None):
        lib = hip.load()
        st = hip._stream()
        geo = _Geom(x)
        C = geo.C
        dev = x.device
        a = geo.args()
        a.x = x.data_ptr()
        a.x_cs = geo.like(x, x, "x")
        fin = (float(eps), float(momentum if momentum is not None else 0.0), _ptr(w), _ptr(b), _ptr(running_mean),
               _ptr(running_var), _ptr(nbt), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr())
        ycs = geo.like(x, y, "y") if geo.layout == 0 else 0
        a.out, a.out_cs = y.data_ptr(), ycs
        a.cw = min(hip.round_up(C, 8), ycs) if geo.layout == 0 else 0
        if res is not None:
            a.res, a.res_cs = res.data_ptr(), geo.like(x, res, "res")
        a.act, a.slope, a.res_first = act, float(slope), 1 if res_first else 0
        small = bool(lib.occd_bn_small_ok(ctypes.byref(a)))
        xc = None
        if exchange:
            small = _BNActFn._agreed_small(small, one_launch, x)
        if small and exchange:
            # synchronised small layer: still ONE launch -- the ranks' statistics are exchanged inside it through the
            # peer-mapped channel mailboxes (shard.SmallAllReduce) when they are installed for the group
            from .shard import channel_exchange
            xc = channel_exchange(group, C, dev)
            if xc is None:
                raise RuntimeError("bn_act: the ranks agreed on the in-kernel exchange for this layer but no peer-memory "
                                   "exchange is installed for its process group any more")
        if small and xc is not None:
            hip._check(lib.occd_bn_fwd_small_xchg(ctypes.byref(a), packed.data_ptr(), *fin, *xc, st), "occd_bn_fwd_small_xchg")
        elif small:
            hip._check(lib.occd_bn_fwd_small(ctypes.byref(a), packed.data_ptr(), *fin, st), "occd_bn_fwd_small")
        else:
            a.nblk = lib.occd_bn_blocks(ctypes.byref(a))
            if a.nblk <= 0:
                raise RuntimeError("occd_bn_blocks failed")
            partial = torch.empty(a.nblk * 2 * hip.round_up(C, 4), device=dev, dtype=torch.float32)
            a.partial = partial.data_ptr()
            hip._check(lib.occd_bn_stats(ctypes.byref(a), st), "occd_bn_stats")
            if exchange:
                import torch.distributed as dist
                hip._check(lib.occd_bn_stats_combine(ctypes.byref(a), packed.data_ptr(), st), "occd_bn_stats_combine")
                from .shard import packed_all_reduce
                packed_all_reduce(packed, group)
                hip._check(lib.occd_bn_finish(packed.data_ptr(), C, *fin, st), "occd_bn_finish")
            else:
                hip._check(lib.occd_bn_stats_finish(ctypes.byref(a), packed.data_ptr(), *fin, st), "occd_bn_stats_finish")
            a.a, a.b = vec[2].data_ptr(), vec[3].data_ptr()
            hip._check(lib.occd_bn_apply(ctypes.byref(a), st), "occd_bn_apply")
        return small

    @staticmethod
    def _agreed_small(local_ok, one_launch, x):
        """The protocol of a synchronised layer is the group's, not this rank's: the one-launch exchange only when the
        ranks agreed on it (every rank's tensor fitted at the layer's first call); then a rank whose tensor has since
        outgrown the one-launch kernel cannot follow and must say so -- its peers are waiting in the channel mailboxes."""
        if not one_launch:
            return False                                 # packed all-reduce: always possible, whatever the local shape
        if not local_ok:
            raise RuntimeError(f"bn_act: this SyncBatchNorm layer was first called with tensors every rank could handle "
                               f"in one launch, now this rank's input {tuple(x.shape)} does not fit; ranks must keep "
                               "comparable shapes per layer, or set OCCDEPTH_SYNCBN_ONE_LAUNCH=0")
        return True

    @staticmethod
    def backward(ctx, gy):
        x, vec, packed, weight, y = ctx.saved_tensors
        act, slope, res_first, has_res, group, exchange, G, one_launch = ctx.cfg
        geo = _Geom(x)
        C = geo.C
        dev = x.device
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gy = gy.contiguous() if geo.layout == 1 else _to_rows(gy)
        k = torch.empty(G, 5, C, device=dev, dtype=torch.float32)         # per group: k1, k2, k3, gw, gb
        want_w = weight is not None
        gx, _ = geo.empty_like(x)
        gres = None
        if has_res:
            if res_first and act != 0:
                gres, _ = geo.empty_like(x)
            else:
                gres = gy                                                   # added after the activation (or no activation)
        n = x.shape[0] // G
        for g in range(G):
            sl = slice(g * n, (g + 1) * n)
            cut = (lambda t: t[sl] if t is not None and G > 1 else t)
            _BNActFn._backward_group(cut(x), cut(gy), cut(y), cut(gx), cut(gres) if (has_res and res_first and act != 0) else None,
                                     vec[g], packed[g], k[g], want_w, act, slope, res_first, group, exchange, one_launch)
        gw = gb = None
        if want_w:
            # parameter gradients: sum over the view groups -- no launch at all for one group, ONE for both vectors otherwise
            # (these were 2 x 266 `aten::sum` launches of a config-2 step, most of them over a single row)
            wb = k[0, 3:5] if G == 1 else k[:, 3:5].sum(0)
            gw, gb = wb[0].to(weight.dtype), wb[1].to(weight.dtype)
        return gx, gw, gb, gres, None, None, None, None, None, None, None, None, None, None, None, None

    @staticmethod
    def _backward_group(x, gy, y, gx, gres, vec, packed, k, want_w, act, slope, res_first, group, exchange, one_launch=None):
        lib = hip.load()
        st = hip._stream()
        geo = _Geom(x)
        C = geo.C
        dev = x.device
        a = geo.args()
        a.x, a.x_cs = x.data_ptr(), geo.like(x, x, "x")
        a.gy, a.gy_cs = gy.data_ptr(), geo.like(x, gy, "gy")
        if y is not None:
            a.y, a.y_cs = y.data_ptr(), geo.like(x, y, "y")
        a.mean, a.invstd, a.a, a.b = (vec[i].data_ptr() for i in range(4))
        a.act, a.slope, a.res_first = act, slope, 1 if res_first else 0
        gw_p, gb_p = (k[3].data_ptr(), k[4].data_ptr()) if want_w else (None, None)
        gcs = geo.like(x, gx, "gx") if geo.layout == 0 else 0
        a.out, a.out_cs = gx.data_ptr(), gcs
        a.cw = min(hip.round_up(C, 8), gcs) if geo.layout == 0 else 0
        if gres is not None:
            a.out2, a.out2_cs = gres.data_ptr(), (geo.like(x, gres, "gres") if geo.layout == 0 else 0)
        small = bool(lib.occd_bn_small_ok(ctypes.byref(a)))
        if exchange:
            small = _BNActFn._agreed_small(small, one_launch, x)
        if small and exchange:
            from .shard import channel_exchange
            xc = channel_exchange(group, C, dev)
            if xc is None:
                raise RuntimeError("bn_act: the forward of this layer used the in-kernel exchange, the backward finds no "
                                   "peer-memory exchange installed for its process group")
            hip._check(lib.occd_bn_bwd_small_xchg(ctypes.byref(a), packed.data_ptr(), gw_p, gb_p, *xc, st),
                       "occd_bn_bwd_small_xchg")
            return
        if small:
            hip._check(lib.occd_bn_bwd_small(ctypes.byref(a), gw_p, gb_p, st), "occd_bn_bwd_small")
            return
        a.nblk = lib.occd_bn_blocks(ctypes.byref(a))
        partial = torch.empty(a.nblk * 2 * hip.round_up(C, 4), device=dev, dtype=torch.float32)
        a.partial = partial.data_ptr()
        hip._check(lib.occd_bn_bwd_reduce(ctypes.byref(a), st), "occd_bn_bwd_reduce")
        kp = [k[i].data_ptr() for i in range(3)]
        if exchange:
            import torch.distributed as dist
            local = torch.empty(2 * C, device=dev, dtype=torch.float32)
            hip._check(lib.occd_bn_bwd_combine(partial.data_ptr(), a.nblk, C, local.data_ptr(), st), "occd_bn_bwd_combine")
            total = local.clone()
            from .shard import packed_all_reduce
            packed_all_reduce(total, group)
            hip._check(lib.occd_bn_bwd_finish(local.data_ptr(), total.data_ptr(), C, packed.data_ptr(), vec[0].data_ptr(),
                                              vec[1].data_ptr(), vec[2].data_ptr(), *kp, gw_p, gb_p, st), "occd_bn_bwd_finish")
        else:
            hip._check(lib.occd_bn_bwd_combine_finish(partial.data_ptr(), a.nblk, C, packed.data_ptr(), vec[0].data_ptr(),
                                                      vec[1].data_ptr(), vec[2].data_ptr(), *kp, gw_p, gb_p, st),
                       "occd_bn_bwd_combine_finish")
        a.k1, a.k2, a.k3 = kp
        hip._check(lib.occd_bn_bwd_apply(ctypes.byref(a), st), "occd_bn_bwd_apply")


def _torch_reference(bn, x, act, slope, res, res_first):
    y = bn(x)
    if res is not None and res_first:
        y = y + res
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = y * torch.sigmoid(y)
    elif act == 3:
        y = F.leaky_relu(y, slope)
    if res is not None and not res_first:
        y = y + res
    return y


# One-launch synchronised small layers (the in-kernel channel exchange); "0" keeps every synchronised layer on the packed
# all-reduce (five launches), whatever the tensor sizes.
ONE_LAUNCH_SYNC = __import__("os").environ.get("OCCDEPTH_SYNCBN_ONE_LAUNCH", "1") == "1"
# ... and only for layers of at most this many channels.  The one-launch kernel is one workgroup per channel and every
# workgroup waits for its peers' packets: each rank must be able to keep its workgroups resident while the peers' run, which
# one GPU per rank guarantees (a GPU holds 2048 such workgroups and dispatches them in channel order).  SEVERAL ranks sharing
# ONE GPU (the two-process tests) share those 2048 slots: a 2304-channel launch of one rank can occupy all of them and starve
# the peer it waits for -- such set-ups cap the layer width here (tests: 512).
ONE_LAUNCH_SYNC_MAX_C = int(__import__("os").environ.get("OCCDEPTH_SYNCBN_ONE_LAUNCH_MAX_C", "4096"))

ENABLED = True      # A/B switch (bench / tests): False sends every site through the backend's batch_norm again

# Statistics groups along the batch dimension (view-batched training, models/OccDepth.py process_rgbs): the reference runs
# the 2-D network once per stereo view, so every BatchNorm sees ONE view's samples per call and updates its running
# statistics once per view, in view order.  With the views stacked view-major into one batch, `view_groups(V)` makes every
# bn_act site normalise each of the V contiguous batch chunks with its own statistics and apply the V running-statistics
# updates in the same order -- the same numbers, half the launches of everything that is not a BatchNorm.
GROUPS = 1


class view_groups:
    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        global GROUPS
        self.saved, GROUPS = GROUPS, self.n

    def __exit__(self, *exc):
        global GROUPS
        GROUPS = self.saved
        return False


def module_call(bn, x):
    """`bn(x)` for a BatchNorm module called directly by model code: group-aware like bn_act."""
    if GROUPS > 1 and bn.training:
        return bn_act(bn, x)
    return bn(x)


def bn_act(bn, x, act=None, slope=0.01, res=None, res_first=False):
    """act(bn(x) [+ res]) [+ res] for a BatchNorm module; fused HIP passes in training mode on the GPU (see module doc)."""
    code = ACT[act] if not isinstance(act, int) else act
    fused = (ENABLED and bn.training and x.is_cuda and _Geom.supported(x) and bn.momentum is not None
             and bn.track_running_stats and (res is None or (res.shape == x.shape and res.dtype == x.dtype))
             and not (code == 2 and res is not None and res_first))
    if fused and res is not None:
        # the residual must be addressable in x's layout
        res = res.contiguous() if (x.is_contiguous() and x.dtype == torch.float32) else _to_rows(res)
    G = GROUPS if bn.training else 1
    if not fused:
        if G > 1:                                    # per-view statistics on the backend: chunk, normalise, concatenate
            n = x.shape[0] // G
            return torch.cat([_torch_reference(bn, x[g * n:(g + 1) * n], code, slope,
                                               res[g * n:(g + 1) * n] if res is not None else None, res_first)
                              for g in range(G)], 0)
        return _torch_reference(bn, x, code, slope, res, res_first)
    sync, group, proto = _sync_protocol(bn, x)
    return _BNActFn.apply(x, bn.weight, bn.bias, res, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps,
                          bn.momentum, code, slope, res_first, group, sync, G, proto)


def is_sync(bn):
    """Statistics over the ranks of a process group?  Both converters produce such modules: `shard.convert_sync_batchnorm`
    (prepare_for_ddp) and `torch.nn.SyncBatchNorm.convert_sync_batchnorm` -- what Lightning's `Trainer(sync_batchnorm=True)`
    applies to the model in the reference's unmodified scripts/train.py:175-206."""
    from . import shard
    return isinstance(bn, (shard.SyncBatchNorm, torch.nn.SyncBatchNorm))


def _sync_protocol(bn, x):
    """(sync, process group, protocol record) of a fused training-mode call.  For a synchronised module on an active group:
    makes sure the peer-memory exchange of the group exists (lazily, collectively, once -- a model converted by torch's /
    Lightning's converter never went through prepare_for_ddp) and fixes the layer's exchange protocol ONCE, at its first
    call, from what ALL ranks report (`occd_bn_small_ok` of the local tensor, MIN over the group): the record lives on the
    module (`_occd_sync_proto`, not part of the state_dict) and every later call on every rank follows it."""
    if not is_sync(bn):
        return False, None, None
    group = getattr(bn, "process_group", None)
    if not _group_active(group):
        return True, group, None
    from . import shard
    sm = shard.ensure_small_all_reduce(group, x.device)
    proto = bn.__dict__.get("_occd_sync_proto")
    if proto is None or proto["installed"] != (sm is not None):
        one = False
        if (sm is not None and ONE_LAUNCH_SYNC and x.shape[1] <= ONE_LAUNCH_SYNC_MAX_C
                and sm.channel_args(x.shape[1], x.device) is not None):
            n = x.shape[0] // (GROUPS if bn.training else 1)
            probe = _Geom(x[:n] if n != x.shape[0] else x)
            a = probe.args()
            a.x = x.data_ptr()
            one = bool(hip.load().occd_bn_small_ok(ctypes.byref(a)))
        if sm is not None and ONE_LAUNCH_SYNC:
            one = shard.agree_flag(one, group, x.device)
        proto = {"one_launch": one, "installed": sm is not None}
        bn.__dict__["_occd_sync_proto"] = proto
    return True, group, proto
