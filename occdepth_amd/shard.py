"""Batch-data-parallel sharding of forward frames: frames are independent in eval mode,
so rank r of W processes frames {r, r+W, ...}; no data-path collective exists.  The only communication is
the benchmark's barrier and the MAX-reduction of the per-rank wall time."""
import torch


def frames_for_rank(n_frames, rank, world):
    return list(range(rank, n_frames, world))


def max_over_ranks(seconds, dist=None, device="cpu"):
    """Largest per-rank elapsed time (what bounds whole-job throughput)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def fence(dist=None):
    """barrier + device sync on both sides of a timed region."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def gather_frames(local_results, n_frames, dist=None):
    """Reassemble {frame_index: tensor} dicts from all ranks on every rank (tests / offline inference)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_results[i] for i in range(n_frames)]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local_results)
    merged = {}
    for p in parts:
        merged.update(p)
    return [merged[i] for i in range(n_frames)]


# ---------------------------------------------------------------------------------------------------------------
# Training step ("collective (training step only)"): gradient all-reduce over RCCL / xGMI.
class GradBuckets:
    """Bucketed, backward-overlapped gradient averaging for one process per GPU.

    The reference trains with Lightning DDP (`scripts/train.py:176-206`, `accelerator="ddp"`), i.e. NCCL's
    25 MB buckets tuned for NVSwitch.  On MI355X the exchange runs over point-to-point xGMI links (per-link
    bound), so few LARGE buckets win: the default 128 MiB gives ~5 collectives for the ~600 MB of fp32 gradients of
    the config-2 model.  Parameters are bucketed in reverse registration order (the order backward produces them).

    * Zero-copy: every `p.grad` IS a view into its bucket's flat buffer (`attach()`), autograd accumulates into it in
      place, the collective runs in place on the flat buffer, and the optimizer reads the averaged gradient through
      the same view -- no gather pass before and no scatter pass after the collective.  `zero_grad()` here is one
      memset per bucket.  If a caller detaches a gradient anyway (`model.zero_grad(set_to_none=True)`), the hook
      copies the fresh gradient into the view and re-attaches it (correct, one extra pass for that tensor).
    * Overlap: a bucket's collective is launched asynchronously from the autograd hook of its last gradient, always
      in bucket order, so every rank issues identical collectives and communication overlaps the rest of backward.
    * Unused parameters (the reference needs `find_unused_parameters=True`) contribute the zeros their view holds.
    * Gradient accumulation: wrap all but the last micro-step in `no_sync()`; the views keep accumulating and only
      the last backward counts down and launches.  A second gradient for the same parameter inside one synchronised
      backward raises instead of silently dropping it.
    * `algo="all_reduce"` (default; RCCL picks ring / direct) or `"rs_ag"`: reduce-scatter + all-gather on the padded
      flat buffer, the two-phase form that drives all 7 xGMI links of a fully connected 8-GPU node at once
      (~1 ms instead of ~7 ms per 600 MB when the ring is per-link bound).  gloo (CPU tests)
      only has all-reduce.
    """

    def __init__(self, params, dist, bucket_bytes=128 << 20, algo=None):
        import os
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.algo = algo or os.environ.get("OCCDEPTH_GRAD_ALGO", "all_reduce")
        if self.algo not in ("all_reduce", "rs_ag"):
            raise ValueError(f"unknown gradient exchange algorithm {self.algo!r}")
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []                       # each: dict(flat, items=[(param, offset, numel)], pending, handles)
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._where = {}
        for b in self.buckets:
            for p, off, n in b["items"]:
                self._where[p] = (b, off, n)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._sync = True
        self.attach()
        self.reset()

    def _close(self, plist):
        total = sum(p.numel() for p in plist)
        padded = -(-total // max(self.world, 1)) * max(self.world, 1)        # rs_ag shards the flat buffer evenly
        flat = torch.zeros(padded, dtype=plist[0].dtype, device=plist[0].device)
        items, off = [], 0
        for p in plist:
            items.append((p, off, p.numel()))
            off += p.numel()
        self.buckets.append({"flat": flat, "items": items, "pending": len(items), "handles": [], "seen": set(), "touched": set()})

    def attach(self):
        """Point every p.grad at its slice of the flat buffers (keeps the contents of existing gradients)."""
        for b in self.buckets:
            for p, off, n in b["items"]:
                view = b["flat"][off:off + n].view_as(p)
                if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                p.grad = view

    def zero_grad(self):
        """One memset per bucket; gradients stay attached."""
        for b in self.buckets:
            b["flat"].zero_()
        self.attach()

    def reset(self):
        for b in self.buckets:
            b["pending"], b["handles"] = len(b["items"]), []
            b["seen"], b["touched"] = set(), set()
        self._next = 0

    def no_sync(self):
        """Context manager for the non-final micro-steps of gradient accumulation (no countdown, no collective)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = old
        return ctx()

    def _launch(self, b):
        if self.world <= 1:
            return
        d, flat = self.dist, b["flat"]
        if self.algo == "rs_ag":
            n = flat.numel() // self.world
            shard = flat[d.get_rank() * n:(d.get_rank() + 1) * n]
            b["handles"] = [d.reduce_scatter_tensor(shard, flat, op=d.ReduceOp.SUM, async_op=True),
                            d.all_gather_into_tensor(flat, shard, async_op=True)]    # same stream: ordered
        else:
            b["handles"] = [d.all_reduce(flat, op=d.ReduceOp.SUM, async_op=True)]

    def _on_grad(self, p):
        b, off, n = self._where[p]
        view = b["flat"][off:off + n].view_as(p)
        if p.grad.data_ptr() != view.data_ptr():          # the caller detached the gradient (set_to_none): adopt it
            if p in b["touched"]:
                view.add_(p.grad)
            else:
                view.copy_(p.grad)
            p.grad = view
        b["touched"].add(p)
        if not self._sync:
            return
        if p in b["seen"]:
            raise RuntimeError("GradBuckets: a second gradient arrived for a parameter inside one synchronised "
                               "backward; wrap the non-final micro-steps of gradient accumulation in no_sync()")
        b["seen"].add(p)
        b["pending"] -= 1
        # collectives must be issued in the same order on every rank: bucket i only after buckets 0 .. i-1
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def finish(self):
        """Call after the (last) backward(): completes all buckets; every p.grad then holds the rank average.
        Parameters that received no gradient on this rank contribute zeros; afterwards their `.grad` is None again
        (what DDP with `find_unused_parameters=True` leaves for the reference's never-executed branches, so AdamW
        skips them exactly as it does there -- every rank runs the same graph, so local == global here)."""
        unused = []
        for b in self.buckets[self._next:]:      # bucket order == launch order on every rank
            for p, off, n in b["items"]:
                if p not in b["touched"]:
                    b["flat"][off:off + n].zero_()
                    unused.append(p)
            self._launch(b)
        for b in self.buckets:
            for h in b["handles"]:
                h.wait()
            if self.world > 1:
                b["flat"].div_(self.world)
        for p in unused:
            p.grad = None
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()


def allreduce_confusion(hist, dist=None):
    """Sum the (C, C) int64 confusion matrices of loss/sscMetrics.SSCMetrics over the ranks (in place).
    `hist` is the matrix or the SSCMetrics object; `dist` defaults to torch.distributed when it is initialised."""
    if dist is None:
        import torch.distributed as dist
        if not dist.is_available():
            return hist
    if hasattr(hist, "hist"):
        if hist.hist is None:
            hist._alloc(torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
        allreduce_confusion(hist.hist, dist)
        return hist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist


class _SyncBNFn(torch.autograd.Function):
    """Training-mode batch normalisation over the frames of ALL ranks with ONE packed collective per direction:
    forward all-reduces [sum(x - c), sum((x - c)^2), count] (2C + 1 floats, c = the running mean every rank shares, so
    the single-pass variance does not cancel), backward all-reduces [sum(gy), sum(gy * xhat)] (2C floats).
    torch.nn.SyncBatchNorm does the same job with an all_gather of per-rank (mean, invstd, count) plus a CUDA-only
    combine kernel; this form is backend-agnostic (RCCL on the GPU, gloo in the CPU tests) and is the host side the
    fused BN-statistics convolution epilogue plugs into."""

    @staticmethod
    def forward(ctx, x, weight, bias, centre, eps, group):
        import torch.distributed as dist
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        shape = [1, C] + [1] * (x.dim() - 2)
        acc = torch.float64 if x.dtype == torch.float64 else torch.float32     # bf16 / fp16 activations: fp32 statistics
        centre = centre.to(acc)
        xc = x.to(acc) - centre.view(shape)
        packed = torch.cat([xc.sum(dims), (xc * xc).sum(dims), xc.new_full((1,), float(x.numel() // C))]).double()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        n = packed[2 * C]
        m1 = packed[:C] / n
        var = (packed[C:2 * C] / n - m1 * m1).clamp_min(0.0)
        mean = (m1 + centre.double()).to(acc)
        invstd = torch.rsqrt(var.to(acc) + eps)
        xhat = (x.to(acc) - mean.view(shape)) * invstd.view(shape)
        y = xhat
        if weight is not None:
            y = y * weight.to(acc).view(shape) + bias.to(acc).view(shape)
        ctx.save_for_backward(xhat, invstd, weight)
        ctx.group, ctx.n = group, float(n)
        ctx.mark_non_differentiable(mean, var)
        return y.to(x.dtype), mean, var.to(acc), n.to(acc)

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gn):
        import torch.distributed as dist
        xhat, invstd, weight = ctx.saved_tensors
        C = xhat.shape[1]
        dims = [0] + list(range(2, xhat.dim()))
        shape = [1, C] + [1] * (xhat.dim() - 2)
        g = gy.to(xhat.dtype)
        sum_dy, sum_dy_xhat = g.sum(dims), (g * xhat).sum(dims)
        gw = sum_dy_xhat.clone() if weight is not None else None      # parameter gradients stay per-rank sums:
        gb = sum_dy.clone() if weight is not None else None           # the gradient buckets average them like any other
        packed = torch.cat([sum_dy, sum_dy_xhat]).double()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(ctx.group) > 1:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=ctx.group)
        mean_dy = (packed[:C] / ctx.n).to(xhat.dtype).view(shape)
        mean_dy_xhat = (packed[C:] / ctx.n).to(xhat.dtype).view(shape)
        scale = invstd if weight is None else invstd * weight.to(xhat.dtype)
        gx = (g - mean_dy - xhat * mean_dy_xhat) * scale.view(shape)
        return gx.to(gy.dtype), gw, gb, None, None, None


class SyncBatchNorm(torch.nn.modules.batchnorm._BatchNorm):
    """Drop-in for BatchNorm{1,2,3}d (same parameters / buffers / state_dict keys) whose training-mode statistics span
    every rank's frames -- the reference's `Trainer(sync_batchnorm=True)` (scripts/train.py:179,195).  Eval mode is
    the plain running-statistics affine (and the eval-path plans fold it exactly like nn.BatchNorm)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.process_group = process_group

    def _check_input_dim(self, x):
        if x.dim() < 2:
            raise ValueError(f"expected at least 2D input (got {x.dim()}D input)")

    def forward(self, x):
        self._check_input_dim(x)
        if not self.training and self.track_running_stats:
            return torch.nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                                                  False, 0.0, self.eps)
        centre = self.running_mean.detach() if self.running_mean is not None else \
            x.new_zeros(self.num_features, dtype=torch.float32)
        y, mean, var, n = _SyncBNFn.apply(x, self.weight, self.bias, centre, self.eps, self.process_group)
        if self.training and self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                unbiased = var * (n / (n - 1).clamp_min(1.0))
                self.running_mean.mul_(1 - mom).add_(mean.to(self.running_mean.dtype), alpha=mom)
                self.running_var.mul_(1 - mom).add_(unbiased.to(self.running_var.dtype), alpha=mom)
        return y


def convert_sync_batchnorm(module, process_group=None):
    """Replace every BatchNorm{1,2,3}d by `SyncBatchNorm` in place of the attribute (parameters and buffers are shared,
    state_dict keys unchanged)."""
    out = module
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(module, SyncBatchNorm):
        out = SyncBatchNorm(module.num_features, module.eps, module.momentum, module.affine,
                            module.track_running_stats, process_group)
        if module.affine:
            out.weight, out.bias = module.weight, module.bias
        out.running_mean, out.running_var = module.running_mean, module.running_var
        out.num_batches_tracked = module.num_batches_tracked
        out.training = module.training
    for name, child in module.named_children():
        new = convert_sync_batchnorm(child, process_group)
        if new is not child:
            setattr(out, name, new)
    return out


def prepare_for_ddp(model, dist=None, bucket_bytes=128 << 20, sync_bn=True):
    """What `Trainer(accelerator="ddp", sync_batchnorm=True)` does for the reference (scripts/train.py:176-206),
    without Lightning: BatchNorm -> SyncBatchNorm (statistics over the frames of all ranks; with 1 frame per GPU the
    per-rank statistics would otherwise be those of a single scene) and the gradient buckets.
    Returns (model, GradBuckets or None)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return model, None
    if sync_bn:
        model = convert_sync_batchnorm(model)
    return model, GradBuckets(model.parameters(), dist, bucket_bytes)
