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
    25 MB buckets tuned for NVSwitch.  On MI355X the all-reduce is a ring over point-to-point xGMI links
    (per-link bound), so few LARGE buckets win: the default 128 MiB gives ~5 collectives for the ~600 MB of fp32
    gradients of the config-2 model.  Parameters are bucketed in reverse registration order (the order backward
    produces them); a bucket's all-reduce is launched asynchronously from the autograd hook of its last gradient,
    so communication overlaps the rest of backward.  Parameters that received no gradient this step (the
    reference needs `find_unused_parameters=True` for them) are filled with zeros in `finish()`, so every rank
    issues the same collectives in the same order whatever its graph looked like.
    """

    def __init__(self, params, dist, bucket_bytes=128 << 20):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []                       # each: dict(flat, items=[(param, offset, numel)], pending, handle)
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
        self.reset()

    def _close(self, plist):
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
        items, off = [], 0
        for p in plist:
            items.append((p, off, p.numel()))
            off += p.numel()
        self.buckets.append({"flat": flat, "items": items, "pending": len(items), "handle": None, "seen": set()})

    def reset(self):
        for b in self.buckets:
            b["pending"], b["handle"] = len(b["items"]), None
            b["seen"] = set()
        self._next = 0

    def _launch(self, b):
        if self.world > 1:
            b["handle"] = self.dist.all_reduce(b["flat"], op=self.dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        b, off, n = self._where[p]
        if p in b["seen"]:
            return                               # gradient accumulation: only the first arrival counts down
        b["seen"].add(p)
        b["flat"][off:off + n].copy_(p.grad.reshape(-1))
        b["pending"] -= 1
        # collectives must be issued in the same order on every rank: bucket i only after buckets 0 .. i-1
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def finish(self):
        """Call after backward(): completes all buckets and leaves the averaged gradient in every p.grad."""
        for b in self.buckets[self._next:]:      # bucket order == launch order on every rank
            for p, off, n in b["items"]:
                if p not in b["seen"]:
                    b["flat"][off:off + n].zero_()
            self._launch(b)
        for b in self.buckets:
            if b["handle"] is not None:
                b["handle"].wait()
            if self.world > 1:
                b["flat"].div_(self.world)
            for p, off, n in b["items"]:
                if p in b["seen"] or self.world > 1:
                    g = b["flat"][off:off + n].view_as(p)
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.copy_(g)
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()


def allreduce_confusion(hist, dist=None):
    """Sum the (C, C) int64 confusion matrices of loss/sscMetrics.SSCMetrics over the ranks (in place)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist


def prepare_for_ddp(model, dist=None, bucket_bytes=128 << 20):
    """What `Trainer(accelerator="ddp", sync_batchnorm=True)` does for the reference (scripts/train.py:176-206),
    without Lightning: BatchNorm -> SyncBatchNorm (statistics all-reduced over RCCL; with 1 frame per GPU the
    per-rank statistics would otherwise be those of a single scene) and the gradient buckets.
    Returns (model, GradBuckets or None)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return model, None
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    return model, GradBuckets(model.parameters(), dist, bucket_bytes)
