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

    def __init__(self, params, dist, bucket_bytes=128 << 20, algo=None, force=False, check_used=None, comm_dtype=None):
        import os
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        # `force`: issue the collectives on a single-rank group too (exercises the RCCL path on one GPU)
        self.active = dist is not None and dist.is_initialized() and (self.world > 1 or force)
        self.algo = algo or os.environ.get("OCCDEPTH_GRAD_ALGO", "all_reduce")
        if self.algo not in ("all_reduce", "rs_ag"):
            raise ValueError(f"unknown gradient exchange algorithm {self.algo!r} (expected 'all_reduce' or 'rs_ag')")
        # the average comes out of the collective itself where the backend can (RCCL: ReduceOp.AVG), so no extra pass
        # over the ~600 MB of gradients follows it; gloo has no AVG: SUM, then one division per bucket
        self.avg_in_collective = self.active and dist.get_backend() == "nccl"
        self.check_used = (os.environ.get("OCCDEPTH_DDP_CHECK", "0") == "1") if check_used is None else bool(check_used)
        # wire format of the exchange: None = the gradients' own dtype (float32); torch.bfloat16 (OCCDEPTH_GRAD_COMM=bf16) halves
        # the bytes on xGMI -- a bucket is cast into a bf16 staging buffer when it launches, reduced there, and cast back in
        # finish() (two extra passes over the bucket on HBM, ~0.3 ms per 600 MB, against 300 MB less per link); the gradients
        # autograd and the optimizer see stay float32
        if comm_dtype is None and os.environ.get("OCCDEPTH_GRAD_COMM", "") in ("bf16", "bfloat16"):
            comm_dtype = torch.bfloat16
        self.comm_dtype = comm_dtype
        # Which parameters receive a gradient is learned from the first synchronised backward (`_expected`): a parameter of a
        # never-executed branch (the reference needs find_unused_parameters=True) would otherwise keep its bucket -- and, because
        # buckets launch in order, EVERY bucket -- pending until finish(), i.e. no collective would overlap the backward at all
        # (round 5 measured exactly that: +13.5 % for the forced single-rank exchange).  From the second step on a bucket counts
        # down over the parameters that did receive one; see `_on_grad` for what happens when the set changes.
        self._expected = None
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
        self.buckets.append({"flat": flat, "items": items, "pending": len(items), "handles": [], "seen": set(), "touched": set(),
                             "comm": None, "launched": False})

    def attach(self):
        """Point every p.grad at its slice of the flat buffers (keeps the contents of existing gradients)."""
        for b in self.buckets:
            for p, off, n in b["items"]:
                view = b["flat"][off:off + n].view_as(p)
                if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                p.grad = view

    def release(self):
        """Undo what `prepare_for_ddp(force=True)` switched on process-wide (single-rank collectives): call it when the
        forced run is over so that later SyncBatchNorm / bn_act calls on single-rank groups skip their collectives again."""
        global FORCE_COLLECTIVES
        if getattr(self, "_restore_force", None) is not None:
            FORCE_COLLECTIVES = self._restore_force
            self._restore_force = None

    def zero_grad(self):
        """One memset per bucket; gradients stay attached."""
        for b in self.buckets:
            b["flat"].zero_()
        self.attach()

    def reset(self):
        for b in self.buckets:
            n = len(b["items"]) if self._expected is None else sum(1 for p, _, _ in b["items"] if p in self._expected)
            b["pending"], b["handles"], b["launched"] = n, [], False
            b["seen"], b["touched"] = set(), set()
        self._next = 0

    def relearn(self):
        """Forget which parameters receive gradients (call after changing what the forward executes)."""
        self._expected = None
        self.reset()

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

    def _launch_ready(self):
        # collectives must be issued in the same order on every rank: bucket i only after buckets 0 .. i-1
        # (a parameter the learned set does not expect holds zeros in its view: zero_grad() / the previous exchange of zeros)
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b):
        b["launched"] = True
        if not self.active:
            return
        d, flat = self.dist, b["flat"]
        if self.comm_dtype is not None and flat.dtype != self.comm_dtype:
            if b["comm"] is None:
                b["comm"] = torch.empty_like(flat, dtype=self.comm_dtype)
            b["comm"].copy_(flat)
            flat = b["comm"]
        op = d.ReduceOp.AVG if self.avg_in_collective else d.ReduceOp.SUM
        if self.algo == "rs_ag":
            n = flat.numel() // self.world
            shard = flat[d.get_rank() * n:(d.get_rank() + 1) * n]
            b["handles"] = [d.reduce_scatter_tensor(shard, flat, op=op, async_op=True),
                            d.all_gather_into_tensor(flat, shard, async_op=True)]    # same stream: ordered
        else:
            b["handles"] = [d.all_reduce(flat, op=op, async_op=True)]

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
        if self._expected is not None and p not in self._expected:
            # the executed graph grew: a parameter that never had a gradient now has one.  If its bucket's collective is
            # already on its way the gradient would be lost on the wire -- refuse; otherwise it simply rides along.
            if b["launched"]:
                raise RuntimeError("GradBuckets: a parameter outside the learned set received a gradient after its bucket was "
                                   "exchanged (the forward now executes other branches); call buckets.relearn() when the "
                                   "executed graph changes")
            return
        b["pending"] -= 1
        self._launch_ready()

    def finish(self):
        """Call after the (last) backward(): completes all buckets; every p.grad then holds the rank average.
        Parameters that received no gradient on this rank contribute zeros; afterwards their `.grad` is None again
        (what DDP with `find_unused_parameters=True` leaves for the reference's never-executed branches, so AdamW
        skips them exactly as it does there -- every rank runs the same graph, so local == global here)."""
        unused = []
        for b in self.buckets[:self._next]:      # launched from the hooks: their untouched views hold the zeros of zero_grad()
            for p, off, n in b["items"]:
                if p not in b["touched"]:
                    unused.append(p)
        for b in self.buckets[self._next:]:      # bucket order == launch order on every rank
            for p, off, n in b["items"]:
                if p not in b["touched"]:
                    b["flat"][off:off + n].zero_()
                    unused.append(p)
            self._launch(b)
        self._next = len(self.buckets)
        for b in self.buckets:
            for h in b["handles"]:
                h.wait()
            if self.active and b["comm"] is not None and b["launched"]:
                b["flat"].copy_(b["comm"])       # back to the gradients' dtype (the views the optimizer reads)
            if self.active and self.world > 1 and not self.avg_in_collective:
                b["flat"].div_(self.world)
        if self._expected is None and self._sync:
            self._expected = set()
            for b in self.buckets:
                self._expected |= b["touched"]
        if self.check_used and self.active:
            self._check_used_sets()
        poll_exchanges()                         # a peer-memory SyncBatchNorm exchange that gave up raises here (no host sync)
        for p in unused:
            p.grad = None
        self.reset()

    def _check_used_sets(self):
        """Debug (OCCDEPTH_DDP_CHECK=1): `finish()` drops the gradient of parameters this rank did not touch from LOCAL
        knowledge, which is only right when every rank ran the same graph (one frame schema per step, as in the
        reference's runs).  If a batch-dependent branch (`'occluded' in batch`, a single-view `gt_depth` frame) differs
        between ranks, a rank would skip AdamW for a parameter the others update: raise instead of diverging silently."""
        flags = torch.tensor([1 if p in b["touched"] else 0 for b in self.buckets for p, _, _ in b["items"]],
                             dtype=torch.int32, device=self.buckets[0]["flat"].device)
        lo, hi = flags.clone(), flags.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            bad = int((lo != hi).sum())
            raise RuntimeError(f"GradBuckets: {bad} parameter(s) received a gradient on some ranks only (the ranks ran "
                               "different graphs); their replicas would diverge")

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


# ---------------------------------------------------------------------------------------------------------------
# Small-message all-reduce over peer-mapped device memory (csrc/ipc_allreduce.hip): SyncBatchNorm's per-layer packets.
class SmallAllReduce:
    """One-kernel, library-free, deterministic all-reduce (SUM) for vectors of up to `max_bytes` among the ranks of `group`
    on ONE node: every rank owns a mailbox in its HBM, exported through hipIpcGetMemHandle and mapped by all peers once;
    a call pushes the rank's vector into every mailbox, waits for all flags, sums the rows in rank order (see the kernel
    file).  532 of these per config-2 training step replace RCCL launches of ~45 us each that sit on one dependency chain.
    Capturable into a hipGraph.  The IPC handles travel through `dist.all_gather_object` on `group` (any backend).

    Failure behaviour (what a collective library gives, kept):
      * set-up is AGREED: the ranks first exchange (hostname, ok) and then the outcome of create / open, so either every
        rank of the group ends up with the exchange installed or every rank raises the same RuntimeError (and has freed
        what it created / closed what it opened) -- never some ranks inside IPC and others on the process group;
      * the in-kernel wait is bounded by `timeout_ms` (default: OCCDEPTH_IPC_TIMEOUT_MS, else 30 minutes -- the process
        group's default collective timeout; <= 0 waits forever).  A wait that gives up POISONS its result: the reduced
        vector / the layer's statistics become NaN, so the loss is NaN from that step on instead of a silently wrong
        normalisation, and the sticky device flag makes `check()` / `poll()` raise.  `poll()` is free of host
        synchronisation (an asynchronous copy of the flag, examined at the NEXT call) and is what the training path calls
        once per step (GradBuckets.finish, the model's on_train_batch_end hook); `check()` synchronises (tests, bench)."""

    MAX_WORLD = 16

    CHANNELS = 4096          # channel capacity of the second mailbox (SyncBatchNorm's in-kernel exchange, csrc/bn.hip BnXchg)

    _inject_failure = None   # tests: ("create" | "open", rank) makes that stage fail on that rank only

    def __init__(self, dist, group=None, max_bytes=64 * 1024, device=None, timeout_ms=None):
        import ctypes
        import os
        import socket
        from . import hip
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.max_bytes = int(max_bytes)
        if timeout_ms is None:
            timeout_ms = int(os.environ.get("OCCDEPTH_IPC_TIMEOUT_MS", str(30 * 60 * 1000)))
        self.timeout_ms = max(0, min(int(timeout_ms), 0x7fffffff))
        self._owned, self._opened, self._own = [], [], None
        self._pending = None

        def agree(ok, what):
            """Every rank learns every rank's outcome of a set-up stage; all raise together when any failed."""
            flags = [None] * self.world
            dist.all_gather_object(flags, (bool(ok), str(what)), group=group)
            bad = [(r, w) for r, (o, w) in enumerate(flags) if not o]
            if bad:
                self._release()
                raise RuntimeError("SmallAllReduce: set-up failed on rank(s) " + ", ".join(f"{r} ({w})" for r, w in bad))

        # stage 0: one node, a supported world size, a GPU and a loadable library -- decided from what ALL ranks report
        lib, err = None, ""
        try:
            self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            if self.device.type != "cuda" or not torch.cuda.is_available():
                raise RuntimeError("no GPU on this rank")
            lib = hip.load()
        except Exception as e:                      # (the product path fails loudly elsewhere; here the group must agree first)
            err = repr(e)
        hosts = [None] * self.world
        dist.all_gather_object(hosts, (socket.gethostname(), lib is not None, err), group=group)
        if self.world > self.MAX_WORLD:
            raise RuntimeError(f"SmallAllReduce: world size {self.world} > {self.MAX_WORLD}")
        if len({h for h, _, _ in hosts}) != 1:
            raise RuntimeError("SmallAllReduce: the group spans several hosts (" + ", ".join(sorted({h for h, _, _ in hosts}))
                               + "): peer-mapped device memory needs one node")
        if not all(ok for _, ok, _ in hosts):
            raise RuntimeError("SmallAllReduce: not every rank can take part: "
                               + "; ".join(f"rank {r}: {e}" for r, (_, ok, e) in enumerate(hosts) if not ok))
        self._lib = lib
        nbytes = lib.occd_ipc_mailbox_bytes(self.world, self.max_bytes)
        cbytes = lib.occd_bn_xchg_mailbox_bytes(self.world, self.CHANNELS)
        if nbytes <= 0 or cbytes <= 0:              # (same arguments on every rank: same outcome on every rank)
            raise RuntimeError("occd_ipc_mailbox_bytes failed")

        # stage 1: create both mailboxes locally, exchange the handles together with the outcome
        owns, handles, err = [], [], ""
        with torch.cuda.device(self.device):
            try:
                if self._inject_failure == ("create", self.rank):
                    raise RuntimeError("injected failure (test)")
                for nb in (nbytes, cbytes):
                    own = ctypes.c_void_p()
                    handle = (ctypes.c_ubyte * 64)()
                    hip._check(lib.occd_ipc_mailbox_create(nb, ctypes.byref(own), handle), "occd_ipc_mailbox_create")
                    self._owned.append(own.value)
                    owns.append(own.value)
                    handles.append(bytes(handle))
            except Exception as e:
                err = repr(e)
            gathered = [None] * self.world
            dist.all_gather_object(gathered, (err == "", err, handles), group=group)
            bad = [(r, e) for r, (ok, e, _) in enumerate(gathered) if not ok]
            if bad:
                self._release()
                raise RuntimeError("SmallAllReduce: mailbox allocation failed on rank(s) " + ", ".join(f"{r} ({e})" for r, e in bad))
            # stage 2: map every peer's two mailboxes; agree on the outcome before anybody pushes
            ptr_sets, err = [], ""
            try:
                if self._inject_failure == ("open", self.rank):
                    raise RuntimeError("injected failure (test)")
                for k in range(2):
                    ptrs = (ctypes.c_void_p * self.world)()
                    for r in range(self.world):
                        if r == self.rank:
                            ptrs[r] = owns[k]
                            continue
                        buf = (ctypes.c_ubyte * 64).from_buffer_copy(gathered[r][2][k])
                        peer = ctypes.c_void_p()
                        hip._check(lib.occd_ipc_mailbox_open(buf, ctypes.byref(peer)), "occd_ipc_mailbox_open")
                        ptrs[r] = peer.value
                        self._opened.append(peer.value)
                    ptr_sets.append(ptrs)
            except Exception as e:
                err = repr(e)
            agree(err == "", err or "ok")
            self._ptrs, self._cptrs = ptr_sets       # vectors (occd_ipc_allreduce); per-channel records (occd_bn_*_small_xchg)
            self._own = owns[0]
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._host_status = torch.zeros(1, dtype=torch.int32).pin_memory()
        dist.barrier(group=group)                 # every mailbox is mapped everywhere before the first push

    def _release(self):
        """Close what was opened, free what was created (set-up failure and close())."""
        lib = getattr(self, "_lib", None)
        if lib is not None:
            for p in self._opened:
                lib.occd_ipc_mailbox_close(p)
            for p in self._owned:
                lib.occd_ipc_mailbox_free(p)
        self._own, self._opened, self._owned = None, [], []

    def all_reduce_(self, t):
        """In-place SUM over the ranks of a contiguous float32 / float64 GPU tensor of <= max_bytes (asynchronous on the
        current stream).  A wait that gives up leaves NaN in `t` (and sets the status flag)."""
        from . import hip
        if not self.usable(t):
            raise RuntimeError("SmallAllReduce: tensor must be a contiguous float32 / float64 tensor of <= max_bytes on the group's device")
        hip._check(self._lib.occd_ipc_allreduce(t.data_ptr(), t.data_ptr(), t.numel(), 0 if t.dtype == torch.float32 else 1,
                                                self._ptrs, self.rank, self.world, self.max_bytes, self.timeout_ms,
                                                self.status.data_ptr(), hip._stream()), "occd_ipc_allreduce")
        return t

    def usable(self, t):
        return (getattr(self, "_own", None) is not None and t.is_cuda and t.device == self.device and t.is_contiguous()
                and t.dtype in (torch.float32, torch.float64) and 0 < t.numel() * t.element_size() <= self.max_bytes)

    _GAVE_UP = ("SmallAllReduce: a peer did not arrive within the time budget; the affected results were set to NaN "
                "(OCCDEPTH_IPC_TIMEOUT_MS raises the budget, OCCDEPTH_SYNCBN_IPC=0 keeps the process group's all_reduce)")

    def check(self):
        """Raise if any exchange since the last check gave up waiting (synchronises the stream)."""
        if int(self.status.item()) != 0:
            self.status.zero_()
            raise RuntimeError(self._GAVE_UP)

    def poll(self):
        """The per-step form of check(): no host synchronisation.  Looks at the flag copy queued by the PREVIOUS poll (if
        that copy has completed) and queues a new one, so a give-up surfaces one step late at the latest -- and the step it
        happened in already carries NaN.  Not callable during stream capture (skipped there)."""
        if getattr(self, "_own", None) is None or torch.cuda.is_current_stream_capturing():
            return
        if self._pending is not None and self._pending.query():
            self._pending = None
            if int(self._host_status[0]) != 0:
                self.status.zero_()
                self._host_status.zero_()
                raise RuntimeError(self._GAVE_UP)
        if self._pending is None:
            self._host_status.copy_(self.status, non_blocking=True)
            self._pending = torch.cuda.Event()
            self._pending.record()

    def close(self):
        if getattr(self, "_own", None) is None:
            return
        torch.cuda.synchronize(self.device)
        try:
            self.dist.barrier(group=self.group)   # nobody is still pushing into a mailbox that is about to go away
        except Exception:
            pass
        self._release()

    def channel_args(self, C, device):
        """(mailboxes, rank, world, cmax, timeout_ms, status) for occd_bn_*_small_xchg, or None when the layer does not fit."""
        if C > self.CHANNELS or torch.device(device) != self.device or getattr(self, "_own", None) is None:
            return None
        return (self._cptrs, self.rank, self.world, self.CHANNELS, self.timeout_ms, self.status.data_ptr())


_SMALL = {}            # process group (None = default) -> SmallAllReduce
_SMALL_TRIED = set()   # groups for which the lazy set-up ran (installed or not): it is attempted once


def install_small_all_reduce(dist, group=None, **kw):
    """Route SyncBatchNorm's packed exchanges of `group` through `SmallAllReduce` (returns it).  Ranks must call this
    collectively; either every rank returns with the exchange installed or every rank raises (see the class).
    `uninstall_small_all_reduce` restores the backend's all_reduce."""
    _SMALL_TRIED.add(group)
    sm = SmallAllReduce(dist, group, **kw)
    _SMALL[group] = sm
    return sm


def uninstall_small_all_reduce(group=None):
    _SMALL_TRIED.discard(group)
    sm = _SMALL.pop(group, None)
    if sm is not None:
        sm.close()


def ensure_small_all_reduce(group=None, device=None):
    """Lazy, collective, once per group: what `prepare_for_ddp` does eagerly, for models that reach synchronised statistics
    through somebody else's conversion -- `torch.nn.SyncBatchNorm.convert_sync_batchnorm`, i.e. Lightning's
    `Trainer(sync_batchnorm=True)` of the reference's scripts/train.py:175-206.  Called from the first synchronised
    `bn_act` of a step, which every rank reaches at the same layer.  OCCDEPTH_SYNCBN_IPC=0, a CPU model, a group of one rank
    (unless forced) or more than 16 ranks keep the process group's all_reduce; so does a set-up that fails (on every rank
    alike, with a warning)."""
    import os
    if group in _SMALL or group in _SMALL_TRIED:
        return _SMALL.get(group)
    import torch.distributed as dist
    if (os.environ.get("OCCDEPTH_SYNCBN_IPC", "1") != "1" or not torch.cuda.is_available() or not dist.is_initialized()
            or dist.get_world_size(group) > SmallAllReduce.MAX_WORLD
            or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES)):
        return None
    _SMALL_TRIED.add(group)         # attempted once per group, whatever the outcome
    try:
        return install_small_all_reduce(dist, group, device=device)
    except Exception as e:          # agreed across the ranks: everybody lands here together
        import warnings
        warnings.warn(f"occdepth_amd: peer-memory all-reduce unavailable ({e!r}); SyncBatchNorm uses the process group")
        return None


def poll_exchanges():
    """Once per training step: raise if a peer-memory exchange gave up (no host synchronisation, see SmallAllReduce.poll)."""
    for sm in list(_SMALL.values()):
        sm.poll()


def channel_exchange(group, C, device):
    """Arguments of the in-kernel per-channel exchange (one-launch synchronised small BatchNorm layers) when a peer-memory
    exchange is installed for `group`, else None."""
    sm = _SMALL.get(group)
    return None if sm is None else sm.channel_args(C, device)


def packed_all_reduce(t, group=None):
    """SUM all-reduce of a SyncBatchNorm statistics packet: the peer-memory kernel when one is installed for the group and
    takes the tensor, the process group's all_reduce otherwise."""
    sm = _SMALL.get(group)
    if sm is not None and sm.usable(t):
        return sm.all_reduce_(t)
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def agree_flag(flag, group=None, device=None):
    """True iff `flag` is true on EVERY rank of `group` (one tiny MIN all-reduce through the process group).  Used once per
    synchronised BatchNorm layer to pick a protocol all ranks then follow, whatever their local tensor shapes."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def _group_active(group):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


# Single-rank process groups normally skip their collectives (there is nothing to exchange); tests and
# `bench.py --train` under OCCDEPTH_FORCE_DIST=1 set this to drive the very same RCCL calls on one GPU.
FORCE_COLLECTIVES = False


def _dense(t):
    """ATen's fused batch-norm kernels take NC* tensors that are dense in the default or in the channels-last order."""
    if t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)) or \
            (t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d)):
        return t
    return t.contiguous()


class _SyncBNFn(torch.autograd.Function):
    """Training-mode batch normalisation over the frames of ALL ranks with ONE packed collective per direction.

    forward : every rank takes its own (mean_r, M2_r / n_r) in one Welford pass (`torch.var_mean`), the ranks' moments are
              merged with Chan's formula -- total mean = sum(n_r mean_r) / n, total M2 = sum(M2_r + n_r mean_r^2) - n mean^2 --
              from ONE all-reduce of the (2C + 1)-element vector [n_r mean_r, M2_r + n_r mean_r^2, n_r].  Only that small
              vector is float64 (so the merge cannot cancel, whatever mean / std is: the per-rank pass is centred on the
              rank's own mean); no full-size float64 copy of the activation is made or saved (bf16 / fp16 inputs are widened to one
              float32 copy for the statistics pass).
    backward: one reduction pass [sum(gy), sum(gy (x - mean))], ONE all-reduce of 2C floats, one element-wise pass.
    On the GPU the two passes per direction are ATen's fused batch-norm kernels (batch_norm_elemt,
    batch_norm_backward_reduce / _elemt -- the ones torch.nn.SyncBatchNorm uses); elsewhere (gloo CPU tests) the same
    arithmetic in plain tensor ops.  The count stays a device tensor: no host synchronisation per layer.
    torch.nn.SyncBatchNorm itself all_gathers (mean, invstd, count) per layer and needs a CUDA-only combine kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        import torch.distributed as dist
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        shape = [1, C] + [1] * (x.dim() - 2)
        acc = torch.float64 if x.dtype == torch.float64 else torch.float32     # bf16 / fp16 activations: fp32 statistics
        x = _dense(x)
        var_l, mean_l = torch.var_mean(x if x.dtype == acc else x.to(acc), dims, correction=0)
        n_l = float(x.numel() // C)
        m64 = mean_l.double()
        packed = torch.cat([m64 * n_l, (var_l.double() + m64 * m64) * n_l, m64.new_full((1,), n_l)])
        if _group_active(group):
            packed_all_reduce(packed, group)
        n = packed[2 * C]
        mean64 = packed[:C] / n
        var64 = (packed[C:2 * C] / n - mean64 * mean64).clamp_min(0.0)
        mean, var = mean64.to(acc), var64.to(acc)
        invstd = torch.rsqrt(var + eps)
        if x.is_cuda and x.dtype != torch.float64:
            y = torch.batch_norm_elemt(x, weight, bias, mean, invstd, eps)
        else:
            scale = invstd if weight is None else invstd * weight.to(acc)
            shift = -mean * scale if bias is None else bias.to(acc) - mean * scale
            y = torch.addcmul(shift.view(shape), x.to(acc), scale.view(shape)).to(x.dtype)
        ctx.save_for_backward(x, mean, invstd, weight, n)
        ctx.group = group
        ctx.mark_non_differentiable(mean, var, n)
        return y, mean, var, n.to(acc)

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gn):
        import torch.distributed as dist
        x, mean, invstd, weight, n = ctx.saved_tensors
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        shape = [1, C] + [1] * (x.dim() - 2)
        acc = mean.dtype
        fused = x.is_cuda and x.dtype != torch.float64
        if fused:
            gy = _dense(gy)
            sum_dy, sum_dy_xmu, gw, gb = torch.batch_norm_backward_reduce(gy, x, mean, invstd, weight, True,
                                                                           weight is not None, weight is not None)
        else:
            g = gy.to(acc)
            xmu = x.to(acc) - mean.view(shape)
            sum_dy, sum_dy_xmu = g.sum(dims), (g * xmu).sum(dims)
            gw = sum_dy_xmu * invstd if weight is not None else None      # parameter gradients stay per-rank sums:
            gb = sum_dy.clone() if weight is not None else None           # the gradient buckets average them like any other
        packed = torch.cat([sum_dy, sum_dy_xmu])
        if _group_active(ctx.group):
            packed_all_reduce(packed, ctx.group)
        if fused:
            gx = torch.batch_norm_backward_elemt(gy, x, mean, invstd, weight, packed[:C], packed[C:],
                                                 n.to(torch.int32).reshape(1))
        else:
            nn_ = n.to(acc)
            mean_dy = (packed[:C] / nn_).view(shape)
            k = (packed[C:] / nn_ * invstd * invstd).view(shape)            # mean(gy (x - mean)) / var
            scale = invstd if weight is None else invstd * weight.to(acc)
            gx = ((g - mean_dy - xmu * k) * scale.view(shape)).to(gy.dtype)
        if weight is not None:
            gw, gb = gw.to(weight.dtype), gb.to(weight.dtype)
        return gx, gw, gb, None, None


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
        if x.is_cuda:
            from . import bn as _bn
            if _bn.ENABLED and _bn._Geom.supported(x) and self.momentum is not None and self.track_running_stats:
                return _bn.bn_act(self, x)                      # K13 passes + the packed all-reduces (csrc/bn.hip)
        y, mean, var, n = _SyncBNFn.apply(x, self.weight, self.bias, self.eps, self.process_group)
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


def prepare_for_ddp(model, dist=None, bucket_bytes=128 << 20, sync_bn=True, force=False, algo=None, grad_buckets=True):
    """What `Trainer(accelerator="ddp", sync_batchnorm=True)` does for the reference (scripts/train.py:176-206),
    without Lightning: BatchNorm -> SyncBatchNorm (statistics over the frames of all ranks; with 1 frame per GPU the
    per-rank statistics would otherwise be those of a single scene) and the gradient buckets.
    Returns (model, GradBuckets or None).  On one rank there is nothing to exchange and the model is returned as is --
    unless `force` (tests, `OCCDEPTH_FORCE_DIST=1 bench.py --train`): then the converted SyncBatchNorm layers and the
    buckets run their collectives on the single-rank group, i.e. the real RCCL calls on one GPU."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return model, None
    global FORCE_COLLECTIVES
    was_forced = FORCE_COLLECTIVES
    if force:
        FORCE_COLLECTIVES = True          # process-wide while these buckets live: `buckets.release()` restores it
    if sync_bn:
        model = convert_sync_batchnorm(model)
        if hasattr(model, "invalidate_graphs"):
            model.invalidate_graphs()     # module surgery: a captured eval graph would keep replaying the old layers
        # SyncBatchNorm's per-layer packets over peer-mapped memory instead of the backend's all_reduce (one node, <= 16
        # ranks, GPU tensors): 532 latency-bound exchanges per config-2 step.  OCCDEPTH_SYNCBN_IPC=0 keeps RCCL.
        if next(model.parameters()).is_cuda:
            ensure_small_all_reduce(None, next(model.parameters()).device)
    if not grad_buckets:                  # (measurement aid: SyncBatchNorm's exchanges alone; FORCE_COLLECTIVES stays as set)
        return model, None
    buckets = GradBuckets(model.parameters(), dist, bucket_bytes, algo=algo, force=force)
    buckets._restore_force = was_forced if force else None
    return model, buckets
