"""Whole-step hipGraph for training: forward + every loss term + backward + gradient exchange +
optimizer update captured ONCE and replayed per step.

A config-2 training step is ~9600 kernel launches, most of them 4-6 us long (BatchNorm pieces, the elementwise glue of
55 MBConv blocks x 2 views, gradient accumulation): the eager step is HOST-bound (GPU busy 189 ms of a 213 ms step,
profiles/r02_train_step_fp32_kernels.csv).  Replaying a captured graph removes the host from the loop.

Requirements (checked or enforced here):
  * nothing in the step may synchronise with the host (`OccDepth.step` is written sync-free; `find_syncs` below lists
    offenders with their Python stack via torch.cuda.set_sync_debug_mode);
  * the batch tensors are STATIC: copy new data into them (`load_batch`) before each replay;
  * the optimizer must be capturable (device-side step counter): `make_capturable` flips the flag before its first step;
  * the learning rate lives in a DEVICE tensor (`make_capturable` converts `group["lr"]`): torch's schedulers update a
    tensor learning rate in place, so MultiStepLR's two milestones need no re-capture.
What a replay does and does not advance (the captured Python code does not run again):
  * device state -- parameters, optimizer moments and step counters, BatchNorm running statistics, the metric's
    confusion matrix, the loss terms in `model.logged` -- advances, it is what the kernels write;
  * host-side counters are mirrored by `__call__`: `model.cur_batch` and the metric's `count` are incremented per replay,
    and the `sem_step_decay_loss` factor (a function of `cur_batch`) is fed through a device scalar that is refreshed
    before every replay (`OccDepth._decay_dev`);
  * `self.log(...)` (Lightning's logger) is NOT called per replay: read `model.logged` instead.
Warm-up: capture needs the optimizer state allocated and the libraries' solvers chosen, which takes real eager steps.
They run on a SNAPSHOT: parameters, buffers, optimizer state, metric counts and `cur_batch` are restored afterwards, so
capturing (or re-capturing) trains nothing -- the reference's eager Lightning loop has no such extra steps.
Reference: scripts/train.py:176-206 drives the same step through PyTorch-Lightning, eagerly.
"""
import contextlib
import os

import torch

# hipStreamCaptureModeThreadLocal: only the capturing thread's calls are checked, so a process group's watchdog thread
# (event queries) cannot invalidate -- or abort -- a capture running beside it (N > 1 ranks, OCCDEPTH_FORCE_DIST=1)
CAPTURE_MODE = "thread_local"


# ROCm 7.2 / gfx950: a hipMemsetAsync captured into a hipGraph fills with its value on the FIRST launch of the instantiated
# graph only; later launches fill with a stale pattern (tools/probe_graph_memset.py, csrc/graph_fix.hip).  ATen's multi-block
# reductions zero their semaphores with such a node, so a `sum` over many rows -- a convolution's bias gradient -- can come
# back unwritten on replays: the intermittent NaN of the captured step.  Every graph of this package is therefore captured
# with `keep_graph=True`, has its memset nodes rewritten as fill kernels (`occd_graph_replace_memsets`) and is instantiated
# afterwards.  OCCDEPTH_GRAPH_FIX_MEMSETS=0 keeps the captured nodes (A/B, debugging).
FIX_MEMSETS = os.environ.get("OCCDEPTH_GRAPH_FIX_MEMSETS", "1").lower() not in ("0", "", "off", "false")


def new_graph():
    """A CUDAGraph object whose hipGraph_t stays editable until `seal_graph`."""
    return torch.cuda.CUDAGraph(keep_graph=True) if FIX_MEMSETS else torch.cuda.CUDAGraph()


def seal_graph(graph):
    """After the capture: memset nodes -> kernel nodes, then instantiate.  Returns the number of nodes rewritten."""
    if not FIX_MEMSETS:
        return 0
    from . import hip
    n = hip.load().occd_graph_replace_memsets(graph.raw_cuda_graph())
    if n < 0:
        raise RuntimeError(f"occd_graph_replace_memsets failed ({n})")
    graph.instantiate()
    return n


def make_capturable(opt):
    """Device-side step counters and a device-side learning rate (before the optimizer's first step)."""
    for g in opt.param_groups:
        g["capturable"] = True
        if not torch.is_tensor(g["lr"]) and g["params"] and g["params"][0].is_cuda:
            g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=g["params"][0].device)
    return opt


class _Snapshot:
    """Everything a warm-up step mutates, restored in place (same tensors, same pointers)."""

    def __init__(self, model, opt):
        self.model, self.opt = model, opt
        self.tensors = [(t, t.detach().clone()) for t in list(model.parameters()) + list(model.buffers())]
        self.opt_state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                          for p, st in opt.state.items()}
        self.cur_batch = getattr(model, "cur_batch", None)
        self.metrics = []
        for name in ("train_metrics", "val_metrics", "test_metrics"):
            m = getattr(model, name, None)
            if m is not None and hasattr(m, "hist"):
                self.metrics.append((m, None if m.hist is None else m.hist.clone(), m.count))

    def restore(self):
        with torch.no_grad():
            for t, saved in self.tensors:
                t.copy_(saved)
            for p, st in self.opt.state.items():
                old = self.opt_state.get(id(p))
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if old is not None and torch.is_tensor(old.get(k)):
                            v.copy_(old[k])
                        else:
                            v.zero_()                      # freshly allocated by the warm-up: back to "never stepped"
            for m, hist, count in self.metrics:
                if m.hist is not None:
                    if hist is not None:
                        m.hist.copy_(hist)
                    else:
                        m.hist.zero_()
                m.count = count
        if self.cur_batch is not None:
            self.model.cur_batch = self.cur_batch


@contextlib.contextmanager
def find_syncs():
    """Warn (with a stack) at every host-synchronising op issued inside the block."""
    torch.cuda.set_sync_debug_mode("warn")
    try:
        yield
    finally:
        torch.cuda.set_sync_debug_mode("default")


class GraphedTrainStep:
    def __init__(self, model, opt, batch, bf16=False, buckets=None, warmup=3, batch_idx=0):
        self.model, self.opt, self.batch, self.bf16, self.buckets, self.batch_idx = model, opt, batch, bf16, buckets, batch_idx
        make_capturable(opt)
        self.graph = None
        self.loss = None
        self.warmup = warmup
        self.error = None
        self.memsets_replaced = 0

    def _eager(self):
        if self.buckets is not None:
            self.buckets.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16):
            loss = self.model.training_step(self.batch, self.batch_idx)
        loss.backward()
        if self.buckets is not None:
            self.buckets.finish()
        self.opt.step()
        return loss

    def _sync_decay(self):
        """`sem_step_decay_loss`: the decay factor is a function of the host counter `cur_batch`; the captured step reads
        it from a device scalar that is refreshed (an asynchronous fill, no synchronisation) before every replay."""
        m = self.model
        if getattr(m, "sem_step_decay_loss", False):
            dev = next(m.parameters()).device
            if getattr(m, "_decay_dev", None) is None or m._decay_dev.device != dev:
                m._decay_dev = torch.ones((), dtype=torch.float32, device=dev)
            m._decay_dev.fill_(max(0.1, 1.0 - m.cur_batch / m.total_batch))

    def capture(self):
        dev = next(self.model.parameters()).device
        torch.cuda.synchronize(dev)
        snap = _Snapshot(self.model, self.opt)
        self.model.cur_batch = getattr(self.model, "cur_batch", 0)
        self._sync_decay()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):                    # lazy initialisation, optimizer state, MIOpen solvers
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = new_graph()
        if self.buckets is None:
            self.opt.zero_grad(set_to_none=True)            # gradients are (re)allocated inside the graph's pool
        ok = True
        try:
            with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
                self.loss = self._eager()
            self.memsets_replaced = seal_graph(graph)
        except (RuntimeError, torch.AcceleratorError) as e:
            self.error = repr(e)
            ok = False
        torch.cuda.synchronize(dev)
        snap.restore()                                      # warm-up and capture trained nothing
        torch.cuda.synchronize(dev)
        if ok:
            self.graph = graph
        return ok

    recapture = capture

    def load_batch(self, new_batch):
        """Copy a new batch into the static tensors of the captured step (same keys, shapes and dtypes)."""
        for k, v in new_batch.items():
            dst = self.batch[k]
            if torch.is_tensor(v):
                dst.copy_(v, non_blocking=True)
            elif isinstance(v, (list, tuple)):
                for d, s in zip(dst, v):
                    if torch.is_tensor(s):
                        d.copy_(s, non_blocking=True)

    def __call__(self):
        if self.graph is None:
            return self._eager()
        m = self.model
        m.cur_batch = getattr(m, "cur_batch", 0) + 1        # what training_step does on the host
        self._sync_decay()
        metric = getattr(m, "train_metrics", None)
        if metric is not None and hasattr(metric, "count"):
            metric.count += 1
        self.graph.replay()
        return self.loss
