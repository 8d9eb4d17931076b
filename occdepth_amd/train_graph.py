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
  * the learning rate is baked into the captured update: re-capture (`GraphedTrainStep.recapture`) when a scheduler
    changes it (MultiStepLR: twice in a run).
Reference: scripts/train.py:176-206 drives the same step through PyTorch-Lightning, eagerly.
"""
import contextlib

import torch


def make_capturable(opt):
    for g in opt.param_groups:
        g["capturable"] = True
    return opt


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

    def capture(self):
        dev = next(self.model.parameters()).device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):                    # lazy initialisation, optimizer state, MIOpen solvers
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        if self.buckets is None:
            self.opt.zero_grad(set_to_none=True)            # gradients are (re)allocated inside the graph's pool
        try:
            with torch.cuda.graph(graph):
                self.loss = self._eager()
        except (RuntimeError, torch.AcceleratorError) as e:
            self.error = repr(e)
            torch.cuda.synchronize(dev)
            return False
        self.graph = graph
        return True

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
        self.graph.replay()
        return self.loss
