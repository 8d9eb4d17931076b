"""Probe (GPU box): WHAT does the captured training step lose when its hipMemsetAsync nodes fill with a wrong pattern?

Run with OCCD_DBG_MEMSET_XOR=4 (csrc/graph_fix.hip perturbs every rewritten fill value -- the runtime's stale pattern made
deterministic): the reduced SemanticKITTI model's step is captured, replayed, and every parameter gradient is compared with
the eager step's from the same state.  The gradients that differ are the ones whose producer relies on a memset node --
ATen's multi-block reductions (their semaphores): the convolutions' bias gradients, `occ_classes.bias` among them.  Without
the variable only biases in front of a BatchNorm are listed (their true gradient is zero: round-off against round-off).
Prints one JSON line.
"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from occdepth_amd import train_graph
from test_train_step import _small_train_setup

torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
m0, batch = _small_train_setup("kitti_small", "cuda")
grads = {}
for mode in ("eager", "graph"):
    m = copy.deepcopy(m0).train()
    m.cur_batch = 0
    opt = train_graph.make_capturable(torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True))
    gs = train_graph.GraphedTrainStep(m, opt, batch, warmup=2)
    if mode == "graph":
        assert gs.capture(), gs.error
    loss = float(gs())
    torch.cuda.synchronize()
    grads[mode] = ({k: p.grad.detach().double().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}, loss,
                   gs.memsets_replaced)
(ge, le, _), (gg, lg, n) = grads["eager"], grads["graph"]
off = {}
for k in ge:
    a, b = ge[k], gg[k]
    if not bool(torch.isfinite(b).all()):
        off[k] = "non-finite"
    else:
        rel = float((a - b).norm() / (a.norm() + 1e-30))
        if rel > 5e-2:
            off[k] = round(rel, 4)
print(json.dumps({"xor": os.environ.get("OCCD_DBG_MEMSET_XOR"), "memset_nodes_rewritten": n, "loss_eager": le, "loss_graph": lg,
                  "parameters": len(ge), "gradients_off": off}))
