"""Probe (GPU box): ATen multi-block reductions inside plain (un-rewritten) and sealed hipGraphs, over many allocation
layouts: how often does a replay return a wrong column sum?  (train_graph.new_graph / seal_graph, csrc/graph_fix.hip)"""
import json
import random
import sys

import torch

from occdepth_amd import train_graph


def trial(sealed, layouts=150, replays=3, seed=0):
    rng = random.Random(seed)
    wrong, cases, rewritten = 0, [], 0
    for k in range(layouts):
        rows = rng.choice([65536, 262144, 524288, 2097152])
        cs, c = rng.choice([(8, 2), (24, 20), (32, 32), (80, 80), (8, 8)])
        inp = torch.randn(rows, cs, device="cuda")
        npad = rng.randrange(0, 12)
        pads_pre = [torch.empty(rng.randrange(1, 5000), device="cuda") for _ in range(rng.randrange(0, 6))]
        inp[:, :c].sum(0, dtype=torch.float32)
        torch.cuda.synchronize()
        g = train_graph.new_graph() if sealed else torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            alive = [torch.empty(rng.randrange(1, 3000), device="cuda") for _ in range(npad)]
            for a in alive[:3]:
                a.fill_(float("nan"))
            tmp = inp * 2.0
            out = inp[:, :c].sum(0, dtype=torch.float32)
            out2 = tmp[:, :c].sum(0, dtype=torch.float32)
        if sealed:
            rewritten += train_graph.seal_graph(g)
        for r in range(replays):
            inp.normal_()
            g.replay()
            want = inp[:, :c].double().sum(0)
            e1 = float((out.double() - want).abs().max() / (want.abs().max() + 1.0))
            e2 = float((out2.double() - 2 * want).abs().max() / (2 * want.abs().max() + 1.0))
            if not (e1 < 1e-3 and e2 < 1e-3):
                wrong += 1
                if len(cases) < 6:
                    cases.append({"layout": k, "rows": rows, "cs": cs, "c": c, "replay": r, "err": [e1, e2]})
        del g, alive, pads_pre, out, out2, tmp, inp
    return {"sealed": sealed, "layouts": layouts, "wrong_replays": wrong, "memsets_rewritten": rewritten, "first": cases}


if __name__ == "__main__":
    res = [trial(False), trial(True)]
    print(json.dumps(res))
    sys.exit(0)
