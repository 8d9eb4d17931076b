"""dev: list the host syncs of one eager training step, then try the whole-step hipGraph and time both."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from occdepth_amd import shard, synthetic, train_graph

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, cfg = bench.build_model(dev, train=True)
with torch.no_grad():
    batch = synthetic.attach_projection(model, synthetic.to_device(synthetic.kitti_frame(seed=0), dev))
synthetic.attach_training_targets(model, batch, cfg, seed=1)
opt = train_graph.make_capturable(model.configure_optimizers()[0][0])
gs = train_graph.GraphedTrainStep(model, opt, batch, warmup=2)
gs._eager(); gs._eager()
torch.cuda.synchronize()
print("---- host syncs of one eager step:", flush=True)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    with train_graph.find_syncs():
        gs._eager()
seen = set()
for x in w:
    key = (x.filename, x.lineno, str(x.message)[:80])
    if key not in seen:
        seen.add(key)
        print("SYNC", x.filename.split("/")[-1], x.lineno, str(x.message)[:160], flush=True)
torch.cuda.synchronize()
def timeit(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print(f"eager step: {timeit(gs._eager):.1f} ms", flush=True)
ok = gs.capture()
print("capture ok:", ok, gs.error, flush=True)
if ok:
    l0 = float(gs()); l1 = float(gs()); l2 = float(gs())
    print(f"graphed step: {timeit(gs):.1f} ms; losses {l0:.4f} {l1:.4f} {l2:.4f}", flush=True)
