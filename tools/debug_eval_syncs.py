"""dev: list host-synchronising ops of one eval forward of the benched configuration."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from occdepth_amd import synthetic, train_graph
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev)
with torch.no_grad():
    batch = synthetic.attach_projection(model, synthetic.to_device(synthetic.kitti_frame(seed=0), dev))
    for _ in range(3):
        model(batch)
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with train_graph.find_syncs():
            model(batch)
seen = set()
for x in w:
    key = (x.filename, x.lineno)
    if key not in seen:
        seen.add(key)
        print("SYNC", x.filename.split("/")[-1], x.lineno, str(x.message)[:120], flush=True)
print("eval forward sync scan done:", len(seen), "distinct sites")
