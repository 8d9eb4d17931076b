"""Per-launch-geometry table of ONE eager config-2 frame (HIP events around every in-repo launch; dev tool, GPU):
    python tools/frame_table.py [n_frames=5] > profiles/r04_frame_per_launch.txt
Rows: kernel kind : geometry tag, launches per frame, ms per frame, algorithmic TF/s and GB/s where the launch reports them.
(ATen / MIOpen launches carry no events: the difference to the stage times printed at the end is theirs + launch gaps.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch


def main(n=5):
    import bench
    from occdepth_amd import hip, synthetic
    dev = torch.device("cuda")
    os.environ.setdefault("OCCDEPTH_GRAPH_ALL", "0")
    os.environ.setdefault("OCCDEPTH_GRAPH_2D", "0")
    model, cfg = bench.build_model(dev)
    model.graph_2d = model.graph_all = False
    with torch.no_grad():
        batch = bench.bench_batch(synthetic.attach_projection(model, synthetic.to_device(synthetic.kitti_frame(seed=0), dev)))
        for _ in range(3):
            model(batch)
        torch.cuda.synchronize()
        with hip.profile() as prof:
            for _ in range(n):
                model(batch)
            torch.cuda.synchronize()
    tot = 0.0
    print(f"# one eager config-2 frame, mean of {n}: kind:geometry, launches / frame, ms / frame, TF/s, GB/s (algorithmic, as reported by the launch)")
    fam = {}
    for k, v in sorted(prof.rows.items(), key=lambda kv: -kv[1]["ms"]):
        ms = v["ms"] / n
        tot += ms
        fam[k.split(":")[0]] = fam.get(k.split(":")[0], 0.0) + ms
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 and v["flops"] > 0 else 0.0
        gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 and v["bytes"] > 0 else 0.0
        print(f"{k:78s} {v['launches'] / n:6.1f} {ms:8.3f} {tf:8.1f} {gb:8.0f}")
    print(f"# sum of in-repo launches: {tot:.3f} ms per frame")
    print("# by kernel kind:", {k: round(v, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
