"""Training-step baseline for SURVEY 8(f) row N1 (dev tool; needs the GPU):  forward (ATen / MIOpen autograd graph)
+ the fused loss statistics kernels + backward + AdamW on one synthetic config-2 stereo frame.

    python tools/bench_train.py [steps=3] [config=kitti_a100|kitti_2080ti] [bf16]
Prints ms per phase.  This is the number later rounds' hand-written backward kernels have to beat."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occdepth_amd import configs, hip, synthetic
from occdepth_amd.loss.sscMetrics import SSCMetrics
from occdepth_amd.models.OccDepth import OccDepth


def main(steps=3, cfg_name="kitti_a100", amp=None):
    torch.manual_seed(0)
    dev = torch.device("cuda")
    cfg = getattr(configs, cfg_name).clone()
    C = cfg.n_classes
    model = OccDepth(class_names=[str(i) for i in range(C)], class_weights=torch.ones(C), class_weights_occ=torch.ones(2),
                     full_scene_size=tuple(cfg.full_scene_size), project_res=configs.PROJECT_RES, config=cfg).to(dev)
    model.train()
    batch = synthetic.to_device(synthetic.kitti_frame(seed=0), dev)
    with torch.no_grad():
        synthetic.attach_projection(model, batch)
    dims = tuple(cfg.full_scene_size)
    g = torch.Generator(device="cuda").manual_seed(1)
    target = torch.randint(0, C, (1, *dims), device=dev, generator=g).to(torch.uint8)
    target[torch.rand(1, *dims, device=dev, generator=g) < 0.2] = 255
    nf = cfg.frustum_size ** 2
    fid = (torch.arange(dims[0], device=dev).view(-1, 1, 1) * 8 // dims[0]) * 8 + \
        (torch.arange(dims[1], device=dev).view(1, -1, 1) * 8 // dims[1]) + torch.zeros(1, 1, dims[2], device=dev, dtype=torch.long)
    masks = torch.stack([fid == f for f in range(nf)], 0)
    batch.update(target=target, frustums_masks=[masks], frustums_class_dists=[torch.rand(nf, C, device=dev)],
                 gt_depth=torch.rand(1, 1, 370, 1220, device=dev) * 50.0)
    opt = model.configure_optimizers()[0][0]               # AdamW (fused on the GPU) + the reference's MultiStepLR
    metric = SSCMetrics(C)
    times = []
    for it in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.time()
        if "CP_mega_matrices" not in batch:                # shaped after the model's relation logits (first step only)
            with torch.no_grad():
                model.eval()
                shp = model(batch)["P_logits"].shape if cfg.context_prior else None
                model.train()
            if shp is not None:
                batch["CP_mega_matrices"] = [(torch.rand(shp[1], shp[3], shp[2], device=dev) < 0.3).float()]
            torch.cuda.synchronize()
            t0 = time.time()
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp == "bf16"):
            loss = model.step(batch, "train", metric)
        torch.cuda.synchronize()
        t1 = time.time()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.time()
        opt.step()
        torch.cuda.synchronize()
        t3 = time.time()
        times.append((t1 - t0, t2 - t1, t3 - t2))
        print(f"step {it}: loss {float(loss):.4f}  fwd+loss {1e3 * (t1 - t0):.1f} ms  bwd {1e3 * (t2 - t1):.1f} ms  "
              f"adamw {1e3 * (t3 - t2):.1f} ms  mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    best = min(times[1:], key=sum)
    print(f"{cfg_name} train step ({'bf16 autocast (3-D convolutions + losses fp32)' if amp == 'bf16' else 'fp32'}, batch 1): {1e3 * sum(best):.1f} ms -> {1 / sum(best):.2f} steps/s "
          f"(fwd+loss {1e3 * best[0]:.1f}, bwd {1e3 * best[1]:.1f}, opt {1e3 * best[2]:.1f})")
    with hip.profile() as prof:
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp == "bf16"):
            loss = model.step(batch, "train", metric)
        loss.backward()
    for k, v in prof.rows.items():
        print(f"  {k:40s} n={v['launches']:3d} {v['ms']:7.3f} ms")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3, sys.argv[2] if len(sys.argv) > 2 else "kitti_a100",
         sys.argv[3] if len(sys.argv) > 3 else None)
