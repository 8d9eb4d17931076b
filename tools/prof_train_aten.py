"""Which ATen / library work is left in the config-2 training step, and WHERE it comes from (dev tool; GPU):

    python tools/prof_train_aten.py [bf16]

One eager bf16-MFMA (or fp32) step of `bench.py --train`'s model under torch.profiler with shapes and Python stacks; prints
the device-time table by operator and, for the operators that are not this repo's kernels, the input shapes and the
innermost repo source line that issued them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from torch.profiler import ProfilerActivity, profile


def main(bf16):
    import bench
    from occdepth_amd import autograd3d, synthetic
    autograd3d.set_bf16_mfma(bf16)
    dev = torch.device("cuda")
    model, cfg = bench.build_model(dev, train=True)
    with torch.no_grad():
        batch = synthetic.attach_projection(model, synthetic.to_device(synthetic.kitti_frame(seed=0), dev))
    synthetic.attach_training_targets(model, batch, cfg, seed=1)
    opt = model.configure_optimizers()[0][0]

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch, 0)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = []
    for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if t <= 0 or not ev.key.startswith("aten::"):
            continue
        where = "?"
        for fr in (ev.stack or []):
            if repo in fr and "/tools/" not in fr and "bench.py" not in fr:
                where = fr.replace(repo + "/", "").strip()
                break
        rows.append((t, ev.count, ev.key, str(ev.input_shapes)[:80], where[:120]))
    print("\n== ATen operators with device time, by (op, shapes, innermost repo frame), top 70 ==")
    for t, n, name, shapes, where in sorted(rows, reverse=True)[:70]:
        print(f"{t / 1e3:8.3f} ms  x{n:<4d} {name:30s} {shapes:82s} {where}")


if __name__ == "__main__":
    main("bf16" in sys.argv[1:])
