"""Where the ATen layout copies / cats / sums / adds of one config-2 training step come from (dev tool; GPU).  torch.profiler's
Python stacks are empty on this stack, so the Python-level entry points are wrapped instead: every call on a CUDA tensor of
>= 2^17 elements is attributed to the innermost frame inside this repository (models / autograd wrappers), with bytes moved.
    python tools/trace_train_copies.py [bf16]
Calls made from C++ (the autograd engine's gradient accumulation, ATen-internal .contiguous()) do not show up here; the
profiler's operator table (tools/prof_train_aten.py) minus this table is their share."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

LOG = collections.defaultdict(lambda: [0, 0])
ACTIVE = [False]
REAL = os.path.realpath(ROOT)


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        f = os.path.realpath(fr.filename)
        if f.startswith(REAL) and "/tools/" not in f and not f.endswith("bench.py"):
            return f"{os.path.relpath(f, REAL)}:{fr.lineno} {fr.name}"
    return "?"


def note(op, t, moved):
    if ACTIVE[0] and torch.is_tensor(t) and t.is_cuda and t.numel() >= (1 << 17):
        k = (op, tuple(t.shape), str(t.dtype).replace("torch.", ""), site())
        LOG[k][0] += 1
        LOG[k][1] += moved


def wrap_method(name, moved_fn, cond=None):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if cond is None or cond(self, out, a, k):
            note("Tensor." + name, self, moved_fn(self, out))
        return out
    setattr(torch.Tensor, name, f)


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
    nb = lambda t: t.numel() * t.element_size()
    wrap_method("contiguous", lambda s, o: 2 * nb(s), lambda s, o, a, k: o.data_ptr() != s.data_ptr())
    wrap_method("copy_", lambda s, o: 2 * nb(s))
    wrap_method("clone", lambda s, o: 2 * nb(s))
    wrap_method("float", lambda s, o: nb(s) + nb(o), lambda s, o, a, k: o.data_ptr() != s.data_ptr())
    wrap_method("to", lambda s, o: nb(s) + nb(o), lambda s, o, a, k: torch.is_tensor(o) and o.data_ptr() != s.data_ptr())
    wrap_method("sum", lambda s, o: nb(s))
    wrap_method("zero_", lambda s, o: nb(s))
    wrap_method("fill_", lambda s, o: nb(s))
    for name in ("add", "add_", "__add__", "__iadd__", "__radd__", "mul", "__mul__", "mul_"):
        wrap_method(name, lambda s, o: 3 * nb(s))
    for fn_name in ("cat", "stack", "zeros", "zeros_like", "ones_like"):
        orig = getattr(torch, fn_name)

        def g(*a, _orig=orig, _n=fn_name, **k):
            out = _orig(*a, **k)
            note("torch." + _n, out, (2 if _n in ("cat", "stack") else 1) * nb(out))
            return out
        setattr(torch, fn_name, g)
    ACTIVE[0] = True
    step()
    torch.cuda.synchronize()
    ACTIVE[0] = False
    tot = sum(v[1] for v in LOG.values())
    print(f"# Python-level copy / cat / sum / add / fill calls on CUDA tensors >= 128 Ki elements in one {'bf16-MFMA' if bf16 else 'fp32'} "
          f"config-2 training step: {sum(v[0] for v in LOG.values())} calls, {tot / 1e9:.2f} GB moved (at ~3 TB/s: {tot / 3e9:.2f} ms)")
    for (op, shape, dt, where), (n, moved) in sorted(LOG.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{moved / 1e6:9.1f} MB  x{n:<3d} {op:18s} {str(shape):34s} {dt:9s} {where}")


if __name__ == "__main__":
    main("bf16" in sys.argv[1:])
