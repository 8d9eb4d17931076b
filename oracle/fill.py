"""TEST INFRASTRUCTURE ONLY -- platform-independent deterministic parameters.

Golden fixtures store only outputs; weights are regenerated from (key name, seed) with numpy's
frozen MT19937 RandomState so the same state_dict can be rebuilt on any box without committing
hundreds of MB.  BatchNorm statistics / affine are randomised (default-init logits are tiny)."""
import zlib

import numpy as np
import torch


def fill_state_dict(sd, seed=0):
    """In place: every floating tensor of `sd` gets values drawn from a stream keyed by its name."""
    out = {}
    for name in sorted(sd):
        t = sd[name]
        if not torch.is_floating_point(t):
            out[name] = t.clone() if name.endswith("num_batches_tracked") else t
            continue
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 32))
        shape = tuple(t.shape)
        if name.endswith("running_var"):
            v = rs.uniform(0.5, 1.5, shape)
        elif name.endswith("running_mean"):
            v = rs.normal(0, 0.1, shape)
        elif name.endswith(".bias"):
            v = rs.normal(0, 0.1, shape)
        elif t.dim() == 1 and name.endswith(".weight"):      # norm scale
            v = rs.uniform(0.5, 1.5, shape)
        elif t.dim() >= 2 and name.endswith(".weight"):      # conv / linear: He-style fan-in scaling
            fan_in = int(np.prod(shape[1:]))
            if "main.0.weight" in name and t.dim() == 5 and ("up_" in name):   # ConvTranspose3d (Cin, Cout, ...)
                fan_in = shape[0] * int(np.prod(shape[2:])) // 8 + 1
            v = rs.normal(0, np.sqrt(2.0 / max(fan_in, 1)), shape)
        else:                                                 # buffers such as voxel_size: keep
            out[name] = t
            continue
        out[name] = torch.from_numpy(v.astype(np.float32)).reshape(shape)
    return out


# ------------------------------------------------------------------------------------------------
# Conditioning.  Random weights + random BN statistics let activations grow by orders of magnitude
# through ~200 layers (logits ~1e18 at config 2), which turns fp32 round-off of any backend into
# O(1e-3) differences after a softmax / sigmoid.  The golden generator therefore calibrates the
# BatchNorm running statistics ONCE on the case's own synthetic input (data-dependent init) and
# stores them in the fixture; tests overlay them on the seeded weights, so every box rebuilds the
# same, well-conditioned state_dict.
def calibrate_bn(model, run_forward):
    """Set every executed BatchNorm's running stats to the batch statistics of one forward pass."""
    import torch.nn as nn
    bns = [m for m in model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
    saved = [(m.momentum, m.training) for m in bns]
    for m in bns:
        m.momentum = 1.0
        m.training = True
    try:
        with torch.no_grad():
            run_forward(model)
    finally:
        for m, (mom, tr) in zip(bns, saved):
            m.momentum = mom
            m.training = tr
    return model


def bn_stats(state_dict):
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in state_dict.items()
            if k.endswith("running_mean") or k.endswith("running_var")}


def overlay(sd, arrays, tag):
    """sd updated with the calibrated statistics stored as '<tag>::<key>' in a golden npz."""
    pre = tag + "::"
    out = dict(sd)
    n = 0
    for name in arrays.files:
        if name.startswith(pre):
            key = name[len(pre):]
            assert key in out and tuple(out[key].shape) == arrays[name].shape, key
            out[key] = torch.from_numpy(arrays[name].copy())
            n += 1
    assert n > 0, f"no calibrated statistics stored under {tag}"
    return out
