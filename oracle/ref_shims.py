"""TEST INFRASTRUCTURE ONLY -- import the real reference (/root/reference) in this container.

The reference needs pytorch_lightning, mmdet, kornia and torch.hub/geffnet, none of which is
installed.  `install()` registers three tiny stand-in modules in sys.modules (semantics per
SURVEY.md section 8c) and patches `UNet2D.build` so the reference's own model code runs on CPU.
Only tests/golden/make_golden.py uses this, and only where /root/reference exists; nothing on the
GPU box or in the product path may import it.

Stand-ins (third-party code that is NOT in /root/reference, restated from their published
behaviour; parity of these three is unpinned by the reference):
  pytorch_lightning==1.4.9  LightningModule -> nn.Module + no-op save_hyperparameters/log
  mmdet==2.20.0             models.backbones.resnet.BasicBlock (conv1,bn1,conv2,bn2, identity skip)
  kornia==0.5.0             utils.create_meshgrid3d, convert_points_{to,from}_homogeneous, transform_points
"""
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "occdepth"))


def _pl():
    m = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    class LightningDataModule:
        pass

    m.LightningModule = LightningModule
    m.LightningDataModule = LightningDataModule
    return m


def _mmdet():
    class BasicBlock(nn.Module):
        def __init__(self, inplanes, planes):
            super().__init__()
            self.conv1 = nn.Conv2d(inplanes, planes, 3, padding=1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)

        def forward(self, x):
            return F.relu(self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x))))) + x)

    mods = {}
    for name in ("mmdet", "mmdet.models", "mmdet.models.backbones", "mmdet.models.backbones.resnet"):
        mods[name] = types.ModuleType(name)
    mods["mmdet"].models = mods["mmdet.models"]
    mods["mmdet.models"].backbones = mods["mmdet.models.backbones"]
    mods["mmdet.models.backbones"].resnet = mods["mmdet.models.backbones.resnet"]
    mods["mmdet.models.backbones.resnet"].BasicBlock = BasicBlock
    return mods


def _kornia():
    k = types.ModuleType("kornia")
    ku = types.ModuleType("kornia.utils")

    def create_meshgrid3d(depth, height, width, normalized_coordinates=True, device=None, dtype=None):
        xs = torch.linspace(0, width - 1, int(width), device=device, dtype=dtype)
        ys = torch.linspace(0, height - 1, int(height), device=device, dtype=dtype)
        zs = torch.linspace(0, depth - 1, int(depth), device=device, dtype=dtype)
        assert not normalized_coordinates
        base = torch.stack(torch.meshgrid([zs, xs, ys], indexing="ij")).transpose(1, 2)
        return base.unsqueeze(0).permute(0, 3, 4, 2, 1)

    def convert_points_to_homogeneous(points):
        return F.pad(points, [0, 1], "constant", 1.0)

    def convert_points_from_homogeneous(points, eps=1e-8):
        z = points[..., -1:]
        scale = torch.where(torch.abs(z) > eps, 1.0 / (z + eps), torch.ones_like(z))
        return scale * points[..., :-1]

    def transform_points(trans_01, points_1):
        shape_inp = list(points_1.shape)
        points_1 = points_1.reshape(-1, points_1.shape[-2], points_1.shape[-1])
        trans_01 = trans_01.reshape(-1, trans_01.shape[-2], trans_01.shape[-1])
        trans_01 = torch.repeat_interleave(trans_01, repeats=points_1.shape[0] // trans_01.shape[0], dim=0)
        points_1_h = convert_points_to_homogeneous(points_1)
        points_0_h = torch.bmm(points_1_h, trans_01.permute(0, 2, 1))
        points_0 = convert_points_from_homogeneous(points_0_h)
        shape_inp[-2] = points_0.shape[-2]
        shape_inp[-1] = points_0.shape[-1]
        return points_0.reshape(shape_inp)

    ku.create_meshgrid3d = create_meshgrid3d
    k.utils = ku
    k.convert_points_to_homogeneous = convert_points_to_homogeneous
    k.convert_points_from_homogeneous = convert_points_from_homogeneous
    k.transform_points = transform_points
    return {"kornia": k, "kornia.utils": ku}


_installed = False


def install():
    """Make `import occdepth...` resolve to the reference with the stand-ins active."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("/root/reference is not present: the real reference can only run in the build container")
    sys.modules.setdefault("pytorch_lightning", _pl())
    for name, mod in {**_mmdet(), **_kornia()}.items():
        sys.modules.setdefault(name, mod)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _installed = True


def patch_unet2d_build(make_backend):
    """Replace the torch.hub download in reference UNet2D.build (unet2d.py:232-255) by
    `make_backend(name)`; everything else in build() is reproduced (global_pool/classifier -> Identity)."""
    install()
    from occdepth.models import unet2d as ref_unet2d

    def build(cls, **kwargs):
        name = kwargs["backbone_2d_name"]
        backend = make_backend(name)
        backend.global_pool = nn.Identity()
        backend.classifier = nn.Identity()
        return cls(backend, num_features=ref_unet2d.NUM_FEATURES[name], **kwargs)

    ref_unet2d.UNet2D.build = classmethod(build)
    return ref_unet2d


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def load_config(rel_path, **overrides):
    """yaml under /root/reference/occdepth/config -> attribute dict (what hydra hands to OccDepth)."""
    import yaml
    with open(os.path.join(REF_ROOT, "occdepth", "config", rel_path)) as f:
        cfg = AttrDict(yaml.safe_load(f))
    cfg.update(overrides)
    return cfg
