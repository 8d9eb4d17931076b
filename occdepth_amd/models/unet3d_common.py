"""Shared forward engine of the two 3-D UNets (SemanticKITTI / NYUv2).

Both networks are: two encoder stages (3 dilated bottlenecks + strided bottleneck each), the context-relation prior at
the coarsest level, two transposed-convolution decoders with skip additions, an optional last upsampling to the full
resolution and the segmentation head(s).  The dataset classes only build the sub-modules (under the attribute names the
reference's checkpoints use) and name them in `LAYOUT`; this class walks that layout in one of two modes:

  * `_VoxMode`  -- the forward-only HIP path: every activation is a channels-last `Vox`, skip additions are fused into the
    decoder's epilogue, outputs are exposed as (B, C, X, Y, Z) views;
  * `_AtenMode` -- the differentiable path (nn.Module calls; 3-D convolutions through autograd3d).
"""
import torch.nn as nn

from ..fused import as_vox, needs_autograd


class _VoxMode:
    prepare = staticmethod(as_vox)

    @staticmethod
    def stage(seq, x):
        for m in seq:
            x = m.forward_vox(x)
        return x

    @staticmethod
    def call(mod, x):
        return mod.forward_vox(x)

    @staticmethod
    def decode(mod, x, skip):
        return mod.forward_vox(x, skip=skip)

    @staticmethod
    def prior(mod, x, res):
        ret = mod.forward_vox(x)
        res["P_logits"] = ret["P_logits"]
        res["x"] = ret["x"].ncdhw()
        return ret["x"]

    @staticmethod
    def expose(t):
        return t.ncdhw()


class _AtenMode:
    prepare = staticmethod(lambda x: x)

    @staticmethod
    def stage(seq, x):
        return seq(x)

    @staticmethod
    def call(mod, x):
        return mod(x)

    @staticmethod
    def decode(mod, x, skip):
        if hasattr(mod, "forward_train"):
            return mod.forward_train(x, skip)          # BatchNorm + ReLU + skip addition in one fused pass
        return mod(x) + skip

    @staticmethod
    def prior(mod, x, res):
        ret = mod(x)
        res.update(ret)
        return ret["x"]

    @staticmethod
    def expose(t):
        return t


class UNet3DBase(nn.Module):
    # attribute names: (encoder stage 1, encoder stage 2, decoder coarse->mid, decoder mid->fine, final upsampling or None,
    #                   main head, occluded head or None)
    LAYOUT = None

    def _run(self, mode, fine):
        enc1, enc2, dec_mid, dec_fine, to_full, head, occluded_head = self.LAYOUT
        res = {}
        mid = mode.stage(getattr(self, enc1), fine)
        coarse = mode.stage(getattr(self, enc2), mid)
        if self.context_prior:
            coarse = mode.prior(self.CP_mega_voxels, coarse, res)
        up_mid = mode.decode(getattr(self, dec_mid), coarse, mid)
        up_fine = mode.decode(getattr(self, dec_fine), up_mid, fine)
        top = mode.call(getattr(self, to_full), up_fine) if to_full else up_fine
        if not self.infer_mode:
            res["x3d_l1"], res["x3d_l2"], res["x3d_l3"] = (mode.expose(t) for t in (up_fine, up_mid, coarse))
        out = mode.call(getattr(self, head), top)
        if self.cascade_cls:
            res["ssc_logit"] = mode.expose(out[0])
            if not self.infer_mode:
                res["occ_logit"] = mode.expose(out[1])
        else:
            res["ssc_logit"] = mode.expose(out)
        if occluded_head and getattr(self, "occluded_cls", False):
            occluded = mode.call(getattr(self, occluded_head), top)
            if not self.infer_mode:
                res["occluded_logit"] = mode.expose(occluded)
        return res

    def forward(self, input_dict):
        mode = _AtenMode if needs_autograd(self) else _VoxMode
        return self._run(mode, mode.prepare(input_dict["x3d"]))
