"""2-D UNet: EfficientNet encoder + AdaBins-style BN decoder (mirror of occdepth/models/unet2d.py).

Convolutions run in PyTorch-ROCm / MIOpen (north_star) except the 3x3 convolutions of the three low-resolution decoder
levels, which take the Winograd-domain MFMA path (csrc/wino2d.hip + batched GEMMs); BatchNorm + activation, bilinear
upsample + concat and the depthwise convolutions are fused HIP passes in eval mode.  Faithful quirks: the decoder taps
encoder features [4, 5, 6, 8, 11] (conv_head output BEFORE bn2), `conv2` is a 1x1 conv with padding=1 (grows the
1/32 map by 2), and `up1` concatenates the raw image.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from ..fused import _stamp, bn_affine_cached, needs_autograd, wino_fused_operands
from .efficientnet import EfficientNet, pw_operands, pw_wins

MODEL_NAME = "tf_efficientnet_b3_ns"
MODEL_CHANNELS = {
    "tf_efficientnet_b3_ns": [3, 24, 32, 48, 136],
    "tf_efficientnet_b4_ns": [3, 24, 32, 56, 160],
    "tf_efficientnet_b5_ns": [3, 32, 40, 64, 176],
    "tf_efficientnet_b7_ns": [3, 32, 48, 80, 224],
}
NUM_FEATURES = {
    "tf_efficientnet_b3_ns": 1536,
    "tf_efficientnet_b4_ns": 1792,
    "tf_efficientnet_b5_ns": 2048,
    "tf_efficientnet_b7_ns": 2560,
}
_SCALES = (16, 8, 4, 2, 1)


class UpSampleBN(nn.Module):
    def __init__(self, skip_input, output_features):
        super().__init__()
        layers = []
        for cin in (skip_input, output_features):
            layers += [nn.Conv2d(cin, output_features, kernel_size=3, stride=1, padding=1),
                       nn.BatchNorm2d(output_features), nn.LeakyReLU()]
        self._net = nn.Sequential(*layers)

    # Winograd-domain GEMMs on the MFMA pipe instead of MIOpen's VALU Winograd: pays when tiles are few and channels
    # many (the 1/16, 1/8, 1/4 levels of config 2; tools/bench_wino.py has the per-level numbers)
    WINOGRAD = os.environ.get("OCCDEPTH_WINOGRAD", "1") == "1"
    WINOGRAD_MAX_PIXELS = int(os.environ.get("OCCDEPTH_WINOGRAD_MAX_PIXELS", "80000"))      # B * H * W
    WINOGRAD_MIN_CIN = int(os.environ.get("OCCDEPTH_WINOGRAD_MIN_CIN", "256"))

    def _wino_operands(self, conv, bn):
        key = _stamp(conv, bn)
        cache = self.__dict__.setdefault("_wino_cache", {})
        hit = cache.get(id(conv))
        if hit is None or hit[0] != key:
            scale, shift = bn_affine_cached(bn)
            if conv.bias is not None:
                shift = shift + scale * conv.bias.detach().float()
            hit = (key, hip.winograd_weights(conv.weight), scale.contiguous(), shift.contiguous())
            cache[id(conv)] = hit
        return hit[1:]

    # experimental: the high-resolution levels in strips of tile rows sized so that V + M of one strip stay cache-resident
    WINOGRAD_HIRES_PIXELS = int(os.environ.get("OCCDEPTH_WINOGRAD_HIRES_PIXELS", "0"))
    WINOGRAD_HIRES_MIN_CIN = int(os.environ.get("OCCDEPTH_WINOGRAD_HIRES_MIN_CIN", "64"))
    WINOGRAD_STRIP_MB = float(os.environ.get("OCCDEPTH_WINOGRAD_STRIP_MB", "96"))

    # K10, the fused Winograd kernel (V in registers, M in the accumulators): levels with at least this many pixels
    # (B * H * W); below, the unfused transforms + batched GEMMs above keep their large-K efficiency
    FUSED_MIN_PIXELS = int(os.environ.get("OCCDEPTH_WINO_FUSED_MIN_PIXELS", "50000"))
    FUSED = os.environ.get("OCCDEPTH_WINO_FUSED", "1") == "1"

    def _fused_operands(self, conv, bn):
        return wino_fused_operands(self, conv, bn)

    def _conv_bn_act(self, f, conv, bn, act):
        B, C, H, W = f.shape
        if self.FUSED and B * H * W >= self.FUSED_MIN_PIXELS:
            upk, shift = self._fused_operands(conv, bn)
            return hip.conv2d_3x3_fused(f, upk, conv.out_channels, shift, "leaky", act.negative_slope)
        if self.WINOGRAD and C >= self.WINOGRAD_MIN_CIN and B * H * W <= self.WINOGRAD_MAX_PIXELS:
            U, scale, shift = self._wino_operands(conv, bn)
            return hip.conv2d_3x3_winograd(f, U, scale, shift, "leaky", act.negative_slope)
        if self.WINOGRAD and C >= self.WINOGRAD_HIRES_MIN_CIN and B * H * W <= self.WINOGRAD_HIRES_PIXELS:
            U, scale, shift = self._wino_operands(conv, bn)
            per_row = 16.0 * B * ((W + 1) // 2) * (C + U.shape[2]) * 4
            rows = max(1, int(self.WINOGRAD_STRIP_MB * 2 ** 20 / per_row))
            return hip.conv2d_3x3_winograd(f, U, scale, shift, "leaky", act.negative_slope, strip_rows=rows)
        return hip.affine_act(conv(f), *bn_affine_cached(bn), "leaky", slope=act.negative_slope)

    def forward(self, x, concat_with):
        if x.is_cuda and not needs_autograd(self) and x.dtype == torch.float32:
            # eval: bilinear-up + concat in one HIP pass; BatchNorm + LeakyReLU fused into the Winograd output transform
            # or applied in one pass behind the MIOpen convolution
            f = hip.upsample_bilinear_cat(x, concat_with)
            n = self._net
            f = self._conv_bn_act(f, n[0], n[1], n[2])
            return self._conv_bn_act(f, n[3], n[4], n[5])
        up = F.interpolate(x, size=concat_with.shape[2:], mode="bilinear", align_corners=True)
        return self._net(torch.cat([up, concat_with], dim=1))


class DecoderBN(nn.Module):
    def __init__(self, num_features, bottleneck_features, out_feature, use_decoder=True, backbone_2d_name=None,
                 return_up_feats=None):
        super().__init__()
        features = int(num_features)
        self.use_decoder = use_decoder
        self.backbone_2d_name = backbone_2d_name
        self.return_up_feats = return_up_feats
        self.conv2 = nn.Conv2d(bottleneck_features, features, kernel_size=1, stride=1, padding=1)
        for s in _SCALES:
            setattr(self, f"out_feature_1_{s}", out_feature)
            setattr(self, f"feature_1_{s}", features // (32 // s))
        if not use_decoder:
            self.resize_output_1_1 = nn.Conv2d(3, out_feature, kernel_size=1)
            self.resize_output_1_2 = nn.Conv2d(32, out_feature * 2, kernel_size=1)
            self.resize_output_1_4 = nn.Conv2d(48, out_feature * 4, kernel_size=1)
            return
        skips = MODEL_CHANNELS[backbone_2d_name]
        prev = features
        for s, skip in zip(_SCALES, reversed(skips)):
            if return_up_feats > s:
                continue
            width = getattr(self, f"feature_1_{s}")
            setattr(self, f"resize_output_1_{s}", nn.Conv2d(width, out_feature, kernel_size=1))
            setattr(self, f"up{s}", UpSampleBN(skip_input=prev + skip, output_features=width))
            prev = width

    def forward(self, features):
        taps = {16: features[8], 8: features[6], 4: features[5], 2: features[4], 1: features[0]}
        f = features[11]
        if (f.is_cuda and not needs_autograd(self) and f.dtype == torch.float32 and pw_wins(f)
                and self.conv2.kernel_size == (1, 1) and self.conv2.padding == (1, 1)):
            # the reference's 1x1 convolution with padding=1 (unet2d.py:65-67): bias on the one-pixel frame, the GEMM
            # (K11s: 2560 -> 2560 on < 1000 pixels) inside it
            wpk, shift = pw_operands(self, self.conv2)
            B, _, H, W = f.shape
            x = shift.view(1, -1, 1, 1).expand(B, self.conv2.out_channels, H + 2, W + 2).contiguous()
            x[:, :, 1:-1, 1:-1] = hip.conv1x1(f, wpk, self.conv2.out_channels, shift)
        else:
            x = self.conv2(f)
        if not self.use_decoder:
            bs = features[4].shape[0]
            return {"1_1": self.resize_output_1_1(features[0]), "1_2": self.resize_output_1_2(features[4]),
                    "1_4": self.resize_output_1_4(features[5]),
                    "global": features[-1].reshape(bs, 2560, -1).mean(2)}
        res = {}
        for s in _SCALES:
            if self.return_up_feats > s:
                continue
            x = getattr(self, f"up{s}")(x, taps[s])
            head = getattr(self, f"resize_output_1_{s}")
            if x.is_cuda and not needs_autograd(self) and x.dtype == torch.float32 and pw_wins(x):
                # 1x1 convolution + bias on the MFMA GEMM (K11), written pixel-major: the 2D->3D lift gathers pixel rows,
                # so the (B, C, H, W) result is returned as a channels-last view and no transpose pass exists
                wpk, shift = pw_operands(self, head)
                res[f"1_{s}"] = hip.conv1x1(x, wpk, head.out_channels, shift, nhwc=True)
            else:
                res[f"1_{s}"] = head(x)
        return res


class Encoder(nn.Module):
    def __init__(self, backend):
        super().__init__()
        self.original_model = backend

    def forward(self, x):
        features = [x]
        for name, mod in self.original_model._modules.items():
            if name == "blocks":
                for stage in mod._modules.values():
                    features.append(stage(features[-1]))
            else:
                features.append(mod(features[-1]))
        return features


class UNet2D(nn.Module):
    def __init__(self, backend, num_features, out_feature, use_decoder=True, backbone_2d_name=None,
                 return_up_feats=1):
        super().__init__()
        self.use_decoder = use_decoder
        self.encoder = Encoder(backend)
        self.decoder = DecoderBN(out_feature=out_feature, use_decoder=use_decoder, bottleneck_features=num_features,
                                 num_features=num_features, backbone_2d_name=backbone_2d_name,
                                 return_up_feats=return_up_feats)

    def forward(self, x, **kwargs):
        return self.decoder(self.encoder(x), **kwargs)

    def get_encoder_params(self):
        return self.encoder.parameters()

    def get_decoder_params(self):
        return self.decoder.parameters()

    @classmethod
    def build(cls, **kwargs):
        """Same contract as the reference (unet2d.py:232-255).  The encoder is the in-repo
        EfficientNet restatement (random init); hub weights, when a local file is named by
        OCCDEPTH_BACKBONE_WEIGHTS, are loaded by key."""
        name = kwargs["backbone_2d_name"]
        print("Loading base model {}...".format(name), end="")
        basemodel = EfficientNet(name)
        weights = os.environ.get("OCCDEPTH_BACKBONE_WEIGHTS")
        if weights:
            basemodel.load_state_dict(torch.load(weights, map_location="cpu"), strict=True)
        print("Done.")
        basemodel.global_pool = nn.Identity()
        basemodel.classifier = nn.Identity()
        m = cls(basemodel, num_features=NUM_FEATURES[name], **kwargs)
        print("INFO: return_up_feats set to : {}.".format(kwargs["return_up_feats"]))
        return m
