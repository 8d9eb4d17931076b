"""2-D UNet: EfficientNet encoder + AdaBins-style BN decoder (mirror of occdepth/models/unet2d.py).

Eval path on the GPU (no autograd): every 3x3 convolution of the decoder runs on the in-repo kernels -- the first one of
each level as nine low-resolution tap GEMMs + the upsample-shift-accumulate kernel K12 + K10 over the skip channels
(no upsample+concat tensor), the second one on K10 (fused Winograd MFMA) or, at the 1/8 and 1/16 levels, on the
Winograd-domain transforms K9 around batched library GEMMs; the 1x1 heads on K11 write pixel-major rows for the lift;
the encoder's `conv_head` is folded into `conv2`.  Training path: 3x3 convolutions forward / data gradient on K10 through
`hip.conv2d_3x3_autograd`, the rest on ATen.  Faithful quirks: the decoder taps encoder features [4, 5, 6, 8, 11]
(conv_head output BEFORE bn2), `conv2` is a 1x1 conv with padding=1 (grows the 1/32 map by 2), and `up1` concatenates
the raw image.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd3d as _ag
from .. import hip
from .. import fused as _fused
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
            hit = (key, hip.matmul_operand(hip.winograd_weights(conv.weight), "b"), scale.contiguous(), shift.contiguous())
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

    TRAIN_K10 = os.environ.get("OCCDEPTH_TRAIN_K10", "1") == "1"

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
            per_row = 16.0 * B * ((W + 1) // 2) * (C + conv.out_channels) * 4
            rows = max(1, int(self.WINOGRAD_STRIP_MB * 2 ** 20 / per_row))
            return hip.conv2d_3x3_winograd(f, U, scale, shift, "leaky", act.negative_slope, strip_rows=rows)
        return hip.affine_act(conv(f), *bn_affine_cached(bn), "leaky", slope=act.negative_slope)

    # The first convolution of a level as "nine tap GEMMs at the LOW resolution + upsample-shift-accumulate + a small
    # convolution over the skip channels" (csrc/nchw2d.hip: upconv_gather_kernel): 9/16 of the Winograd-domain multiplies
    # of the upsampled channels and no upsample+concat tensor.  OCCDEPTH_UPCONV=0 restores upsample+concat -> K10 / K9.
    UPCONV = os.environ.get("OCCDEPTH_UPCONV", "1") == "1"
    # The tap GEMM (Cup -> 9 Cout: 720 ... 11520 rows, K = 160 ... 2560) is a plain large GEMM: the library's LDS-tiled
    # kernels (torch.matmul -> hipBLASLt / rocBLAS, ~100 TF/s) beat K11, whose per-wave register tiles re-read both
    # operands from L2 (measured 28-35 TF/s on these shapes).  OCCDEPTH_UPCONV_LIB_BELOW = B * h * w below which the
    # library is used (default: always) keeps K11 selectable for A/B.
    UPCONV_LIB_BELOW = int(os.environ.get("OCCDEPTH_UPCONV_LIB_BELOW", str(1 << 62)))
    UPCONV_FOLD_BELOW = int(os.environ.get("OCCDEPTH_UPCONV_FOLD_BELOW", "0"))        # B * h * w (experiment)
    UPCONV_FUSE_SKIP = os.environ.get("OCCDEPTH_UPCONV_FUSE_SKIP", "1") == "1"

    def _upconv_operands(self, conv, bn, cup):
        key = (_stamp(conv, bn), cup)
        hit = self.__dict__.get("_upconv_cache")
        if hit is None or hit[0] != key:
            scale, shift = bn_affine_cached(bn)
            if conv.bias is not None:
                shift = shift + scale * conv.bias.detach().float()
            w = conv.weight.detach().float()
            cout = w.shape[0]
            # rows t * Cout + co (t = ky * 3 + kx) of the tap GEMM, BatchNorm scale folded in
            w9 = (w[:, :cup] * scale.view(-1, 1, 1, 1)).permute(2, 3, 0, 1).reshape(9 * cout, cup).contiguous()
            wskip = (w[:, cup:] * scale.view(-1, 1, 1, 1)).contiguous()
            hit = (key, hip.pw_pack_weights(w9), hip.matmul_operand(w9, "a"),
                   hip.wino_pack_weights(w[:, cup:].contiguous(), scale), shift.contiguous(), wskip)
            self.__dict__["_upconv_cache"] = hit
        return hit[1:]

    @staticmethod
    def _tap_planes(shape, device):
        # K16 writes into rows on a 128-byte pitch; the library GEMM of the exact-fp32 mode (OCCDEPTH_GEMM_X3=0) allocates its own
        # dense result (a padded `out` would cost it a copy of the tap planes: 0.4 ms per frame)
        return hip.padded_rows(shape, device) if hip.GEMM_X3 else None

    def _first_conv_upconv(self, x, skip, conv, bn, act):
        wpk9, w9, upk_skip, shift, wskip = self._upconv_operands(conv, bn, x.shape[1])
        cout = conv.out_channels
        B, cup, h, w = x.shape
        if B > 1 and B * h * w <= self.UPCONV_FOLD_BELOW:
            # few pixels per image: ONE GEMM over the pixels of all images (the operand copy is small here)
            z = hip.matmul(w9, x.permute(1, 0, 2, 3).reshape(cup, B * h * w),
                           out=self._tap_planes((9 * cout, B * h * w), x.device)).view(9 * cout, B, h, w)
            batch_inner = True
        else:
            batch_inner = False
            if B * h * w < self.UPCONV_LIB_BELOW:
                xc = x if x.is_contiguous() else x.contiguous()
                # K16 (csrc/gemm_x3.hip) into tap planes on a 128-byte pitch (written 2x faster; K12 takes the strides)
                z = hip.matmul(w9, xc.view(B, cup, h * w), out=self._tap_planes((B, 9 * cout, h * w), x.device)).view(B, 9 * cout, h, w)
            else:
                z = hip.conv1x1(x, wpk9, 9 * cout)
        rw = (w - 1) / max(skip.shape[3] - 1, 1)
        if self.UPCONV_FUSE_SKIP and skip.shape[1] <= 4 and rw * 258.0 + 3.0 <= 191.0:
            # few skip channels (the 1/1 level: the raw image): their 3x3 convolution, the shift and the LeakyReLU ride in
            # K12's epilogue -- K10's fixed per-workgroup cost is 4x what a K = 3 convolution's bytes cost
            return hip.upconv_gather(z, cout, skip.shape[2:], batch_inner=batch_inner, skip=skip, wskip=wskip, shift=shift,
                                     slope=act.negative_slope)
        u = hip.upconv_gather(z, cout, skip.shape[2:], batch_inner=batch_inner)
        return hip.conv2d_3x3_fused(skip, upk_skip, cout, shift, "leaky", act.negative_slope, res=u, res_first=True)

    def _forward_train_cl(self, x, concat_with):
        """bf16-mode training (autograd3d.BF16_MFMA, BASELINE configs[3]): the level in channels-last memory -- the two 3x3
        convolutions (forward, data gradient, weight gradient) on the bf16-MFMA implicit-GEMM kernels K2b / K8b as X = 1
        volumes, BatchNorm + LeakyReLU as fused K13 passes on pixel rows, upsampling + concatenation as one channels-last launch
        with a gather backward (hip._UpCatClFn)."""
        from ..bn import bn_act
        n = self._net
        if x.dtype == torch.float32 and concat_with.dtype == torch.float32:
            f = hip.upsample_bilinear_cat_cl_autograd(x, concat_with)      # one launch forward, one gather launch backward
        else:
            up = F.interpolate(x.contiguous(memory_format=torch.channels_last), size=concat_with.shape[2:], mode="bilinear",
                               align_corners=True)
            f = torch.cat([up, concat_with.contiguous(memory_format=torch.channels_last)], dim=1)
        f = bn_act(n[1], _ag.conv2d_cl(f, n[0].weight, n[0].bias, padding=1), "leaky", n[2].negative_slope)
        return bn_act(n[4], _ag.conv2d_cl(f, n[3].weight, n[3].bias, padding=1), "leaky", n[5].negative_slope)

    def forward(self, x, concat_with):
        if _fused.on_gpu(x) and not needs_autograd(self) and x.dtype == torch.float32:
            n = self._net
            if (self.UPCONV and self.FUSED and n[0].kernel_size == (3, 3) and n[0].padding == (1, 1)
                    and n[0].stride == (1, 1) and x.shape[0] * n[0].out_channels <= 65535):
                f = self._first_conv_upconv(x, concat_with, n[0], n[1], n[2])
            else:
                # bilinear-up + concat in one HIP pass; BatchNorm + LeakyReLU fused into the Winograd output transform
                # or applied in one pass behind the MIOpen convolution
                f = self._conv_bn_act(hip.upsample_bilinear_cat(x, concat_with), n[0], n[1], n[2])
            return self._conv_bn_act(f, n[3], n[4], n[5])
        if _ag.BF16_MFMA and x.is_cuda and needs_autograd(self):
            return self._forward_train_cl(x, concat_with)
        if self.TRAIN_K10 and x.is_cuda and x.dtype == torch.float32 and concat_with.dtype == torch.float32:
            f = hip.upsample_bilinear_cat_autograd(x, concat_with)       # one pass instead of upsample + concat copy
        else:
            up = F.interpolate(x, size=concat_with.shape[2:], mode="bilinear", align_corners=True)
            f = torch.cat([up, concat_with], dim=1)
        if self.TRAIN_K10 and f.is_cuda and f.dtype in (torch.float32, torch.bfloat16, torch.float16):
            # training on the GPU: the two 3x3 convolutions (forward and data gradient) on K10, BatchNorm / LeakyReLU
            # on ATen; OCCDEPTH_TRAIN_K10=0 restores MIOpen for A/B
            from ..bn import bn_act
            n = self._net
            f = bn_act(n[1], hip.conv2d_3x3_autograd(f, n[0].weight, n[0].bias), "leaky", n[2].negative_slope)
            return bn_act(n[4], hip.conv2d_3x3_autograd(f, n[3].weight, n[3].bias), "leaky", n[5].negative_slope)
        from ..bn import bn_act
        n = self._net
        f = bn_act(n[1], n[0](f), "leaky", n[2].negative_slope)          # (= self._net(f); group-aware BatchNorm)
        return bn_act(n[4], n[3](f), "leaky", n[5].negative_slope)


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

    def _conv2_merged(self, f, conv_head):
        """conv2(conv_head(f)) as ONE convolution: both are 1x1 and nothing sits between them (the reference taps
        features[11] = conv_head's raw output, unet2d.py:146,183), so W = W_conv2 . W_head (formed once in float64) and the
        2560 -> 2560 GEMM on the 1/32 map disappears: 640 -> 2560 with conv2's bias and conv2's padding=1 frame."""
        key = _stamp(self.conv2, conv_head)
        hit = self.__dict__.get("_merged_head")
        if hit is None or hit[0] != key:
            w2 = self.conv2.weight.detach().double().flatten(1)
            wh = conv_head.weight.detach().double().flatten(1)
            w = (w2 @ wh).float().reshape(w2.shape[0], wh.shape[1], 1, 1).contiguous()
            hit = (key, w, hip.matmul_operand(w.view(w.shape[0], -1), "a") if w.is_cuda else None)
            self.__dict__["_merged_head"] = hit
        pad = self.conv2.padding
        if (hip.GEMM_X3 and _fused.on_gpu(f) and self.conv2.stride == (1, 1) and self.conv2.bias is not None
                and f.shape[1] % 8 == 0):
            # a 1x1 convolution with padding: the frame is bias only -- pad the (small) input with zeros and run ONE GEMM (K16)
            fp = F.pad(f, (pad[1], pad[1], pad[0], pad[0]))
            B, C, H, W = fp.shape
            w2d = hit[2] if hit[2] is not None else hit[1].view(hit[1].shape[0], C)
            return hip.matmul(w2d, fp.view(B, C, H * W), bias=self.conv2.bias.detach().float().contiguous()).view(B, -1, H, W)
        return F.conv2d(f, hit[1], self.conv2.bias, self.conv2.stride, pad)

    def forward(self, features, merged_head=None, on_scale=None):
        """on_scale(s, feature): optional callback right after the 1/s feature head has been launched (the caller may start work
        that only needs that scale -- FLoSP-Depth's DepthNet on the 1/8 feature -- while the finer levels are still running)."""
        taps = {16: features[8], 8: features[6], 4: features[5], 2: features[4], 1: features[0]}
        cl_train = _ag.BF16_MFMA and features[0].is_cuda and needs_autograd(self)
        if merged_head is not None:
            x = self._conv2_merged(features[10], merged_head)
        elif cl_train:
            x = _ag.conv2d_cl(features[11], self.conv2.weight, self.conv2.bias, padding=self.conv2.padding[0])
        else:
            x = self.conv2(features[11])
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
            if _fused.on_gpu(x) and not needs_autograd(self) and x.dtype == torch.float32 and pw_wins(x):
                # 1x1 convolution + bias on the MFMA GEMM (K11), written pixel-major: the 2D->3D lift gathers pixel rows,
                # so the (B, C, H, W) result is returned as a channels-last view and no transpose pass exists
                wpk, shift = pw_operands(self, head)
                res[f"1_{s}"] = hip.conv1x1(x, wpk, head.out_channels, shift, nhwc=True)
            elif cl_train:
                res[f"1_{s}"] = _ag.conv2d_cl(x, head.weight, head.bias)       # pixel rows: what the lift gathers from
            else:
                res[f"1_{s}"] = head(x)
            if on_scale is not None:
                on_scale(s, res[f"1_{s}"])
        return res


# eval fast path: the encoder stem (conv_stem + bn1 + swish) as one HIP launch instead of MIOpen + BatchNormFwdInfer + an
# activation pass (OCCDEPTH_STEM_FUSED=0 restores them for A/B)
STEM_FUSED = os.environ.get("OCCDEPTH_STEM_FUSED", "1") == "1"


class Encoder(nn.Module):
    def __init__(self, backend):
        super().__init__()
        self.original_model = backend

    HEAD = ("conv_head", "bn2", "act2", "global_pool", "classifier")

    def forward(self, x, skip_head=False):
        """The reference's feature list (unet2d.py:175-190).  skip_head: leave None in the slots of conv_head and of
        the modules behind it -- the decoder only taps conv_head's output (features[11]) and, in the eval path, gets it
        folded into its own first convolution (DecoderBN._conv2_merged); bn2 / act2 / pool / classifier are dead code."""
        features = [x]
        om = self.original_model
        names = list(om._modules)
        fused_stem = (names[:3] == ["conv_stem", "bn1", "act1"] and _fused.on_gpu(x) and not _fused.needs_autograd(self)
                      and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and STEM_FUSED
                      and om.conv_stem.kernel_size == (3, 3) and om.conv_stem.stride in ((1, 1), (2, 2))
                      and om.conv_stem.bias is None and om.conv_stem.groups == 1 and om.conv_stem.dilation == (1, 1))
        if fused_stem:
            # eval fast path: conv_stem + bn1 + act1 (swish) in ONE launch; the decoder taps features[4, 5, 6, 8, 11] only, so
            # the slots of the stem's own intermediates stay empty (like skip_head's)
            features += [None, None, hip.stem_conv3x3(x, om.conv_stem.weight, *_fused.bn_affine_cached(om.bn1),
                                                      om.conv_stem.stride[0], "swish")]
        for name, mod in self.original_model._modules.items():
            if fused_stem and name in ("conv_stem", "bn1", "act1"):
                continue
            if name == "blocks":
                for stage in mod._modules.values():
                    features.append(stage(features[-1]))
            elif skip_head and name in self.HEAD:
                features.append(None)
            elif isinstance(mod, nn.modules.batchnorm._BatchNorm):
                from ..bn import module_call
                features.append(module_call(mod, features[-1]))       # (per-view statistics in view-batched training)
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

    # eval path: conv_head folded into the decoder's conv2 (OCCDEPTH_MERGE_HEAD=0 restores the two GEMMs for A/B)
    MERGE_HEAD = os.environ.get("OCCDEPTH_MERGE_HEAD", "1") == "1"

    def forward(self, x, **kwargs):
        head = getattr(self.encoder.original_model, "conv_head", None)
        if (self.MERGE_HEAD and self.use_decoder and _fused.on_gpu(x) and not needs_autograd(self) and x.dtype == torch.float32
                and isinstance(head, nn.Conv2d) and head.kernel_size == (1, 1) and head.stride == (1, 1)
                and head.bias is None and head.groups == 1 and self.decoder.conv2.kernel_size == (1, 1)
                and list(self.encoder.original_model._modules)[:5] == ["conv_stem", "bn1", "act1", "blocks", "conv_head"]
                and len(self.encoder.original_model.blocks) == 7):      # i.e. features[10] feeds conv_head, features[11] conv2
            return self.decoder(self.encoder(x, skip_head=True), merged_head=head, **kwargs)
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
