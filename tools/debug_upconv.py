"""dev: per-level A/B of the decoder's first-convolution forms on the nyu_small / kitti_small product models."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_cases as gc
from test_oracle_vs_golden import build_product
from occdepth_amd.models.unet2d import UpSampleBN
torch.backends.cuda.matmul.allow_tf32 = False
for name in sys.argv[1:] or ["nyu_small"]:
    m, cfg, sd = build_product(name)
    m = m.cuda().eval()
    batch = gc.occdepth_batch(name)
    img = batch["img"][:, 0].cuda()
    dec = m.net_rgb.decoder
    with torch.no_grad():
        feats = m.net_rgb.encoder(img)
        x = dec.conv2(feats[11])
        taps = {16: feats[8], 8: feats[6], 4: feats[5], 2: feats[4], 1: feats[0]}
        for s in (16, 8, 4, 2, 1):
            up = getattr(dec, f"up{s}")
            outs = {}
            for tag, on, lib in (("concat", False, 0), ("upconv_matmul", True, 1 << 62), ("upconv_k11", True, 0)):
                UpSampleBN.UPCONV, UpSampleBN.UPCONV_LIB_BELOW = on, lib
                outs[tag] = up(x, taps[s])
            ref = outs["concat"]
            print(name, "level", s, "x", tuple(x.shape), "skip", tuple(taps[s].shape), "contig", x.is_contiguous(), taps[s].is_contiguous(),
                  {k: f"{float((v - ref).abs().max() / ref.abs().max()):.2e}" for k, v in outs.items() if k != "concat"})
            x = ref
