"""Convolution geometries of the 3-D stack's training graph (shared by the CPU host-logic and the GPU tests)."""
import torch

# name: (cin, cout, kernel, stride, padding, dilation, dims, bias)
CONV_CASES = {
    "head_k3": (32, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), (12, 10, 8), True),      # modules.py:76,98,118
    "head_k3_d2": (32, 32, (3, 3, 3), (1, 1, 1), (2, 2, 2), (2, 2, 2), (9, 10, 8), False),   # modules.py:25
    "head_k3_d3": (16, 24, (3, 3, 3), (1, 1, 1), (3, 3, 3), (3, 3, 3), (8, 9, 10), False),
    "classes": (34, 20, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), (6, 7, 8), True),         # modules.py:118 (34 -> 20)
    "k1": (64, 48, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (6, 5, 8), False),             # DDR.py:33,42
    "axis_z": (16, 16, (1, 1, 3), (1, 1, 1), (0, 0, 2), (1, 1, 2), (5, 6, 12), False),        # DDR.py:38 (dilated)
    "axis_y_s2": (16, 16, (1, 3, 1), (1, 2, 1), (0, 1, 0), (1, 1, 1), (6, 10, 8), False),     # DDR.py:38 (stride 2)
    "axis_x_s2_odd": (8, 16, (3, 1, 1), (2, 1, 1), (1, 0, 0), (1, 1, 1), (9, 4, 8), False),
    "k3_s2_p1": (16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), (8, 8, 8), True),        # CRP3D.py:33 (even size)
    "k3_s2_p0": (16, 32, (3, 3, 3), (2, 2, 2), (0, 0, 0), (1, 1, 1), (9, 7, 9), True),        # CRP3D.py:33 (odd size)
    "wide": (72, 100, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (4, 4, 6), True),           # channel tiles > 1, ragged
    "nyu_z15": (8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), (5, 4, 15), False),         # Z not a multiple of 8
    # the reduced NYU config of the train-step fixture: feature 20 -> planes 5 (ragged channel counts everywhere)
    "nyu_p5_axis": (5, 5, (1, 1, 3), (1, 1, 1), (0, 0, 1), (1, 1, 1), (10, 6, 10), False),
    "nyu_p5_s2": (5, 5, (1, 3, 1), (1, 2, 1), (0, 1, 0), (1, 1, 1), (10, 6, 10), False),
    "nyu_p5_s2x": (5, 5, (3, 1, 1), (2, 1, 1), (1, 0, 0), (1, 1, 1), (5, 3, 5), False),
    "nyu_20_5": (20, 5, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (10, 6, 10), False),
    "nyu_5_20": (5, 20, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (10, 6, 10), False),
    "nyu_head": (20, 12, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), (20, 12, 20), True),
    "nyu_aspp": (20, 20, (3, 3, 3), (1, 1, 1), (3, 3, 3), (3, 3, 3), (20, 12, 20), False),
    # the 2-D decoder as X = 1 volumes (bf16-mode training): 3x3, the 1x1 heads, and `conv2` = 1x1 with padding 1, whose
    # data gradient crops (negative pad)
    "dec_3x3": (24, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), (1, 9, 20), True),             # unet2d.py:24-46
    "dec_head_1x1": (16, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), (1, 9, 20), True),          # unet2d.py:120-131
    "dec_conv2_1x1_p1": (32, 24, (1, 1, 1), (1, 1, 1), (0, 1, 1), (1, 1, 1), (1, 5, 7), True),      # unet2d.py:65-67
}
# name: (cin, cout, kernel, stride, padding, output_padding, dims, bias)
CONVT_CASES = {
    "up_s2": (32, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), (4, 5, 4), True),           # modules.py:192
    "up_s1": (16, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), (0, 0, 0), (5, 4, 8), False),
    "nyu_up": (40, 20, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), (5, 3, 5), True),
}


def conv_tensors(name, batch=2):
    cin, cout, k, s, p, d, dims, bias = CONV_CASES[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    x = torch.randn(batch, cin, *dims, generator=g)
    w = torch.randn(cout, cin, *k, generator=g) * 0.2
    b = torch.randn(cout, generator=g) if bias else None
    return x, w, b


def convt_tensors(name, batch=2):
    cin, cout, k, s, p, op, dims, bias = CONVT_CASES[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    x = torch.randn(batch, cin, *dims, generator=g)
    w = torch.randn(cin, cout, *k, generator=g) * 0.2
    b = torch.randn(cout, generator=g) if bias else None
    return x, w, b
