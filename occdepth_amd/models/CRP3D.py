"""3-D Context Relation Prior (mirror of occdepth/models/CRP3D.py:9-97).

Eval-mode dataflow on the HIP kernels (N = voxels at this level, M = mega voxels = N/8):
  x_agg   = ASPP(x)                                              6 x K2 (3x3x3, dilated)
  mega    = mega_context(x_agg)          (M rows x 2C)           K2, stride 2
  logit_r = context_prior_logits[r](x_agg)   (N rows x M)        K16 (round 6): the R 1x1x1 convolutions of an image as ONE
                                                                 batched launch (column bias)  -> P_logits[:, r]
  ctx_r   = sigmoid(logit_r) @ mega          (N rows x 2C)       K16 (round 5): the R products of an image as ONE
                                                                 batched launch, sigmoid applied while A is staged
  x       = resize([x | ctx_0 .. ctx_{R-1}])                     ctx_r written straight into its
                                                                 channel slice of the concat rows
"""
import torch
import torch.nn as nn

from .. import hip
from ..autograd3d import Conv3d, ConvTranspose3d  # noqa: F401  (nn.Conv3d subclasses: HIP forward/backward in training)
from ..fused import ACT_SIGMOID, ConvPlan, Vox, as_vox, gemm_rows, needs_autograd, pack_rows, run_parallel
from .modules import ASPP, Process
import os

# the relation products sigmoid(P_logits) @ mega as one batched K16 launch (OCCDEPTH_CRP_K16=0: four K2 launches, A/B)
CRP_PRODUCTS_K16 = os.environ.get("OCCDEPTH_CRP_K16", "1") == "1"
# round 6: the R relation-logit 1x1x1 convolutions (reference :54-62) as ONE batched K16 launch per image -- rows = voxels, the
# weights transposed + split once (GemmPacked role "b"), the convolution bias as K16's column bias (ABI 13); four generic K2
# launches at 38 TF/s before (OCCDEPTH_CRP_LOGITS_K16=0 restores them)
CRP_LOGITS_K16 = os.environ.get("OCCDEPTH_CRP_LOGITS_K16", "1") == "1"


class CPMegaVoxels(nn.Module):
    def __init__(self, feature, size, n_relations=4, bn_momentum=0.0003):
        super().__init__()
        self.size = tuple(int(s) for s in size)
        self.n_relations = n_relations
        print("n_relations", self.n_relations)
        self.flatten_size = self.size[0] * self.size[1] * self.size[2]
        self.feature = feature
        self.context_feature = feature * 2
        self.flatten_context_size = (self.size[0] // 2) * (self.size[1] // 2) * (self.size[2] // 2)
        padding = tuple((s + 1) % 2 for s in self.size)

        self.mega_context = nn.Sequential(
            Conv3d(feature, self.context_feature, stride=2, padding=padding, kernel_size=3))
        self.context_prior_logits = nn.ModuleList([
            nn.Sequential(Conv3d(self.feature, self.flatten_context_size, padding=0, kernel_size=1))
            for _ in range(n_relations)])
        self.aspp = ASPP(feature, [1, 2, 3])
        self.resize = nn.Sequential(
            Conv3d(self.context_feature * self.n_relations + feature, feature, kernel_size=1, padding=0,
                      bias=False),
            Process(feature, nn.BatchNorm3d, bn_momentum, dilations=[1]))
        self._plans = None

    def forward_vox(self, x):
        """x: Vox (B, X, Y, Z, C) -> {"x": Vox, "P_logits": (B, R, M, N) tensor view}."""
        if self._plans is None:
            self._plans = {
                "mega": ConvPlan(self.mega_context[0]),
                "logits": [ConvPlan(seq[0]) for seq in self.context_prior_logits],
                "resize": ConvPlan(self.resize[0]),
            }
        pl = self._plans
        B, dims, dev = x.batch, x.dims, x.buf.device
        C, C2, R = self.feature, self.context_feature, self.n_relations
        M, N = self.flatten_context_size, self.flatten_size
        if dims != self.size:
            raise RuntimeError(f"CPMegaVoxels built for {self.size}, got {dims}")
        if C % 8 or C2 % 8:
            raise NotImplementedError("CRP concat rows need feature % 8 == 0")

        cat = torch.empty((B,) + dims + (C + R * C2,), device=dev, dtype=torch.float32)
        cat[..., :C] = x.buf[..., x.coff:x.coff + C]           # torch.cat's first operand
        x_agg = self.aspp.forward_vox(x)
        m_cs = hip.round_up(M, 8)
        logits = torch.empty((R * B,) + dims + (m_cs,), device=dev, dtype=torch.float32)
        mega_plan = pl["mega"]
        mega_plan._prepare()
        for plan in pl["logits"]:
            plan._prepare()                                     # (weight packing allocates: not inside the side streams)
        mega = Vox.empty(B, mega_plan.out_dims(x_agg.dims), mega_plan.cout, dev)   # (B, X/2, Y/2, Z/2, 2C): rows = mega voxels
        lgs = [Vox(logits[r * B:(r + 1) * B], M) for r in range(R)]
        # mega_context (64 workgroups) and the R relation-logit convolutions all read x_agg and nothing else
        if (x.buf.is_cuda and CRP_LOGITS_K16 and hip.GEMM_X3 and C % 8 == 0 and x_agg.coff == 0 and x_agg.cs % 4 == 0
                and M % 4 == 0 and x_agg.buf.dtype == torch.float32):
            wb, bias_n = self._logit_operands()
            if m_cs != M:
                logits.zero_()                                  # (the K2 form wrote the pad columns as zeros)

            def logit_gemm():
                for b in range(B):
                    a_op = x_agg.buf[b].reshape(N, x_agg.cs)[:, :C]                      # rows = voxels, k = channels
                    out = torch.as_strided(logits, (R, N, M), (B * N * m_cs, m_cs, 1), b * N * m_cs)
                    hip.gemm_x3(a_op, wb, out=out, bias_n=bias_n)
            first = [lambda: mega_plan(x_agg, out=mega), logit_gemm]
        else:
            first = [lambda: mega_plan(x_agg, out=mega)] + [(lambda r=r: pl["logits"][r](x_agg, out=lgs[r])) for r in range(R)]
        if x.buf.is_cuda:
            run_parallel(first)
        else:
            for th in first:
                th()
        if x.buf.is_cuda and CRP_PRODUCTS_K16 and hip.GEMM_X3 and M % 8 == 0 and m_cs % 4 == 0:
            # round 5: the R relation products sigmoid(logits_r) @ mega of one image as ONE batched K16 launch (3-way bf16
            # split, float32-level accuracy): A = the relations' logit rows (N x M each, the sigmoid applied while they are
            # staged), B = the mega rows (M x 2C, shared over the relations), C = the relations' slices of the concat rows.
            # Four generic K2 launches at 35 TF/s (0.25 ms per frame at config 2) before.
            Ct = C + R * C2
            for b in range(B):
                a_op = torch.as_strided(logits, (R, N, M), (B * N * m_cs, m_cs, 1), b * N * m_cs)
                b_op = mega.buf[b].reshape(M, mega.cs)[:, :C2]
                out = torch.as_strided(cat, (R, N, C2), (C2, Ct, 1), b * N * Ct + C)
                hip.gemm_x3(a_op, b_op, out=out, sigmoid_a=True)
        else:
            packed = []                                         # the mega rows are the B operand of all R products
            for b in range(B):
                rows_b = mega.buf[b].reshape(M, mega.cs)[:, :C2]
                packed.append(pack_rows(rows_b if rows_b.is_contiguous() else rows_b.contiguous()))

            def product(r):
                def run():
                    for b in range(B):
                        gemm_rows(Vox(lgs[r].buf[b:b + 1], M), packed[b], Vox(cat[b:b + 1], C2, C + r * C2), act_in=ACT_SIGMOID)
                return run

            second = [product(r) for r in range(R)]             # sigmoid(logits_r) @ mega: independent, disjoint outputs
            if x.buf.is_cuda:
                run_parallel(second)
            else:
                for th in second:
                    th()
        y = pl["resize"](Vox(cat, C + R * C2))
        y = self.resize[1].forward_vox(y)
        p_logits = logits.view(R, B, N, m_cs)[..., :M].permute(1, 0, 3, 2)
        return {"P_logits": p_logits, "x": y}

    def _logit_operands(self):
        """(GemmPacked role-"b" image of the R transposed 1x1x1 weights (R, C, M), column bias (R, M)), cached until a weight
        or bias changes."""
        from ..fused import _stamp
        convs = [seq[0] for seq in self.context_prior_logits]
        key = _stamp(*convs)
        hit = self.__dict__.get("_logit_ops")
        if hit is None or hit[0] != key:
            w = torch.stack([c.weight.detach().float().reshape(c.out_channels, c.in_channels).t().contiguous() for c in convs])
            bias = torch.stack([c.bias.detach().float() if c.bias is not None else
                                torch.zeros(c.out_channels, device=w.device) for c in convs]).contiguous()
            hit = (key, hip.GemmPacked(w, "b"), bias)
            self.__dict__["_logit_ops"] = hit
        return hit[1], hit[2]

    def _forward_autograd(self, inp):
        bs = inp.shape[0]
        x_agg = self.aspp(inp)
        mega = self.mega_context(x_agg).reshape(bs, self.context_feature, -1).transpose(1, 2)  # (bs, M, 2C)
        logits, ctx = [], []
        for head in self.context_prior_logits:
            lg = head(x_agg).reshape(bs, self.flatten_context_size, self.flatten_size)
            logits.append(lg.unsqueeze(1))
            ctx.append(torch.bmm(torch.sigmoid(lg.transpose(1, 2)), mega))                        # (bs, N, 2C)
        ctx = torch.cat(ctx, dim=2).transpose(1, 2).reshape(bs, -1, *self.size)
        x = self.resize(torch.cat([inp, ctx], dim=1))
        return {"P_logits": torch.cat(logits, dim=1), "x": x}

    def forward(self, input):
        if needs_autograd(self):
            return self._forward_autograd(input)
        ret = self.forward_vox(as_vox(input))
        ret["x"] = ret["x"].ncdhw()
        return ret
