"""Scene-completion losses (mirror of occdepth/loss/ssc_loss.py + the inline frustum loss of
occdepth/models/OccDepth.py:487-521), computed from ONE statistics pass over the logits.

The reference runs a softmax per loss, a python loop over the classes and a python loop over the 64 frustums --
about a hundred full passes over the (B, C, X, Y, Z) tensor and as many host syncs (`if torch.sum(...) > 0`).
All of those losses depend on the logits only through a few hundred sums over voxels; `hip.ssc_loss_stats`
(K5, csrc/loss.hip) produces them in one pass, the formulas below are then evaluated on that small float64
vector (no host sync: the data-dependent branches become `torch.where`), and `hip.ssc_loss_grad` (K6) maps
d loss / d sums back to d loss / d logits in one more pass.
"""
import torch

from .. import hip


class _SscStats(torch.autograd.Function):
    """logits (B, C, ...) -> float64 vector of sums [P(C) | N(C) | T(C) | M | CEnum | CEden | F(F*C)]."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits, target, masks, weights, map_occ):
        # (B, C, S) planes and the 3-D stack's channels-last voxel rows are both read in place (round 5: the per-step
        #  168 MB layout copies of `ssc_logit` -- here, in the backward and in the metric update -- are gone); any other
        #  layout is made contiguous inside hip.ssc_loss_stats
        raw = hip.ssc_loss_stats(logits, target, masks, weights, map_occ)
        C = logits.shape[1]
        F = 0 if masks is None else masks.shape[1]
        ctx.save_for_backward(logits, target, masks, weights)
        ctx.map_occ = map_occ
        return raw.double() * hip.ssc_stats_scale(C, F, raw.device)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        logits, target, masks, weights = ctx.saved_tensors
        grad = hip.ssc_loss_grad(logits, target, masks, weights, g.float().contiguous(), ctx.map_occ)
        return grad, None, None, None, None


def _prep_target(target):
    return target if target.dtype == torch.uint8 else target.to(torch.uint8)


def _prep_masks(masks):
    if masks is None:
        return None
    if isinstance(masks, (list, tuple)):
        masks = torch.stack(list(masks))
    masks = masks.contiguous()
    return masks.view(torch.uint8) if masks.dtype == torch.bool else masks.to(torch.uint8)


def ssc_stats(pred, target, class_weights=None, frustums_masks=None, map_occ=False):
    w = None if class_weights is None else class_weights.to(device=pred.device, dtype=torch.float32).contiguous()
    return _SscStats.apply(pred.float(), _prep_target(target).contiguous(), _prep_masks(frustums_masks), w,
                           bool(map_occ))


def _neg_log(x):
    """F.binary_cross_entropy(x, ones): -log x with torch's clamp of the log at -100."""
    return -torch.clamp(torch.log(x), min=-100.0)


def _split(st, C):
    return st[:C], st[C:2 * C], st[2 * C:3 * C], st[3 * C], st[3 * C + 1], st[3 * C + 2], st[3 * C + 3:]


def ce_from_stats(st, C):
    return st[3 * C + 1] / st[3 * C + 2]


def geo_scal_from_stats(st, C):
    """ssc_loss.py:17-41 with nonempty = 1 - p_0 over labelled voxels."""
    P, N, T, M = st[:C], st[C:2 * C], st[2 * C:3 * C], st[3 * C]
    inter = (M - T[0]) - (P[0] - N[0])
    precision = inter / (M - P[0])
    recall = inter / (M - T[0])
    spec = N[0] / T[0]
    return _neg_log(precision) + _neg_log(recall) + _neg_log(spec)


def sem_scal_from_stats(st, C):
    """ssc_loss.py:44-87: per class present in the target, BCE(precision) [if sum p > 0] + BCE(recall)
    + BCE(specificity) [if any other labelled voxel]; mean over the classes present."""
    P, N, T, M = st[:C], st[C:2 * C], st[2 * C:3 * C], st[3 * C]
    present = T > 0
    one = torch.ones_like(P)
    zero = torch.zeros_like(P)
    # every masked-out lane gets the harmless argument 1 BEFORE the log: where() alone would still back-propagate
    # 0 * inf = NaN through log(0) of an absent class
    has_p = present & (P > 0)
    others = present & ((M - T) > 0)
    precision = torch.where(has_p, N / torch.where(has_p, P, one), one)
    recall = torch.where(present, N / torch.where(present, T, one), one)
    spec = torch.where(others, ((M - T) - (P - N)) / torch.where(others, M - T, one), one)
    per_class = _neg_log(precision) + _neg_log(recall) + _neg_log(spec)
    return torch.where(present, per_class, zero).sum() / present.sum()


def frustum_from_stats(st, C, frustums_class_dists):
    """OccDepth.py:487-521: KL(target proportion || predicted proportion) per frustum, over its non-zero classes,
    averaged over the frustums that have both probability mass and ground-truth counts."""
    Fm = st[3 * C + 3:].view(-1, C)
    if isinstance(frustums_class_dists, (list, tuple)):
        frustums_class_dists = torch.stack(list(frustums_class_dists))
    cnt = frustums_class_dists.to(device=st.device, dtype=torch.float64).sum(0)          # (F, C)
    total_cnt = cnt.sum(1, keepdim=True)
    total_prob = Fm.sum(1, keepdim=True)
    valid = ((total_prob > 0) & (total_cnt > 0)).squeeze(1)
    one = torch.ones_like(total_cnt)
    tgt = cnt / torch.where(total_cnt > 0, total_cnt, one)
    cum = Fm / torch.where(total_prob > 0, total_prob, one)
    sel = valid.unsqueeze(1) & (tgt != 0)                      # KL_sep: only the non-zero classes of the target
    onefc = torch.ones_like(tgt)
    tgt_s = torch.where(sel, tgt, onefc)                       # masked lanes: log(1) = 0, finite gradient
    cum_s = torch.where(sel, cum, onefc)
    kl = (torch.where(sel, tgt, torch.zeros_like(tgt)) * (torch.log(tgt_s) - torch.log(cum_s))).sum(1)
    return torch.where(valid, kl, torch.zeros_like(kl)).sum() / valid.sum()


# ---- the reference's function surface (ssc_loss.py) -----------------------------------------------------------
def KL_sep(p, target):
    """KL divergence on the non-zero classes of `target` (ssc_loss.py:6-14)."""
    nz = target != 0
    return (target[nz] * (torch.log(target[nz]) - torch.log(p[nz]))).sum()


def CE_ssc_loss(pred, target, class_weights):
    C = pred.shape[1]
    return ce_from_stats(ssc_stats(pred, target, class_weights), C).to(pred.dtype)


def sem_scal_loss(pred, ssc_target):
    C = pred.shape[1]
    return sem_scal_from_stats(ssc_stats(pred, ssc_target), C).to(pred.dtype)


def geo_scal_loss(pred, ssc_target):
    C = pred.shape[1]
    return geo_scal_from_stats(ssc_stats(pred, ssc_target), C).to(pred.dtype)


def frustum_proportion_loss(pred, frustums_masks, frustums_class_dists):
    C = pred.shape[1]
    dummy = torch.full((pred.shape[0],) + tuple(pred.shape[2:]), 255, dtype=torch.uint8, device=pred.device)
    return frustum_from_stats(ssc_stats(pred, dummy, None, frustums_masks), C, frustums_class_dists).to(pred.dtype)


def ssc_losses(pred, target, class_weights, frustums_masks=None, frustums_class_dists=None, ce=True, sem_scal=True,
               geo_scal=True):
    """Every switched-on scene-completion loss of one step from a single statistics pass -> dict of scalars."""
    C = pred.shape[1]
    st = ssc_stats(pred, target, class_weights, frustums_masks)
    out = {}
    if ce:
        out["loss_ssc"] = ce_from_stats(st, C).to(pred.dtype)
    if sem_scal:
        out["loss_sem_scal"] = sem_scal_from_stats(st, C).to(pred.dtype)
    if geo_scal:
        out["loss_geo_scal"] = geo_scal_from_stats(st, C).to(pred.dtype)
    if frustums_masks is not None:
        out["loss_frustums"] = frustum_from_stats(st, C, frustums_class_dists).to(pred.dtype)
    return out


def occ_ce_loss(occ_pred, target, class_weights_occ):
    """CE of the cascade occupancy head against the target collapsed to {empty, occupied, 255}
    (OccDepth.py:411-418); the relabelling happens inside the statistics kernel."""
    C = occ_pred.shape[1]
    return ce_from_stats(ssc_stats(occ_pred, target, class_weights_occ, map_occ=True), C).to(occ_pred.dtype)
