"""Relation (context-prior) loss, mirror of occdepth/loss/CRP_loss.py:4-24.

On the GPU the loss is ONE statistics pass over the (B, R, M, N) relation logits (`hip.relation_bce_stats`, csrc/loss.hip)
and its gradient one more pass (`hip.relation_bce_grad`): the class-balanced BCE-with-logits is
    1 / (R T) * sum_r ( pos_weight_r * sum_{y=1} softplus(-x) + sum_{y=0} softplus(x) ),   pos_weight_r = #neg_r / #pos_r,
i.e. a function of 3 R sums (fixed point: deterministic), so the permute / reshape / cat / float() / count / BCE chain of the
reference (~20 launches over 8.4 M elements at config 2, the labels converted to float32) disappears.
"""
import torch
import torch.nn.functional as F

from .. import hip


def _stack_labels(CP_mega_matrices, device):
    if isinstance(CP_mega_matrices, (list, tuple)):
        CP_mega_matrices = torch.stack(list(CP_mega_matrices))
    return CP_mega_matrices.to(device)


class _RelationBCE(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits, labels):
        B, R, M, N = logits.shape
        st = hip.relation_bce_stats(logits, labels).double()             # (R, 3): #pos, Spos, Sneg (Q24)
        total = float(B * M * N)
        cnt_pos = st[:, 0]
        pos_weight = (total - cnt_pos) / cnt_pos                         # (inf / nan without positives, like the reference)
        ctx.save_for_backward(logits, labels, pos_weight)
        return ((pos_weight * st[:, 1] + st[:, 2]).sum() / (hip.REL_Q24 * R * total)).float()

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        logits, labels, pos_weight = ctx.saved_tensors
        B, R, M, N = logits.shape
        k = g.double() / float(R * B * M * N)
        coef = torch.stack([pos_weight * k, k.expand_as(pos_weight)], 1).float().contiguous()
        return hip.relation_bce_grad(logits, labels, coef), None


def compute_super_CP_multilabel_loss(pred_logits, CP_mega_matrices):
    """pred_logits (bs, n_relations, n_mega_voxels, N); CP_mega_matrices: bs x (n_relations, N, n_mega_voxels).
    Class-balanced BCE-with-logits: pos_weight[r] = #negatives / #positives of relation r over the batch."""
    bs, n_relations = pred_logits.shape[:2]
    labels = _stack_labels(CP_mega_matrices, pred_logits.device)
    if hip.relation_bce_usable(pred_logits, labels):
        return _RelationBCE.apply(pred_logits, labels.contiguous())
    logits = pred_logits.permute(1, 0, 3, 2).reshape(n_relations, -1)                    # (R, bs * N * mega)
    labels = labels.permute(1, 0, 2, 3).reshape(n_relations, -1).float()
    cnt_pos = labels.sum(1)
    cnt_neg = (labels == 0).sum(1)
    pos_weight = (cnt_neg / cnt_pos).unsqueeze(1)
    return F.binary_cross_entropy_with_logits(logits, labels, pos_weight=pos_weight)
