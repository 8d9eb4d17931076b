"""Relation (context-prior) loss, mirror of occdepth/loss/CRP_loss.py:4-24."""
import torch
import torch.nn.functional as F


def compute_super_CP_multilabel_loss(pred_logits, CP_mega_matrices):
    """pred_logits (bs, n_relations, n_mega_voxels, N); CP_mega_matrices: bs x (n_relations, N, n_mega_voxels).
    Class-balanced BCE-with-logits: pos_weight[r] = #negatives / #positives of relation r over the batch."""
    bs, n_relations = pred_logits.shape[:2]
    logits = pred_logits.permute(1, 0, 3, 2).reshape(n_relations, -1)                    # (R, bs * N * mega)
    if isinstance(CP_mega_matrices, (list, tuple)):
        CP_mega_matrices = torch.stack(list(CP_mega_matrices))
    labels = CP_mega_matrices.to(pred_logits.device).permute(1, 0, 2, 3).reshape(n_relations, -1).float()
    cnt_pos = labels.sum(1)
    cnt_neg = (labels == 0).sum(1)
    pos_weight = (cnt_neg / cnt_pos).unsqueeze(1)
    return F.binary_cross_entropy_with_logits(logits, labels, pos_weight=pos_weight)
