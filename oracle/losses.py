"""CPU restatement (numpy, float64) of the reference's training-step losses and SSC metrics.

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by the product package.
Pinned by tests/golden/losses.npz, which tests/golden/make_golden.py generates by running the real
reference functions and the real `OccDepth.step` (with its forward replaced by a fixed out_dict).

Reference lines restated here:
  occdepth/loss/ssc_loss.py:6-14    KL_sep
  occdepth/loss/ssc_loss.py:17-41   geo_scal_loss
  occdepth/loss/ssc_loss.py:44-87   sem_scal_loss
  occdepth/loss/ssc_loss.py:90-99   CE_ssc_loss
  occdepth/loss/CRP_loss.py:4-24    compute_super_CP_multilabel_loss
  occdepth/loss/depth_loss.py:7-87  DepthClsLoss
  occdepth/models/OccDepth.py:378-533  step (loss assembly + the inline frustum-proportion loss, lines 487-521)
  occdepth/loss/sscMetrics.py:70-204   SSCMetrics.add_batch / get_stats
"""
import numpy as np


def softmax(x, axis=1):
    x = np.asarray(x, np.float64)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def _neg_log(x):
    """F.binary_cross_entropy(x, ones) for a scalar: -max(log x, -100) (torch clamps the log)."""
    with np.errstate(divide="ignore"):
        return -max(np.log(x), -100.0)


def ce_ssc_loss(pred, target, class_weights, ignore=255):
    """ssc_loss.py:90-99: weighted CE, mean over the weights of the labelled voxels."""
    pred = np.asarray(pred, np.float64)
    c = pred.shape[1]
    x = np.moveaxis(pred, 1, -1).reshape(-1, c)
    t = np.asarray(target).reshape(-1).astype(np.int64)
    keep = t != ignore
    x, t = x[keep], t[keep]
    m = x.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(x - m).sum(1))
    nll = lse - x[np.arange(len(t)), t]
    w = np.asarray(class_weights, np.float64)[t]
    return float((w * nll).sum() / w.sum())


def geo_scal_loss(pred, target):
    p = softmax(pred, 1)
    t = np.asarray(target)
    mask = t != 255
    empty = p[:, 0][mask]
    nonempty = 1.0 - empty
    nt = (t != 0)[mask].astype(np.float64)
    inter = (nt * nonempty).sum()
    precision = inter / nonempty.sum()
    recall = inter / nt.sum()
    spec = ((1 - nt) * empty).sum() / (1 - nt).sum()
    return _neg_log(precision) + _neg_log(recall) + _neg_log(spec)


def sem_scal_loss(pred, target):
    p_all = softmax(pred, 1)
    t_all = np.asarray(target)
    mask = t_all != 255
    t = t_all[mask]
    loss, count = 0.0, 0.0
    for i in range(p_all.shape[1]):
        p = p_all[:, i][mask]
        ct = (t == i).astype(np.float64)
        if ct.sum() > 0:
            count += 1.0
            nom = (p * ct).sum()
            lc = 0.0
            if p.sum() > 0:
                lc += _neg_log(nom / p.sum())
            lc += _neg_log(nom / ct.sum())
            if (1 - ct).sum() > 0:
                lc += _neg_log(((1 - p) * (1 - ct)).sum() / (1 - ct).sum())
            loss += lc
    return loss / count


def frustum_proportion_loss(pred, frustums_masks, frustums_class_dists):
    """OccDepth.py:487-521.  masks (bs, F, X, Y, Z) bool, dists (bs, F, C)."""
    p = softmax(pred, 1)                                   # NOT masked by target != 255 (reference behaviour)
    bs, c = p.shape[:2]
    masks = np.asarray(frustums_masks).astype(np.float64)
    cnt = np.asarray(frustums_class_dists, np.float64).sum(0)           # (F, C)
    loss, nonempty = 0.0, 0
    for f in range(masks.shape[1]):
        prob = (masks[:, f][:, None] * p).transpose(1, 0, 2, 3, 4).reshape(c, -1)
        cum = prob.sum(1)
        total_cnt, total_prob = cnt[f].sum(), prob.sum()
        if total_prob > 0 and total_cnt > 0:
            tgt = cnt[f] / total_cnt
            cum = cum / total_prob
            nz = tgt != 0
            with np.errstate(divide="ignore"):
                loss += float((tgt[nz] * (np.log(tgt[nz]) - np.log(cum[nz]))).sum())     # KL_sep
            nonempty += 1
    return loss / nonempty


def relation_loss(pred_logits, cp_mega_matrices):
    """CRP_loss.py:4-24.  pred_logits (bs, R, A, B); CP matrices: bs x (R, B, A)."""
    pred_logits = np.asarray(pred_logits, np.float64)
    bs, r = pred_logits.shape[:2]
    logits = np.concatenate([pred_logits[i].transpose(0, 2, 1).reshape(r, -1) for i in range(bs)], 1).T
    labels = np.concatenate([np.asarray(cp_mega_matrices[i], np.float64).reshape(r, -1) for i in range(bs)], 1).T
    cnt_neg = (labels == 0).sum(0)
    cnt_pos = labels.sum(0)
    pw = cnt_neg / cnt_pos
    # BCEWithLogits with pos_weight: pw*y*softplus(-x) + (1-y)*softplus(x), mean over all elements
    sp = lambda z: np.maximum(z, 0) + np.log1p(np.exp(-np.abs(z)))
    return float((pw[None] * labels * sp(-logits) + (1 - labels) * sp(logits)).mean())


def downsampled_gt_depth(gt, factor, d_bound):
    """depth_loss.py:14-52: min-pool of non-zero depths over factor x factor, binned, one-hot (bin 0 dropped)."""
    gt = np.asarray(gt, np.float64)
    bn, h, w = gt.shape
    channels = int((d_bound[1] - d_bound[0]) / d_bound[2])
    g = gt.reshape(bn, h // factor, factor, w // factor, factor).transpose(0, 1, 3, 2, 4).reshape(-1, factor * factor)
    g = np.where(g == 0.0, 1e5, g).min(-1)
    g = (g.astype(np.float32) - np.float32(d_bound[0] - d_bound[2])) / np.float32(d_bound[2])   # fp32 like the reference
    g = np.where((g < channels + 1) & (g >= 0.0), g, 0.0)
    idx = g.astype(np.int64)
    onehot = np.zeros((len(idx), channels + 1))
    onehot[np.arange(len(idx)), idx] = 1.0
    return onehot[:, 1:]


def depth_loss(depth_labels, depth_preds, factor, d_bound):
    """depth_loss.py:54-87.  labels (N, ncam, oriH, oriW), preds (N, ncam, D, H, W) probabilities."""
    preds = np.asarray(depth_preds, np.float64)
    n, ncam, d, h, w = preds.shape
    lab = np.asarray(depth_labels, np.float64).reshape(-1, *np.asarray(depth_labels).shape[2:])
    ori_h, ori_w = lab.shape[1:]
    # F.interpolate(mode="nearest") to (H*f, W*f): src = floor(dst * in / out)
    oh, ow = h * factor, w * factor
    iy = np.minimum((np.arange(oh) * np.float32(ori_h / oh)).astype(np.int64), ori_h - 1)
    ix = np.minimum((np.arange(ow) * np.float32(ori_w / ow)).astype(np.int64), ori_w - 1)
    lab = lab[:, iy][:, :, ix]
    onehot = downsampled_gt_depth(lab, factor, d_bound)
    channels = onehot.shape[1]
    p = preds.reshape(n * ncam, d, h, w).transpose(0, 2, 3, 1).reshape(-1, channels)
    fg = onehot.max(1) > 0
    p, y = p[fg], onehot[fg]
    with np.errstate(divide="ignore"):
        bce = -(y * np.maximum(np.log(p), -100) + (1 - y) * np.maximum(np.log1p(-p), -100))
    return float(bce.sum() / max(1.0, fg.sum()))


def confusion(y_pred, y_true, n_classes):
    """hist[t, p] over voxels with t != 255 (everything SSCMetrics.add_batch accumulates derives from it)."""
    t = np.asarray(y_true).reshape(-1).astype(np.int64)
    p = np.asarray(y_pred).reshape(-1).astype(np.int64)
    keep = t != 255
    return np.bincount(t[keep] * n_classes + p[keep], minlength=n_classes * n_classes).reshape(n_classes, n_classes)


def metrics_from_confusion(hist):
    """sscMetrics.py:70-118 restated on the confusion matrix.
    completion: occupied = label > 0;  semantic: per-class tp/fp/fn over labelled voxels."""
    hist = np.asarray(hist, np.float64)
    tps = np.diag(hist)
    fps = hist.sum(0) - tps
    fns = hist.sum(1) - tps
    c_tp = hist[1:, 1:].sum()
    c_fp = hist[0, 1:].sum()
    c_fn = hist[1:, 0].sum()
    if c_tp != 0:
        precision, recall, iou = c_tp / (c_tp + c_fp), c_tp / (c_tp + c_fn), c_tp / (c_tp + c_fp + c_fn)
    else:
        precision, recall, iou = 0, 0, 0
    iou_ssc = tps / (tps + fps + fns + 1e-5)
    return {"precision": precision, "recall": recall, "iou": iou, "iou_ssc": iou_ssc,
            "iou_ssc_mean": np.mean(iou_ssc[1:]), "tps": tps, "fps": fps, "fns": fns,
            "completion": np.array([c_tp, c_fp, c_fn])}
