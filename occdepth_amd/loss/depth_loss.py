"""Depth-distribution loss of FLoSP-Depth (same class / method names as occdepth/loss/depth_loss.py:7-87).

Ground truth: a sparse metric depth map per camera.  Every `downsample_factor`^2 cell keeps its NEAREST measured
depth (0 = no measurement), which is binned with the LID step of the model's depth axis and one-hot encoded; cells
without a valid bin do not contribute.  Loss: binary cross-entropy between the predicted per-pixel depth
distribution and that one-hot, summed over bins, averaged over the cells that have a measurement.
"""
import torch
import torch.nn.functional as F

from .. import hip

_NO_RETURN = 1e5        # stands in for "no lidar return" while taking the minimum of a cell


def nearest_depth_per_cell(depth, cell):
    """(M, H, W) -> (M, H / cell, W / cell): smallest non-zero depth of each cell x cell block (_NO_RETURN if none)."""
    m, h, w = depth.shape
    blocks = depth.reshape(m, h // cell, cell, w // cell, cell)
    return blocks.masked_fill(blocks == 0.0, _NO_RETURN).amin(dim=(2, 4))


def depth_bin_onehot(depth, d_bound, n_bins):
    """Metric depth -> one-hot over the n_bins depth channels; depths outside (and bin 0) give an all-zero row."""
    lo, _, step = d_bound
    index = (depth - (lo - step)) / step
    index = torch.where((index >= 0.0) & (index < n_bins + 1), index, torch.zeros_like(index)).long()
    return F.one_hot(index, num_classes=n_bins + 1)[..., 1:].reshape(-1, n_bins).float()


class _DepthBCE(torch.autograd.Function):
    """The whole loss as one pass over the predictions (`hip.depth_bce_stats`: nearest resample of the label map, nearest
    depth per cell, LID bin, BCE summed over the bins of the measured cells) and its gradient as one more
    (`hip.depth_bce_grad`) -- csrc/loss.hip; ~25 ATen launches and a (cells, D) one-hot tensor in the reference form."""

    @staticmethod
    def forward(ctx, prob, gt, cell, d_off, d_step):
        st = hip.depth_bce_stats(prob, gt, cell, d_off, d_step).double()
        measured = st[1].clamp(min=1.0)
        ctx.save_for_backward(prob, gt, measured)
        ctx.geom = (cell, d_off, d_step)
        return (st[0] / (hip.REL_Q24 * measured)).float()

    @staticmethod
    def backward(ctx, g):
        prob, gt, measured = ctx.saved_tensors
        gscale = (g.double() / measured).float().reshape(1)
        return hip.depth_bce_grad(prob, gt, *ctx.geom, gscale), None, None, None, None


class DepthClsLoss:
    def __init__(self, downsample_factor, d_bound):
        self.downsample_factor = downsample_factor
        self.d_bound = d_bound
        self.depth_channels = int((d_bound[1] - d_bound[0]) / d_bound[2])

    def _get_downsampled_gt_depth(self, gt_depths):
        """(B, N, H, W) metric depth -> (B*N*h*w, D) one-hot rows."""
        b, n, h, w = gt_depths.shape
        nearest = nearest_depth_per_cell(gt_depths.reshape(b * n, h, w), self.downsample_factor)
        return depth_bin_onehot(nearest, self.d_bound, self.depth_channels)

    def get_depth_loss(self, depth_labels, depth_preds):
        n_pred, cams_pred, bins, h, w = depth_preds.shape
        n_gt, cams_gt, src_h, src_w = depth_labels.shape
        assert n_pred * cams_pred == n_gt * cams_gt, \
            f"N_pred: {n_pred}, n_cam_pred: {cams_pred}, N_gt: {n_gt}, n_cam_label: {cams_gt}"
        cell = self.downsample_factor
        if hip.depth_bce_usable(depth_preds, depth_labels) and bins == self.depth_channels:
            prob = depth_preds.reshape(-1, bins, h, w)
            if prob[0].is_contiguous():
                gt = depth_labels.reshape(-1, src_h, src_w).float().contiguous()
                # (gt - (d0 - d2)) / d2 as the reference evaluates it: the python scalars in double, the tensor in float32
                return _DepthBCE.apply(prob, gt, int(cell), float(self.d_bound[0] - self.d_bound[2]), float(self.d_bound[2]))
        # the label map is first resampled (nearest) to exactly cell x the prediction grid
        labels = F.interpolate(depth_labels.reshape(-1, 1, src_h, src_w), (h * cell, w * cell), mode="nearest")
        target = self._get_downsampled_gt_depth(labels)                                  # (cells, D)
        prob = depth_preds.reshape(-1, bins, h, w).permute(0, 2, 3, 1).reshape(-1, self.depth_channels)
        measured = target.amax(dim=1) > 0.0
        with torch.autocast(prob.device.type, enabled=False):         # BCE on probabilities is banned under autocast
            per_cell = F.binary_cross_entropy(prob.float(), target, reduction="none").sum(dim=1)
            # masked sum instead of boolean indexing: same value, no host sync on the number of measured cells
            return (per_cell * measured).sum() / measured.sum().float().clamp(min=1.0)
