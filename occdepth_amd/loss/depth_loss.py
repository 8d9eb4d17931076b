"""Depth-distribution loss of FLoSP-Depth, mirror of occdepth/loss/depth_loss.py:7-87."""
import torch
import torch.nn.functional as F


class DepthClsLoss:
    def __init__(self, downsample_factor, d_bound):
        self.downsample_factor = downsample_factor
        self.d_bound = d_bound
        self.depth_channels = int((self.d_bound[1] - self.d_bound[0]) / self.d_bound[2])

    def _get_downsampled_gt_depth(self, gt_depths):
        """(B, N, H, W) metric depth -> (B*N*h*w, D) one-hot of the NEAREST non-zero depth of each
        factor x factor cell (0 = no measurement), bin 0 / out-of-range dropped."""
        f = self.downsample_factor
        B, N, H, W = gt_depths.shape
        g = gt_depths.reshape(B * N, H // f, f, W // f, f)
        g = torch.where(g == 0.0, torch.full_like(g, 1e5), g).amin(dim=(2, 4))
        g = (g - (self.d_bound[0] - self.d_bound[2])) / self.d_bound[2]
        g = torch.where((g < self.depth_channels + 1) & (g >= 0.0), g, torch.zeros_like(g))
        onehot = F.one_hot(g.long(), num_classes=self.depth_channels + 1)
        return onehot.view(-1, self.depth_channels + 1)[:, 1:].float()

    def get_depth_loss(self, depth_labels, depth_preds):
        N_pred, n_cam_pred, D, H, W = depth_preds.shape
        N_gt, n_cam_label, oriH, oriW = depth_labels.shape
        assert N_pred * n_cam_pred == N_gt * n_cam_label, \
            f"N_pred: {N_pred}, n_cam_pred: {n_cam_pred}, N_gt: {N_gt}, n_cam_label: {n_cam_label}"
        f = self.downsample_factor
        labels = F.interpolate(depth_labels.reshape(N_gt * n_cam_label, 1, oriH, oriW), (H * f, W * f), mode="nearest")
        onehot = self._get_downsampled_gt_depth(labels)                                   # (cells, D)
        preds = depth_preds.reshape(N_pred * n_cam_pred, D, H, W).permute(0, 2, 3, 1).reshape(-1, self.depth_channels)
        fg = onehot.amax(1) > 0.0
        # masked sum instead of boolean indexing: same value, no host sync on the number of foreground cells
        with torch.autocast(preds.device.type, enabled=False):      # BCE on probabilities is banned under autocast
            bce = F.binary_cross_entropy(preds.float(), onehot, reduction="none").sum(1)
            return (bce * fg).sum() / torch.clamp(fg.sum().float(), min=1.0)
