"""SSC metrics, mirror of occdepth/loss/sscMetrics.py (SSCMetrics.add_batch / get_stats / reset).

The reference pulls the logits to the host every step (OccDepth.py:523-526: `.cpu().numpy()`, np.argmax) and
counts tp/fp/fn with numpy `where` per class.  Here the arg-max and the counting are one HIP pass (K7,
csrc/loss.hip) into a (C, C) int64 confusion matrix that stays on the GPU; nothing synchronises until
`get_stats()`.  All counters of the reference derive from that matrix:
    tps[j] = hist[j, j]   fps[j] = sum_t hist[t, j] - tps[j]   fns[j] = sum_p hist[j, p] - tps[j]
    completion (occupied = label > 0): tp = hist[1:, 1:].sum(), fp = hist[0, 1:].sum(), fn = hist[1:, 0].sum()
"""
import numpy as np
import torch

from .. import hip


class SSCMetrics:
    def __init__(self, n_classes, device=None):
        self.n_classes = n_classes
        self.device = torch.device(device) if device is not None else None   # None: wherever the first batch lives
        self.reset()

    def reset(self):
        # The matrix is allocated with the first batch (the model is built before .to(device)) and from then on zeroed
        # IN PLACE: a captured training-step hipGraph (train_graph.py) keeps accumulating into this very buffer, so
        # replacing the tensor would leave the replays counting into an orphan and `get_stats()` reading zeros.
        hist = self.__dict__.get("hist")
        if hist is not None:
            hist.zero_()
        else:
            self.hist = None
        self.count = 1e-8

    def _alloc(self, device):
        if self.hist is None:
            self.hist = torch.zeros(self.n_classes, self.n_classes, dtype=torch.int64,
                                    device=self.device if self.device is not None else device)
        return self.hist

    def _u8(self, a, device):
        t = torch.as_tensor(a)
        return t.to(device=device, dtype=torch.uint8).contiguous()

    def add_batch(self, y_pred, y_true, nonempty=None, nonsurface=None):
        """y_pred / y_true: (B, X, Y, Z) class volumes (numpy or torch), 255 = unlabelled in y_true."""
        if nonempty is not None or nonsurface is not None:
            raise NotImplementedError("nonempty / nonsurface masks are not used by the reference's step")
        self.count += 1
        dev = self.device
        if dev is None:
            dev = y_true.device if torch.is_tensor(y_true) and y_true.is_cuda else \
                torch.device("cuda" if torch.cuda.is_available() else "cpu")
        hist = self._alloc(dev)
        hip.ssc_confusion(hist, self._u8(y_true, hist.device), labels=self._u8(y_pred, hist.device))

    def add_batch_logits(self, ssc_logit, y_true):
        """Fused variant of the step's `np.argmax(ssc_pred) -> add_batch`: logits (B, C, X, Y, Z) on the GPU."""
        self.count += 1
        hist = self._alloc(ssc_logit.device)
        hip.ssc_confusion(hist, self._u8(y_true, hist.device), logits=ssc_logit.detach().float())     # (planes or channels-last rows: read in place)

    # -- host-side views (synchronise) ----------------------------------------------------------------------
    def _counts(self):
        if self.hist is None:
            h = np.zeros((self.n_classes, self.n_classes), dtype=np.float64)
        else:
            h = self.hist.cpu().numpy().astype(np.float64)
        tps = np.diag(h).copy()
        return h, tps, h.sum(0) - tps, h.sum(1) - tps

    @property
    def tps(self):
        return self._counts()[1]

    @property
    def fps(self):
        return self._counts()[2]

    @property
    def fns(self):
        return self._counts()[3]

    def get_stats(self):
        h, tps, fps, fns = self._counts()
        c_tp, c_fp, c_fn = h[1:, 1:].sum(), h[0, 1:].sum(), h[1:, 0].sum()
        if c_tp != 0:
            precision = c_tp / (c_tp + c_fp)
            recall = c_tp / (c_tp + c_fn)
            iou = c_tp / (c_tp + c_fp + c_fn)
        else:
            precision, recall, iou = 0, 0, 0
        iou_ssc = tps / (tps + fps + fns + 1e-5)
        return {"precision": precision, "recall": recall, "iou": iou, "iou_ssc": iou_ssc,
                "iou_ssc_mean": np.mean(iou_ssc[1:])}

    def merge_(self, other_hist):
        """Add another rank's confusion matrix (after an all-reduce / gather)."""
        self._alloc(other_hist.device).add_(other_hist.to(self.hist.device))
        return self
