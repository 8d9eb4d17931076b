"""Prediction wire formats, mirrors of
  occdepth/scripts/generate_output.py:94-134        -> per-frame pickle {"y_pred": uint16 (X, Y, Z), ...}
  occdepth/scripts/generate_kitti_submission.py:74-85 -> SemanticKITTI `.label` (uint16, learning_map_inv applied)
The class volume comes from the GPU arg-max kernel (`hip.argmax_labels`, first maximum wins like np.argmax; the
reference's softmax before the arg-max does not change it), optionally mapped through the inverse label table in
the same pass; only the final uint16 volume crosses PCIe (4 MB per config-2 frame instead of 168 MB of logits).
"""
import os
import pickle

import numpy as np
import torch

from . import hip

# SemanticKITTI learning_map_inv (data/semantic_kitti/semantic-kitti.yaml:146-166): train id -> dataset label id
KITTI_LEARNING_MAP_INV = (0, 10, 11, 15, 18, 20, 30, 31, 32, 40, 44, 48, 49, 50, 51, 70, 71, 72, 80, 81)


def get_inv_map():
    """io_data.py:99-113."""
    return np.array(KITTI_LEARNING_MAP_INV, dtype=np.int32)


def predict_labels(ssc_logit, inv_map=None):
    """(B, C, X, Y, Z) GPU logits -> (B, X, Y, Z) uint16 numpy volume (arg-max over C, optional label LUT)."""
    vol = hip.argmax_labels(ssc_logit, lut=inv_map)
    return vol.cpu().numpy().astype(np.uint16)


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def output_dict(y_pred_i, batch, i, dataset):
    """The dict generate_output.py pickles for sample i of a batch (same keys per dataset)."""
    out = {"y_pred": np.asarray(y_pred_i, dtype=np.uint16)}
    if "target" in batch:
        out["target"] = _np(batch["target"][i]).astype(np.uint16)
    if dataset == "NYU":
        out["cam_pose"] = _np(batch["cam_pose"][i])
        out["vox_origin"] = _np(batch["vox_origin"][i])
    elif dataset == "tartanair":
        out["vox_origin"] = np.array([-6, -3, 0])
        out["T_velo_2_cam"] = _np(batch["T_velo_2_cam"][i])
        out["fov_mask_1"] = _np(batch["fov_mask_1"][i])
    elif dataset == "kitti":
        out["fov_mask_1"] = _np(batch["fov_mask_1"][i])
        out["cam_k"] = _np(batch["cam_k"][i])
        out["T_velo_2_cam"] = _np(batch["T_velo_2_cam"][i])
    else:
        raise NotImplementedError(dataset)
    return out


def output_path(root, batch, i, dataset):
    if dataset == "NYU":
        return os.path.join(root, batch["name"][i] + ".pkl")
    return os.path.join(root, batch["sequence"][i], batch["frame_id"][i] + ".pkl")


def write_outputs(ssc_logit, batch, root, dataset):
    """generate_output.py's inner loop for one batch; returns the written paths."""
    y_pred = predict_labels(ssc_logit)
    paths = []
    for i in range(y_pred.shape[0]):
        path = output_path(root, batch, i, dataset)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as handle:
            pickle.dump(output_dict(y_pred[i], batch, i, dataset), handle)
        paths.append(path)
    return paths


def write_kitti_submission(ssc_logit, batch, root):
    """generate_kitti_submission.py's inner loop: <root>/sequences/<seq>/predictions/<frame>.label, uint16 labels in
    the dataset's own ids, flattened in (X, Y, Z) order."""
    labels = predict_labels(ssc_logit, inv_map=get_inv_map())
    paths = []
    for i in range(labels.shape[0]):
        d = os.path.join(root, "sequences", batch["sequence"][i], "predictions")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, batch["frame_id"][i] + ".label")
        labels[i].reshape(-1).tofile(path)
        paths.append(path)
    return paths
