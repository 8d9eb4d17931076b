"""OccDepth top module (mirror of occdepth/models/OccDepth.py:30-376, the forward hot path).

Constructor signature, config attributes read, sub-module names (`net_rgb`, `projects`,
`flosp_depth`, `net_3d_decoder`) and the `forward(batch) -> dict` contract are the reference's.
Forward without autograd (eval under torch.no_grad()):
    2-D UNet per view (PyTorch-ROCm / MIOpen + HIP helpers; both views in one batch, optionally replayed as a hipGraph)
 -> K1a FLoSP-Depth frustum sample + K1b fused multi-scale Stereo-SFA lift   (HIP, HBM-bound)
 -> channels-last 3-D UNet + CRP + cascade head on fp32-MFMA implicit GEMM     (HIP, MFMA-bound)
With autograd (training, or eval with gradients enabled) the reference's per-sample / per-scale structure runs on ATen
autograd, with the 3-D convolutions on the HIP forward / dgrad / wgrad kernels (autograd3d.py).  `step` and the
Lightning `*_step` hooks assemble the reference's losses from one statistics pass (loss/ssc_loss.py) and keep the
SSC metrics on the GPU.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..fused import needs_autograd
from ..train_graph import new_graph, seal_graph
from .SFA import SFA, lift_scales, lift_scales_proj
from .flosp_depth import flosp_depth_conf_map
from .flosp_depth.flosp_depth import FlospDepth
from .unet2d import UNet2D
from .unet3d_kitti import UNet3D as UNet3DKitti
from .unet3d_nyu import UNet3D as UNet3DNYU

# hipStreamCaptureModeThreadLocal: only the capturing thread's calls are checked, so a process group's watchdog thread
# (event queries) cannot invalidate -- or abort -- a capture running beside it (N > 1 ranks, OCCDEPTH_FORCE_DIST=1)
CAPTURE_MODE = "thread_local"

try:  # the reference's base class; absent in this image -> plain nn.Module with the hooks it uses
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover - depends on the environment
    class _Base(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class OccDepth(_Base):
    def __init__(self, class_names, class_weights, class_weights_occ=None, full_scene_size=None, project_res=[],
                 config=None, infer_mode=False):
        super().__init__()
        self.project_res = project_res
        self.full_scene_size = full_scene_size
        self.class_names = class_names
        self.class_weights = class_weights
        self.class_weights_occ = class_weights_occ
        for key in ("dataset", "frustum_size", "project_scale", "n_relations", "lr", "weight_decay", "fp_loss",
                    "context_prior", "relation_loss", "CE_ssc_loss", "sem_scal_loss", "geo_scal_loss", "n_classes",
                    "feature", "feature_2d_oc", "trans_2d_to_3d", "cascade_cls", "occluded_cls",
                    "sem_step_decay_loss", "multi_view_mode", "share_2d_backbone_gradient", "use_stereo_depth_gt",
                    "use_lidar_depth_gt", "use_depth_gt"):
            setattr(self, key, getattr(config, key))
        print("INFO: Use cascade cls: {}".format(self.cascade_cls))
        print("INFO: Use occluded cls: {}".format(self.occluded_cls))
        self.infer_mode = infer_mode
        # eval fast path, OFF by default (the default forward is the reference's: per-view 2-D passes, fresh tensors).
        # `enable_fast_eval()` -- or OCCDEPTH_FAST_EVAL=1 in the environment of an unmodified scripts/eval.py /
        # generate_output.py run -- switches all three on; the individual variables switch them one by one.
        env = os.environ.get
        fast = env("OCCDEPTH_FAST_EVAL", "0") == "1"
        self.batch_views = fast or env("OCCDEPTH_BATCH_VIEWS", "0") == "1"   # all views through net_rgb as one batch
        self.graph_2d = self.batch_views and (fast or env("OCCDEPTH_GRAPH_2D", "0") == "1")    # 2-D network as one hipGraph
        self.graph_all = self.batch_views and (fast or env("OCCDEPTH_GRAPH_ALL", "0") == "1")  # WHOLE forward as one hipGraph
        # graph_all replays into static output buffers; by default the caller gets fresh tensors (the reference's contract:
        # a caller may keep `pred` across forwards).  False hands out the static buffers themselves (valid until the next
        # forward of this model) and saves the ~0.1 ms copy of ~300 MB at config 2.
        self.clone_graph_outputs = env("OCCDEPTH_CLONE_OUTPUTS", "1") == "1"
        self._graphs = {}
        self.batch_views_train = os.environ.get("OCCDEPTH_TRAIN_BATCH_VIEWS", "1") == "1"
        # training fast path, OFF by default (the default `training_step` returns the loss and the Trainer runs backward and
        # the optimizer, as in the reference).  `enable_fast_train()` -- or OCCDEPTH_FAST_TRAIN=1 in the environment of an
        # unmodified scripts/train.py run -- switches the module to manual optimisation: `training_step` then replays the
        # WHOLE step (forward, losses, backward, gradient exchange, AdamW) from one hipGraph (train_graph.GraphedTrainStep,
        # the step bench.py --train times).  OCCDEPTH_FAST_TRAIN_BF16=1: the bf16-MFMA convolution mode of configs[3].
        self._fast_train = None
        self._opt = self._sched = None
        self.fast_train = False
        if env("OCCDEPTH_FAST_TRAIN", "0") == "1":
            self.enable_fast_train(bf16=env("OCCDEPTH_FAST_TRAIN_BF16", "0") == "1")
        self.fused_lift = True    # training on the GPU: HIP lift + one-launch backward (lift_autograd.py) where it applies
        if infer_mode:
            self.context_prior = False
        assert not (config.use_stereo_depth_gt and config.use_lidar_depth_gt), "only with one depth data supported."
        self.with_depth_gt = self.use_stereo_depth_gt or self.use_lidar_depth_gt or self.use_depth_gt
        self.depth_loss_w = config.depth_loss_weight

        if self.dataset == "NYU":
            self.net_3d_decoder = UNet3DNYU(self.n_classes, nn.BatchNorm3d, n_relations=self.n_relations,
                                            feature=self.feature, full_scene_size=self.full_scene_size,
                                            context_prior=self.context_prior, cascade_cls=self.cascade_cls,
                                            infer_mode=self.infer_mode)
        elif self.dataset == "kitti":
            self.net_3d_decoder = UNet3DKitti(self.n_classes, nn.BatchNorm3d, project_scale=self.project_scale,
                                              feature=self.feature, full_scene_size=self.full_scene_size,
                                              context_prior=self.context_prior, cascade_cls=self.cascade_cls,
                                              occluded_cls=self.occluded_cls, infer_mode=self.infer_mode)
        self.net_rgb = UNet2D.build(out_feature=self.feature_2d_oc, use_decoder=True,
                                    backbone_2d_name=config.backbone_2d_name,
                                    return_up_feats=config.return_up_feats)
        self.save_hyperparameters()
        from ..loss.sscMetrics import SSCMetrics
        self.train_metrics = SSCMetrics(self.n_classes)      # reference :131-133 (confusion matrices live on the GPU here)
        self.val_metrics = SSCMetrics(self.n_classes)
        self.test_metrics = SSCMetrics(self.n_classes)
        self.metrics_allreduce = False    # opt-in: sum the confusion matrices over ranks before the epoch statistics
        self.init_2d_to_3d_trans(config)
        print("INFO: Use step decay loss: {}".format(self.sem_step_decay_loss))
        batch_size = config.batch_size_per_gpu * config.n_gpus
        if self.dataset == "kitti":
            self.total_batch = (3834 // batch_size) * 30
        elif self.dataset == "NYU":
            self.total_batch = (795 // batch_size) * 30
        else:
            raise NotImplementedError(self.dataset)
        self.cur_batch = 0

    def init_2d_to_3d_trans(self, config):
        print("INFO: Selected 2d->3d transformation method: {}".format(self.trans_2d_to_3d))
        if self.trans_2d_to_3d not in ("flosp", "flosp_depth"):
            raise NotImplementedError(f"{self.trans_2d_to_3d} is not supported yet.")
        self.scale_2ds = [1, 2, 4, 8]
        self.projects = nn.ModuleDict({
            str(s): SFA(config.full_scene_size, project_scale=self.project_scale, dataset=self.dataset)
            for s in self.scale_2ds})
        if self.trans_2d_to_3d == "flosp_depth":
            conf = dict(flosp_depth_conf_map[self.dataset])
            conf.update({
                "scene_size": config.full_scene_size,
                "project_scale": config.project_scale,
                "output_channels": config.feature,
                "depth_net_conf": dict(in_channels=config.feature,
                                       mid_channels=conf["depth_net_conf"]["mid_channels"]),
                "return_depth": self.with_depth_gt,
                "infer_mode": self.infer_mode,
            })
            self.flosp_depth_conf = conf
            self.flosp_depth = FlospDepth(**conf)
            if self.with_depth_gt:
                from ..loss.depth_loss import DepthClsLoss
                self.depth_loss_fn = DepthClsLoss(downsample_factor=conf["downsample_factor"], d_bound=conf["d_bound"])

    # ---------------------------------------------------------------- 2-D side
    def _stamp_of(self, root, slot):
        """Exact fingerprint of every tensor a captured graph bakes pointers / folded copies of: in-place updates
        (optimizer.step, load_state_dict) bump `_version`; replaced storage changes `data_ptr`.  Tuples, not sums (a sum
        can collide); the tensor list is cached until train() / _apply() / load_state_dict() drop the graphs."""
        ts = self.__dict__.get(slot)
        if ts is None:
            ts = list(root.parameters()) + list(root.buffers())
            self.__dict__[slot] = ts
        return (tuple([t._version for t in ts]), tuple([t.data_ptr() for t in ts]))

    def _net_rgb_stamp(self):
        return self._stamp_of(self.net_rgb, "_net_rgb_tensors")

    def _drop_graphs(self):
        self._graphs.clear()
        self.__dict__.pop("_net_rgb_tensors", None)
        self.__dict__.pop("_all_tensors", None)

    def enable_fast_eval(self, clone_outputs=True, graph=True):
        """Switch the eval forward to the benched configuration: both views through the 2-D network as one batch and the
        whole forward replayed from ONE captured hipGraph (`graph=False`: batched views only).  `clone_outputs=True` (default)
        returns fresh tensors per call like the reference; False returns the graph's static output buffers, valid until the
        next forward.  The same switches from the environment: OCCDEPTH_FAST_EVAL=1 (OCCDEPTH_CLONE_OUTPUTS=0)."""
        self.batch_views = True
        self.graph_2d = self.graph_all = bool(graph)
        self.clone_graph_outputs = bool(clone_outputs)
        self._drop_graphs()
        return self

    def enable_fast_train(self, bf16=False, autocast=False):
        """Manual optimisation + the whole training step as ONE replayed hipGraph (see `_fast_training_step`).  Call before
        `trainer.fit` (Lightning reads `automatic_optimization` when the loop starts)."""
        self.fast_train = True
        self.fast_train_bf16, self.fast_train_autocast = bool(bf16), bool(autocast)
        self.automatic_optimization = False
        self._fast_train = None
        return self

    def invalidate_graphs(self):
        """Forget every captured hipGraph (they are re-captured on the next forward / training step)."""
        self._drop_graphs()
        self._fast_train = None

    def train(self, mode=True):
        self._drop_graphs()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._drop_graphs()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._drop_graphs()
        return super().load_state_dict(*a, **k)

    # Round 6 experiment (OCCDEPTH_DEPTHNET_OVERLAP=1): FLoSP-Depth's DepthNet (seven 3x3 convolutions + gate + softmax on the 1/8
    # map: ~0.45 ms of 120-workgroup launches) only needs the 1/8 feature, which the decoder has after its second level; started
    # on a side stream there, it runs beside the 1/4, 1/2 and 1/1 levels (~5 ms of chip-filling launches) instead of after them.
    # Captured into the whole-forward hipGraph as a parallel branch (fork = the decoder's callback, join = in front of the lift).
    depthnet_overlap = os.environ.get("OCCDEPTH_DEPTHNET_OVERLAP", "0") == "1"

    def _depthnet_hook(self, batch, img, bs, n_views):
        """The decoder callback that launches `_depth_volume` early, or None when the conditions of the fused eval lift (the only
        consumer of a deferred frustum) do not hold."""
        if not (self.depthnet_overlap and self.trans_2d_to_3d == "flosp_depth" and self.dataset == "kitti" and img.is_cuda
                and not needs_autograd(self)):
            return None
        if self._lift_calibration(batch, img, "projected_pix_{}".format(self.project_scale)) is None:
            return None
        layer = "1_{}".format(self.flosp_depth_conf["downsample_factor"])
        want = int(self.flosp_depth_conf["downsample_factor"])

        def hook(s, feat):
            if s != want:
                return
            cur = torch.cuda.current_stream(feat.device)
            side = self.__dict__.get("_depth_stream")
            if side is None or side.device != feat.device:
                side = self.__dict__["_depth_stream"] = torch.cuda.Stream(device=feat.device)
            side.wait_stream(cur)                              # fork: everything up to the 1/8 feature head
            feat.record_stream(side)
            with torch.cuda.stream(side):
                views = feat.reshape(bs, n_views, *feat.shape[1:])
                x_rgb = [{layer: views[:, i]} for i in range(n_views)]
                self.__dict__["_early_depth"] = self._depth_volume(batch, x_rgb, None, defer_sample=True)
        return hook

    def _net_rgb_graphed(self, x, on_scale=None):
        """The 2-D network is ~1200 launches of mostly 5-40 us kernels: with `graph_2d` (opt-in, eval only) it is
        captured once per input shape into a hipGraph and replayed, so the host never gates the GPU there.  The 3-D
        stack stays outside (its launches are few, long, and individually timed by bench.py).  A captured graph holds
        pointers to the weights AND to tensors derived from them (folded BatchNorm, packed / Winograd-domain weights);
        every entry therefore carries a stamp of the network's tensors and is re-captured when it no longer matches
        (train(), load_state_dict() and device / dtype moves drop the cache outright)."""
        if not self.graph_2d or not x.is_cuda or torch.cuda.is_current_stream_capturing():
            # (inside the whole-forward capture the network is part of it)
            return self.net_rgb(x) if on_scale is None else self.net_rgb(x, on_scale=on_scale)
        key = (tuple(x.shape), x.device)
        stamp = self._net_rgb_stamp()
        entry = self._graphs.get(key)
        if entry is not None and entry[3] != stamp:
            entry = None
        if entry is None:
            static_in = x.clone()
            try:
                side = torch.cuda.Stream(device=x.device)
                side.wait_stream(torch.cuda.current_stream(x.device))
                with torch.cuda.stream(side):
                    for _ in range(2):                         # warm-up: lazy initialisation, weight caches, MIOpen solvers
                        self.net_rgb(static_in)
                torch.cuda.current_stream(x.device).wait_stream(side)
                graph = new_graph()
                with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
                    static_out = self.net_rgb(static_in)
                seal_graph(graph)
                entry = (graph, static_in, static_out, stamp)
            except (RuntimeError, torch.AcceleratorError) as e:   # capture is an optimisation, never a requirement
                import warnings
                warnings.warn(f"occdepth_amd: hipGraph capture of the 2-D network failed ({e!r}); running eagerly")
                self.graph_2d = False
                self.graph_2d_error = repr(e)
                torch.cuda.synchronize(x.device)
                return self.net_rgb(x)
            self._graphs[key] = entry
        graph, static_in, static_out, _ = entry
        static_in.copy_(x)
        graph.replay()
        return static_out

    def process_rgbs(self, img, batch, n_views):
        bs = img.shape[0]
        if not needs_autograd(self) and self.batch_views and n_views > 1:
            # eval: BN uses running stats, so the views can share one batched pass (opt-in: a different
            # conv batch size changes backend algorithm choice and hence fp32 round-off)
            self.__dict__.pop("_early_depth", None)
            both = self._net_rgb_graphed(img.reshape(bs * n_views, *img.shape[2:]),
                                         on_scale=self._depthnet_hook(batch, img, bs, n_views))
            x_rgb = [{k: v.reshape(bs, n_views, *v.shape[1:])[:, i] for k, v in both.items()}
                     for i in range(n_views)]
        elif self.training and self.batch_views_train and n_views > 1 and not self.share_2d_backbone_gradient:
            # training: both views through the 2-D network as ONE view-major batch with per-view BatchNorm statistics
            # (bn.view_groups) -- the numbers of the reference's per-view loop, half the launches and no gradient
            # accumulation kernels for the shared parameters
            from .. import bn as bn_mod
            stacked = img.transpose(0, 1).reshape(n_views * bs, *img.shape[2:])
            with bn_mod.view_groups(n_views):
                both = self.net_rgb(stacked)
            x_rgb = [{k: v[i * bs:(i + 1) * bs] for k, v in both.items()} for i in range(n_views)]
        else:
            x_rgb = [self.net_rgb(img[:, 0])]
            for i in range(1, n_views):
                if self.share_2d_backbone_gradient:
                    with torch.no_grad():
                        x_rgb.append(self.net_rgb(img[:, i]))
                else:
                    x_rgb.append(self.net_rgb(img[:, i]))
        if n_views == 1 and "gt_depth" in batch:
            bf = batch["virtual_bf"][0].to(device) if "virtual_bf" in batch else None
            x_rgb.append({"1_" + str(s): self.generate_virtual_img(batch, x_rgb[0]["1_" + str(s)], s, bf)
                          for s in self.project_res})
            n_views = 2
        return x_rgb, n_views

    def generate_virtual_img(self, batch, x_single_rgb, scale_2d, bf):
        """Virtual right view: shift the sampling grid by the disparity bf / depth (reference :233-260,
        including its use of sample 0's disparity for the whole batch)."""
        depth = batch["gt_depth"].to(device)
        n, c, h, w = x_single_rgb.shape
        depth_s = F.interpolate(depth, size=(h, w), mode="bilinear", align_corners=False)
        dx = torch.div(bf / int(scale_2d), depth_s).type_as(x_single_rgb)
        dx = torch.where(torch.isinf(dx), torch.zeros_like(dx), dx)
        ys = torch.arange(-1, 1, 2 / h)
        xs = torch.arange(-1, 1, 2 / w)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        grid = torch.stack((gx, gy), dim=2).unsqueeze(0).repeat(n, 1, 1, 1).to(device).type_as(dx)
        grid[..., 0] = grid[..., 0] + (dx * 2 / w)[0]
        return F.grid_sample(x_single_rgb, grid, mode="bilinear", padding_mode="border", align_corners=False)

    # ---------------------------------------------------------------- 2D -> 3D
    def _depth_volume(self, batch, x_rgb, vox_origin, defer_sample=False):
        layer = "1_{}".format(self.flosp_depth_conf["downsample_factor"])
        n_views = 1 if self.dataset == "NYU" else len(x_rgb)
        img_feat = torch.stack([x_rgb[j][layer] for j in range(n_views)], 1).to(device)
        if self.infer_mode:
            kw = {"img_feat": img_feat, "grids": batch["grids"], "scaled_pixel_size": batch["scaled_pixel_size"]}
        else:
            kw = {"img_feat": img_feat, "cam_k": batch["cam_k"], "T_velo_2_cam": batch["T_velo_2_cam"],
                  "ida_mats": batch["ida_mats"], "vox_origin": vox_origin}
        if defer_sample:
            kw["defer_sample"] = True
        if self.with_depth_gt:
            return self.flosp_depth(**kw)
        return self.flosp_depth(**kw), None

    # The eval lift can project the voxels and sample the depth frustum inside the kernel (csrc/lift.hip lift_proj_kernel)
    # from the batch's calibration -- no voxel->pixel tables read.  `lift_in_kernel`:
    #   "auto"  (default) when the batch carries NO `projected_pix_{s}` / `fov_mask_{s}` tables (nothing to contradict: the
    #           dataloader's numba vox2pix pass can be dropped); a batch that brings tables is lifted from ITS tables,
    #           whatever origin / image-space transform they were built with;
    #   True    also when tables are present, provided the batch carries the float64 extrinsics the tables were made from
    #           (`T_velo_2_cam_f64`; float32 extrinsics move a few voxels by one pixel) -- the caller vouches that the tables
    #           are the plain vox2pix of that calibration;
    #   False   never.
    # OCCDEPTH_LIFT_PROJ = auto | 1 | 0.
    lift_in_kernel = {"1": True, "0": False}.get(os.environ.get("OCCDEPTH_LIFT_PROJ", "auto"), "auto")

    def _lift_calibration(self, batch, img, key):
        """(cam_E (B, V, 4, 4), cam_k (B, V, 3, 3)) float64 device tensors for hip.lift_proj, or None when the in-kernel
        projection does not apply (NYU geometry, multi-point patterns, tables present and not vouched for)."""
        if not self.lift_in_kernel or self.dataset != "kitti" or "cam_k" not in batch:
            return None
        have_tables = key in batch
        if have_tables and (self.lift_in_kernel != True or "T_velo_2_cam_f64" not in batch):   # noqa: E712 ("auto" is truthy)
            return None
        ext = batch.get("T_velo_2_cam_f64", batch.get("T_velo_2_cam"))
        if ext is None:
            return None
        if have_tables and batch[key][0].shape[-2] != 1:
            return None                                          # multi-point pattern: table path
        dims = [int(d) // int(self.project_scale) for d in self.full_scene_size]
        if any(v & (v - 1) for v in dims[1:] + [int(r) for r in self.project_res]):
            return None                                          # the fused kernel indexes with shifts
        dev = img.device
        E = torch.stack([e.to(dev) for e in ext]).to(torch.float64)          # (B, V, 4, 4): 1-2 small launches
        k = torch.stack([c.to(dev) for c in batch["cam_k"]]).to(torch.float64)
        if E.shape[:2] != img.shape[:2] or E.shape[1] > 2 or k.shape[:2] != E.shape[:2]:
            return None
        return E.contiguous(), k.contiguous()

    def _forward_2d_to_3d(self, batch, x_rgb, img, bs, vox_origin):
        """eval: returns (Vox, depth_pred); training: ((B, C, X, Y, Z) tensor, depth_pred)."""
        key = "projected_pix_{}".format(self.project_scale)
        mkey = "fov_mask_{}".format(self.project_scale)
        scales = [int(s) for s in self.project_res]
        depth_vol = depth_pred = None
        if not needs_autograd(self):
            cam = self._lift_calibration(batch, img, key)
            if cam is not None:
                frustum = None
                if self.trans_2d_to_3d == "flosp_depth":
                    early = self.__dict__.pop("_early_depth", None)
                    if early is not None:                       # launched beside the decoder's fine levels: join here
                        torch.cuda.current_stream(img.device).wait_stream(self.__dict__["_depth_stream"])
                        frustum, depth_pred = early
                    else:
                        frustum, depth_pred = self._depth_volume(batch, x_rgb, vox_origin, defer_sample=True)
                feats = [[x_rgb[v]["1_" + str(s)] for v in range(len(x_rgb))] for s in scales]
                H, W = img.shape[-2:]
                vox = lift_scales_proj(feats, scales, cam[0], cam[1], self._kitti_origin(batch), 0.2 * self.project_scale, (W, H),
                                       self.projects[str(scales[0])].scene_size, self.project_scale, self.dataset,
                                       frustum=frustum, scale_const=100.0)
                return vox, depth_pred
        if self.trans_2d_to_3d == "flosp_depth":
            depth_vol, depth_pred = self._depth_volume(batch, x_rgb, vox_origin)
        if not needs_autograd(self):
            if key in batch:
                pix = torch.stack([p.to(device) for p in batch[key]])
                fov = torch.stack([m.to(device) for m in batch[mkey]])
            else:
                pix, fov = self.project_voxels_on_gpu(batch, img)
            feats = [[x_rgb[v]["1_" + str(s)] for v in range(len(x_rgb))] for s in scales]
            flat = depth_vol.reshape(bs, -1).contiguous() if depth_vol is not None else None
            vox = lift_scales(feats, scales, pix, fov, self.projects[str(scales[0])].scene_size,
                              self.project_scale, self.dataset, depth_scale=flat, scale_const=100.0)
            return vox, depth_pred
        from .. import lift_autograd
        pix_all = torch.stack([p.to(device) for p in batch[key]])
        fov_all = torch.stack([m.to(device) for m in batch[mkey]])
        feats = [[x_rgb[v]["1_" + str(s)] for v in range(len(x_rgb))] for s in scales]
        if self.fused_lift and lift_autograd.usable(feats, pix_all) and self.dataset == "kitti":
            # training on the GPU, single-point patterns: the fused HIP lift with its one-launch backward
            # (the reference's per-sample / per-scale SFA calls and the `* depth * 100` in one differentiable op)
            x3ds = lift_autograd.lift_scales_autograd(feats, scales, pix_all, fov_all,
                                                      self.projects[str(scales[0])].scene_size, self.project_scale,
                                                      self.dataset, depth_scale=depth_vol, scale_const=100.0)
            return x3ds, depth_pred
        x3ds = []
        for i in range(bs):
            pix, fov = pix_all[i], fov_all[i]
            x3d = None
            for s in scales:
                stack = torch.stack([x_rgb[j]["1_" + str(s)] for j in range(len(x_rgb))], 1).to(device)
                part = self.projects[str(s)](stack[i], torch.div(pix, s, rounding_mode="floor"), fov)
                x3d = part if x3d is None else x3d + part
            x3ds.append(x3d)
        x3ds = torch.stack(x3ds)
        if depth_vol is not None:
            if self.dataset == "NYU":
                depth_vol = depth_vol.permute(0, 1, 2, 4, 3).contiguous()
            x3ds = x3ds * depth_vol * 100
        return x3ds, depth_pred

    def _kitti_origin(self, batch=None):
        """SemanticKITTI voxel origin (kitti_dataset.py:82: vox_origin = (0, -25.6, -2) for the 51.2 m wide scene): the grid
        is centred on the sensor in y, whatever the scene width of a reduced test config.  A batch that carries its own
        `vox_origin` (host or device tensor / array / tuple; the reference's kitti collate does not) overrides it."""
        vo = None if batch is None else batch.get("vox_origin")
        if vo is not None:
            v = vo if torch.is_tensor(vo) else torch.as_tensor(vo[0] if isinstance(vo, (list, tuple)) and torch.is_tensor(vo[0]) else vo)
            if not v.is_cuda:
                return tuple(float(x) for x in v.reshape(-1, 3)[0])
            # A device tensor (Lightning's transfer_batch_to_device moves every batch tensor): the origin is a constant of
            # the dataset, so it is read back ONCE per model (one host sync, outside any graph capture) and the host copy
            # is reused for every later batch that brings a device `vox_origin` -- never silently ignored.
            cached = getattr(self, "_vox_origin_from_device", None)
            if cached is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("batch['vox_origin'] is a device tensor first seen during graph capture; run one eager "
                                       "forward first or pass it as a host tensor")
                cached = tuple(float(x) for x in v.detach().reshape(-1, 3)[0].cpu())
                self._vox_origin_from_device = cached
                default = (0.0, -0.1 * float(self.full_scene_size[1]), -2.0)
                if cached != default:
                    import warnings
                    warnings.warn("occdepth_amd: using the batch's device vox_origin %s (read once, assumed constant over "
                                  "the dataset) instead of the SemanticKITTI default %s" % (cached, default))
            return cached
        return (0.0, -0.1 * float(self.full_scene_size[1]), -2.0)

    def project_voxels_on_gpu(self, batch, img):
        """when the batch carries no `projected_pix_{s}` / `fov_mask_{s}` (the dataloader's
        numba `vox2pix`, kitti_dataset.py:253-273), compute them on the GPU from the calibration
        (SemanticKITTI geometry: vox_origin (0, -25.6, -2), 0.2 m voxels x project_scale, pattern_id 0)."""
        from .. import hip
        if self.dataset != "kitti":
            raise NotImplementedError("on-GPU voxel projection is wired for the SemanticKITTI geometry only")
        ps = self.project_scale
        dims = tuple(int(s) // ps for s in self.full_scene_size)
        H, W = img.shape[-2:]
        pix, fov = [], []
        for i in range(img.shape[0]):
            # the dataloader projects with the calibration file's float64 extrinsics; the batch only carries their
            # float32 copy.  `T_velo_2_cam_f64`, when present, reproduces the dataloader's tables bit for bit.
            ext = batch.get("T_velo_2_cam_f64", batch["T_velo_2_cam"])
            views = [hip.project_voxels(ext[i][v].detach().cpu().double().numpy(),
                                        batch["cam_k"][i][v].detach().cpu().double().numpy(), self._kitti_origin(batch),
                                        0.2 * ps, dims, W, H, device=img.device) for v in range(img.shape[1])]
            pix.append(torch.stack([p for p, _ in views]))
            fov.append(torch.stack([m for _, m in views]))
        return torch.stack(pix), torch.stack(fov)

    # ---------------------------------------------------------------- whole-forward hipGraph
    @staticmethod
    def _batch_signature(batch):
        sig = []
        for k in sorted(batch):
            v = batch[k]
            if torch.is_tensor(v):
                sig.append((k, tuple(v.shape), v.dtype, v.device))
            elif isinstance(v, (list, tuple)) and v and all(torch.is_tensor(t) for t in v):
                sig.append((k, tuple((tuple(t.shape), t.dtype, t.device) for t in v)))
        return tuple(sig)

    @staticmethod
    def _copy_batch(dst, src):
        for k, v in src.items():
            d = dst.get(k)
            if torch.is_tensor(v) and torch.is_tensor(d):
                if d.data_ptr() != v.data_ptr():
                    d.copy_(v, non_blocking=True)
            elif isinstance(v, (list, tuple)) and isinstance(d, list):
                for dd, vv in zip(d, v):
                    if torch.is_tensor(vv) and dd.data_ptr() != vv.data_ptr():
                        dd.copy_(vv, non_blocking=True)

    def _forward_graphed(self, batch):
        """`graph_all` (opt-in, eval under no_grad, GPU batch): the WHOLE forward -- 2-D network, FLoSP-Depth, lift, 3-D
        stack -- captured once per batch signature into ONE hipGraph and replayed per frame: the host only copies the
        frame's tensors into the graph's static inputs (~20 MB at config 2) and launches the replay, so no launch gap is
        left between the 2-D network, the lift and the 3-D stack (rocprof r02: 1.4 ms of a 26.8 ms frame).  The returned
        tensors are the graph's static outputs: valid until the next forward of this model (clone what must survive).
        Like `graph_2d`, an entry is re-captured when any parameter / buffer changed (exact (_version, data_ptr) stamp),
        and a failed capture falls back to the eager path with a warning (`graph_all_error`)."""
        key = ("all", self._batch_signature(batch))
        stamp = self._stamp_of(self, "_all_tensors")
        entry = self._graphs.get(key)
        if entry is not None and entry[3] != stamp:
            entry = None
        if entry is None:
            static = {k: (v.clone() if torch.is_tensor(v) else
                          [t.clone() if torch.is_tensor(t) else t for t in v] if isinstance(v, (list, tuple)) else v)
                      for k, v in batch.items()}
            dev = batch["img"].device
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(2):                         # warm-up: plans, packed weights, library solvers, K2s counters
                        self._forward_impl(static, allow_graph_2d=False)
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                graph = new_graph()
                with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
                    static_out = self._forward_impl(static, allow_graph_2d=False)
                seal_graph(graph)
                entry = (graph, static, static_out, stamp)
            except (RuntimeError, torch.AcceleratorError) as e:   # capture is an optimisation, never a requirement
                import warnings
                warnings.warn(f"occdepth_amd: hipGraph capture of the whole forward failed ({e!r}); running eagerly")
                self.graph_all = False
                self.graph_all_error = repr(e)
                torch.cuda.synchronize(dev)
                return self._forward_impl(batch)
            self._graphs[key] = entry
        graph, static, static_out, _ = entry
        self._copy_batch(static, batch)
        graph.replay()
        if self.clone_graph_outputs:
            return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in static_out.items()}
        return dict(static_out)

    def forward(self, batch):
        if self.graph_all and not needs_autograd(self) and batch["img"].is_cuda \
                and not torch.cuda.is_current_stream_capturing():
            return self._forward_graphed(batch)
        return self._forward_impl(batch)

    def _forward_impl(self, batch, allow_graph_2d=True):
        if allow_graph_2d or not self.graph_2d:
            return self._forward_body(batch)
        self.graph_2d = False                  # (the whole-forward capture contains the 2-D network: no nested replay)
        try:
            return self._forward_body(batch)
        finally:
            self.graph_2d = True

    def _forward_body(self, batch):
        img = batch["img"].to(device)
        bs, n_views = img.shape[:2]
        x_rgb, n_views = self.process_rgbs(img, batch, n_views)
        if self.dataset in ("NYU", "tartanair"):
            vox_origin = batch["vox_origin"]
        elif self.dataset == "kitti":
            vox_origin = None
        else:
            raise NotImplementedError("dataset is not supported: {}".format(self.dataset))
        x3ds, depth_pred = self._forward_2d_to_3d(batch, x_rgb, img, bs, vox_origin)
        out = dict(self.net_3d_decoder({"x3d": x3ds}))
        if self.with_depth_gt and self.trans_2d_to_3d == "flosp_depth":
            out["depth_pred"] = depth_pred
        return out

    # ---------------------------------------------------------------- Lightning hooks
    def _log(self, key, value):
        self.logged[key] = value.detach()
        # inside the fast training step (its capture, its warm-up steps on a snapshot, its eager fall-back) the Trainer's
        # logger is fed AFTER the step from `self.logged` (_fast_training_step): a logger call inside a capture would be
        # baked into the graph -- or synchronise the host, which a capture forbids
        if not self.__dict__.get("_in_fast_step", False):
            self.log(key, value.detach(), on_epoch=True, sync_dist=True)

    def step(self, batch, step_type, metric):
        """Loss assembly of occdepth/models/OccDepth.py:378-533.  Same terms, same switches, same log keys; the
        scene-completion terms (CE, sem_scal, geo_scal, frustum proportion) come from one statistics pass over
        `ssc_logit` (loss/ssc_loss.py), and the metric update stays on the GPU (no `.cpu().numpy()` per step)."""
        from ..loss import ssc_loss
        from ..loss.CRP_loss import compute_super_CP_multilabel_loss
        self.logged = {}
        out_dict = self(batch)
        ssc_pred = out_dict["ssc_logit"]
        dev = ssc_pred.device
        target = batch["target"].to(dev)
        loss = 0
        if self.context_prior and self.relation_loss:
            loss_rel_ce = compute_super_CP_multilabel_loss(out_dict["P_logits"], batch["CP_mega_matrices"])
            loss = loss + loss_rel_ce
            self._log(step_type + "/loss_relation_ce_super", loss_rel_ce)

        use_fp = self.fp_loss and step_type != "test"
        masks = dists = None
        if use_fp:
            masks = torch.stack(list(batch["frustums_masks"])).to(dev)
            dists = torch.stack(list(batch["frustums_class_dists"])).float().to(dev)
        terms = ssc_loss.ssc_losses(ssc_pred, target, self._on_device("class_weights", dev), masks, dists,
                                    ce=self.CE_ssc_loss, sem_scal=self.sem_scal_loss, geo_scal=self.geo_scal_loss)
        if self.CE_ssc_loss:
            loss = loss + terms["loss_ssc"]
            self._log(step_type + "/loss_ssc", terms["loss_ssc"])
            if self.cascade_cls:
                loss_occ = ssc_loss.occ_ce_loss(out_dict["occ_logit"], target, self._on_device("class_weights_occ", dev))
                loss = loss + loss_occ
                self._log(step_type + "/loss_occ", loss_occ)
            if self.occluded_cls and "occluded" in batch:
                loss_occluded = ssc_loss.CE_ssc_loss(out_dict["occluded_logit"], batch["occluded"].to(dev),
                                                     torch.ones(2, device=dev))
                loss = loss + loss_occluded
                self._log(step_type + "/loss_occluded", loss_occluded)

        if self.with_depth_gt and self.trans_2d_to_3d == "flosp_depth" and "gt_depth" in batch:
            if self.use_stereo_depth_gt:
                depth_pred = out_dict["depth_pred"][:, 0].unsqueeze(1)        # only the left camera has ground truth
            elif self.use_lidar_depth_gt or self.use_depth_gt:
                depth_pred = out_dict["depth_pred"]
            else:
                raise NotImplementedError("Only stereo depth gt supported.")
            loss_depth = self.depth_loss_fn.get_depth_loss(batch["gt_depth"].to(dev), depth_pred) * self.depth_loss_w
            loss = loss + loss_depth
            self._log(step_type + "/loss_depth", loss_depth)

        if self.sem_scal_loss:
            decay = max(0.1, (1 - self.cur_batch / self.total_batch)) if self.sem_step_decay_loss else 1.0
            if self.sem_step_decay_loss and getattr(self, "_decay_dev", None) is not None and ssc_pred.is_cuda:
                # device scalar a captured step reads (train_graph.GraphedTrainStep refreshes it before every replay).  An
                # EAGER step -- a failed capture's fall-back, or a plain training_step once the scalar exists -- refreshes it
                # here from the host formula (asynchronous fill, no sync), so the factor can never freeze at its capture value
                if not torch.cuda.is_current_stream_capturing():
                    self._decay_dev.fill_(decay)
                decay = self._decay_dev
            loss_sem_scal = terms["loss_sem_scal"] * decay
            loss = loss + loss_sem_scal
            self._log(step_type + "/loss_sem_scal", loss_sem_scal)
        if self.geo_scal_loss:
            loss = loss + terms["loss_geo_scal"]
            self._log(step_type + "/loss_geo_scal", terms["loss_geo_scal"])
        if use_fp:
            loss = loss + terms["loss_frustums"]
            self._log(step_type + "/loss_frustums", terms["loss_frustums"])

        if metric is not None:
            if hasattr(metric, "add_batch_logits"):
                metric.add_batch_logits(ssc_pred, target)
            else:                                        # a reference-style (numpy) metric object
                metric.add_batch(ssc_pred.detach().argmax(1).cpu().numpy(), target.cpu().numpy())
        self._log(step_type + "/loss", loss)
        return loss

    def _on_device(self, name, dev):
        """float32 device copy of a host-side attribute tensor (class weights), cached: `.to(dev)` per step is a host
        sync on the loss path."""
        src = getattr(self, name)
        cache = self.__dict__.setdefault("_dev_attr", {})
        hit = cache.get((name, str(dev)))
        if hit is None or hit[0] is not src:
            hit = (src, src.to(dev).float())
            cache[(name, str(dev))] = hit
        return hit[1]

    def training_step(self, batch, batch_idx):
        if self.fast_train and not self.__dict__.get("_in_fast_step", False):
            return self._fast_training_step(batch, batch_idx)
        self.cur_batch += 1
        return self.step(batch, "train", self.train_metrics)

    # ---- OCCDEPTH_FAST_TRAIN: the benched training step behind the Trainer's `training_step`
    @staticmethod
    def _batch_signature(batch):
        sig = []
        for k in sorted(batch):
            v = batch[k]
            if torch.is_tensor(v):
                sig.append((k, tuple(v.shape), v.dtype))
            elif isinstance(v, (list, tuple)) and v and all(torch.is_tensor(t) for t in v):
                sig.append((k, tuple((tuple(t.shape), t.dtype) for t in v)))
            else:
                sig.append((k, repr(v) if isinstance(v, (int, float, str, bool, type(None))) else type(v).__name__))
        return tuple(sig)

    def _fast_optimizer(self):
        """The raw torch optimizer of this module: the one `configure_optimizers` built (Lightning calls it before the first
        step), else what the Trainer reports through `self.optimizers()`."""
        if self._opt is not None:
            return self._opt
        opts = self.optimizers()
        opt = opts[0] if isinstance(opts, (list, tuple)) else opts
        return getattr(opt, "optimizer", opt)

    def _fast_training_step(self, batch, batch_idx):
        """`training_step` under manual optimisation (`automatic_optimization = False`: Lightning 1.4.9 then calls neither
        backward nor optimizer.step, models/OccDepth.py:535-541 + scripts/train.py:208 otherwise drive them eagerly).
        First call: the batch becomes the STATIC batch of a `GraphedTrainStep` (device copies), the step is captured (its
        warm-up steps run on a snapshot: capturing trains nothing) and replayed; later calls copy the new batch into the
        static tensors and replay.  Several ranks: the module is wrapped in DDP by the Trainer, whose reducer only acts on a
        backward that follows the forward -- here the backward is INSIDE the step, so the gradient average is taken by
        `shard.GradBuckets` (captured with the step) and DDP's hooks stay idle.  What the replay cannot do is done here:
        `self.log` of the step's loss terms (device scalars the graph writes), the host counters (GraphedTrainStep.__call__).
        Falls back to the eager manual step with a warning when the capture fails, on the CPU, or for a batch whose
        keys / shapes differ from the captured one (that batch alone runs eagerly)."""
        import warnings
        from .. import autograd3d, shard, train_graph
        on_gpu = next(self.parameters()).is_cuda
        opt = self._fast_optimizer()
        st = self._fast_train
        if st is None:
            st = self._fast_train = {"graph": None, "sig": None, "buckets": None, "warned": False, "logged": None}
            if on_gpu:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    st["buckets"] = shard.GradBuckets(self.parameters(), dist)
                dev = next(self.parameters()).device
                static = {k: ([t.to(dev).clone() if torch.is_tensor(t) else t for t in v] if isinstance(v, (list, tuple))
                              else (v.to(dev).clone() if torch.is_tensor(v) else v)) for k, v in batch.items()}
                if self.fast_train_bf16:
                    autograd3d.set_bf16_mfma(True)
                gs = train_graph.GraphedTrainStep(self, opt, static, bf16=self.fast_train_autocast, buckets=st["buckets"],
                                                  warmup=2, batch_idx=batch_idx)
                self.__dict__["_in_fast_step"] = True
                try:
                    ok = gs.capture()
                finally:
                    self.__dict__["_in_fast_step"] = False
                st["graph"], st["sig"] = gs, self._batch_signature(static)
                st["logged"] = dict(self.logged) if ok else None     # the device scalars the captured step writes
                if not ok:
                    warnings.warn(f"occdepth_amd: the training step could not be captured ({gs.error}); running it eagerly")
            else:
                warnings.warn("occdepth_amd: OCCDEPTH_FAST_TRAIN needs the model on the GPU; running the eager manual step")
        gs = st["graph"]
        self.__dict__["_in_fast_step"] = True
        try:
            if gs is not None and self._batch_signature(batch) == st["sig"]:
                gs.load_batch(batch)
                loss = gs()                                  # replay (or the eager step on the static batch after a failed capture)
                if gs.graph is not None:
                    self.logged = dict(st["logged"])
            else:
                if gs is not None and not st["warned"]:
                    st["warned"] = True
                    warnings.warn("occdepth_amd: a training batch differs in keys / shapes from the captured one; such "
                                  "batches run the eager step")
                loss = self._manual_eager_step(batch, batch_idx, opt, st["buckets"])
        finally:
            self.__dict__["_in_fast_step"] = False
        for k, v in self.logged.items():                    # the replay does not run Python: log what the graph wrote
            self.log(k, v, on_epoch=True, sync_dist=True)
        return loss.detach()

    def _manual_eager_step(self, batch, batch_idx, opt, buckets=None):
        """One manual-optimisation step without a graph: zero_grad, training_step, backward, [gradient average], optimizer."""
        if buckets is not None:
            buckets.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(self.fast_train_autocast and next(self.parameters()).is_cuda)):
            loss = self.training_step(batch, batch_idx)
        loss.backward()
        if buckets is not None:
            buckets.finish()
        opt.step()
        return loss

    def on_train_epoch_start(self, *args, **kwargs):
        """Manual optimisation: Lightning 1.4 leaves the LR schedule to the module.  MultiStepLR counts epochs: bring it to
        the current epoch (idempotent -- a Trainer that did step it finds nothing to do)."""
        if not self.fast_train or self._sched is None:
            return
        epoch = int(getattr(self, "current_epoch", 0) or 0)
        while self._sched.last_epoch < epoch:
            self._sched.step()

    def on_train_batch_end(self, *args, **kwargs):
        """Once per step (Lightning calls it after backward + optimizer step): a peer-memory SyncBatchNorm exchange that
        gave up waiting for a rank raises here instead of training on -- its results are already NaN (shard.SmallAllReduce).
        No host synchronisation: the device flag is copied asynchronously and examined one step later."""
        from .. import shard
        shard.poll_exchanges()

    def validation_step(self, batch, batch_idx):
        self.step(batch, "val", self.val_metrics)

    def _epoch_stats(self, metric):
        """`SSCMetrics.get_stats()` of this rank's counts -- the reference's semantics: each rank logs the statistics of
        its own confusion matrix and Lightning averages them (`sync_dist=True`).  With `metrics_allreduce` the
        confusion matrices are summed over the ranks first (the statistically exact variant)."""
        if self.metrics_allreduce:
            from .. import shard
            shard.allreduce_confusion(metric)
        return metric.get_stats()

    def validation_epoch_end(self, outputs):
        """Reference :542-557: per-class IoU, mIoU, IoU, Precision, Recall of the train and val metrics, then reset."""
        for prefix, metric in (("train", self.train_metrics), ("val", self.val_metrics)):
            stats = self._epoch_stats(metric)
            for i, class_name in enumerate(self.class_names):
                self.log("{}_SemIoU/{}".format(prefix, class_name), stats["iou_ssc"][i], sync_dist=True)
            self.log("{}/mIoU".format(prefix), stats["iou_ssc_mean"], sync_dist=True)
            self.log("{}/IoU".format(prefix), stats["iou"], sync_dist=True)
            self.log("{}/Precision".format(prefix), stats["precision"], sync_dist=True)
            self.log("{}/Recall".format(prefix), stats["recall"], sync_dist=True)
            metric.reset()

    def test_step(self, batch, batch_idx):
        self.step(batch, "test", self.test_metrics)

    def test_epoch_end(self, outputs):
        """Reference :562-580: the printed evaluation report of scripts/eval.py."""
        classes = self.class_names
        for prefix, metric in (("test", self.test_metrics),):
            print("{}======".format(prefix))
            stats = self._epoch_stats(metric)
            print("Precision={:.4f}, Recall={:.4f}, IoU={:.4f}".format(
                stats["precision"] * 100, stats["recall"] * 100, stats["iou"] * 100))
            print("class IoU: {}, ".format(classes))
            print(" ".join(["{:.4f}, "] * len(classes)).format(*(stats["iou_ssc"] * 100).tolist()))
            print("mIoU={:.4f}".format(stats["iou_ssc_mean"] * 100))
            metric.reset()

    def configure_optimizers(self):
        """Reference :582-600: AdamW + MultiStepLR, schedule by dataset."""
        from torch.optim.lr_scheduler import MultiStepLR
        if self.dataset in ("NYU", "kitti"):
            milestones, gamma = [18, 24], 0.4
        elif self.dataset == "tartanair":
            milestones, gamma = [20], 0.1
        else:
            raise NotImplementedError("dataset is not supported: {}".format(self.dataset))
        params = list(self.parameters())
        opt = None
        if params and all(p.is_cuda and p.is_floating_point() for p in params):
            try:        # the fused (multi-tensor, single-launch) implementation of the same AdamW update
                opt = torch.optim.AdamW(params, lr=self.lr, weight_decay=self.weight_decay, fused=True)
            except (RuntimeError, ValueError):
                opt = None
        if opt is None:
            opt = torch.optim.AdamW(params, lr=self.lr, weight_decay=self.weight_decay)
        self._opt = opt
        if self.fast_train and params and params[0].is_cuda:
            from .. import train_graph
            train_graph.make_capturable(opt)            # device-side step counter and learning rate, before the first step
        self._sched = MultiStepLR(opt, milestones=milestones, gamma=gamma)
        return [opt], [self._sched]
