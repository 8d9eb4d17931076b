"""Run the config-2 lift (K1b) a few times per XCD placement, for rocprofv3 --pmc passes (HBM bytes of the gather kernel):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python tools/pmc_lift.py [fused] [modes...]
`fused`: the table-free kernel (occd_lift_proj_fwd: projection + frustum sample inside) instead of occd_lift_fwd."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occdepth_amd import hip, synthetic
from occdepth_amd.models.SFA import voxel_layout

torch.manual_seed(0)
b = synthetic.kitti_frame(seed=1)
views = [hip.project_voxels(b["T_velo_2_cam_f64"][0][v].numpy(), b["cam_k"][0][v].numpy(), (0.0, -25.6, -2.0), 0.4,
                            (128, 128, 16), 1220, 370, device="cuda") for v in range(2)]
pix = torch.stack([p for p, _ in views]).unsqueeze(0)
fov = torch.stack([m for _, m in views]).unsqueeze(0)
sizes = [(370, 1220), (185, 610), (93, 305), (47, 153)]
rows = [[torch.randn(1, h, w, 64, device="cuda") for _ in range(2)] for h, w in sizes]
n_dims, out_dims, strides = voxel_layout((256, 256, 32), 2, "kitti")
out = hip.Vox.empty(1, out_dims, 64, "cuda")
depth = torch.rand(1, 262144, device="cuda")
fused = "fused" in sys.argv[1:]
modes = [int(m) for m in sys.argv[1:] if m != "fused"] or [0]
if fused:
    from occdepth_amd.models.flosp_depth.flosp_depth import _grid_to_lidar
    cam_E = b["T_velo_2_cam_f64"][0].unsqueeze(0).cuda().contiguous()
    cam_k = b["cam_k"][0].unsqueeze(0).cuda().contiguous()
    dvol = torch.softmax(torch.randn(1, 2, 104, 47, 153, device="cuda"), 2).contiguous()
    g2l = _grid_to_lidar([0, -25.6, -2, 51.2, 25.6, 4.4], (128, 128, 16)).cuda()
    intr = torch.zeros(1, 2, 4, 4, device="cuda")
    intr[:, :, :3, :3] = b["cam_k"][0].float().cuda()
    intr[:, :, 3, 3] = 1
    fr = hip.Frustum(dvol, (b["T_velo_2_cam"][0].cuda().unsqueeze(0) @ g2l).contiguous(), intr[:, :, :3, :].contiguous(),
                     torch.eye(4, device="cuda").repeat(1, 2, 1, 1).contiguous(), (128, 128, 16), (370, 1220), 2.0, 54.0, True)


def launch(m):
    if fused:
        hip.lift_proj(rows, [1, 2, 4, 8], cam_E, cam_k, (0.0, -25.6, -2.0), 0.4, (1220, 370), n_dims, strides, out, frustum=fr,
                      xcd_mode=m)
    else:
        hip.lift(rows, [1, 2, 4, 8], pix, fov, n_dims, strides, out, depth_scale=depth, xcd_mode=m)


for m in modes:
    for _ in range(4):
        launch(m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch(m)
    e1.record()
    torch.cuda.synchronize()
    print(f"xcd_mode {m}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
print("done")
