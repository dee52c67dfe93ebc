"""DTU-shaped synthetic inputs: seeded cameras, projection matrices, depth planes and images for the
benchmark, the measurement tools and the tests.  Pure input generation -- no part of the path is computed here.

Cameras: feature-resolution intrinsics f = 0.565 * W_img (DTU training cameras), principal point at the centre; the
reference view is rotated ~1 degree and shifted a few mm; source view k is rotated +-(2..6) degrees about x and y and
translated +-(30..100) mm in x, a third of that in y, a tenth in z, alternating sign -- never a pure x translation with
identity rotation (calDepthHypo's 2x2 system would be singular at y = 0).  Depth planes 425 + 2.65 k mm (dtu_yao.py:290)."""
import math

import torch


def _rot_xy(ax_deg, ay_deg):
    ax, ay = math.radians(ax_deg), math.radians(ay_deg)
    rx = torch.tensor([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    ry = torch.tensor([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
    return (ry @ rx).float()


def synthetic_cameras(nviews, feat_h, feat_w, img_w):
    """-> K [3,3] at feature resolution, E [nviews,4,4]."""
    f = 0.565 * img_w * (feat_w / img_w)
    K = torch.tensor([[f, 0, feat_w / 2.0], [0, f, feat_h / 2.0], [0, 0, 1]], dtype=torch.float32)
    exts = []
    for v in range(nviews):
        E = torch.eye(4)
        if v == 0:
            E[:3, :3] = _rot_xy(1.0, -0.7)
            E[:3, 3] = torch.tensor([5.0, 2.0, 0.5])
        else:
            sgn = 1.0 if v % 2 else -1.0
            mag = 30.0 + 70.0 * ((v * 37) % 100) / 100.0
            E[:3, :3] = _rot_xy(sgn * (2.0 + (v * 1.3) % 4.0), -sgn * (2.0 + (v * 2.1) % 4.0))
            E[:3, 3] = torch.tensor([sgn * mag, -sgn * mag / 3.0, sgn * mag / 10.0])
        exts.append(E)
    return K, torch.stack(exts)


def synthetic_mvsnet_inputs(batch, nviews, img_h, img_w, ndepth, seed=1, depth_min=425.0, interval=2.65):
    """-> imgs [B,N,3,H,W] (randn), proj_matrices [B,N,4,4] (K.[R|t] at feature resolution), depth_values [B,D]."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(batch, nviews, 3, img_h, img_w, generator=g)
    fh, fw = img_h // 4, img_w // 4
    K, E = synthetic_cameras(nviews, fh, fw, img_w)
    proj = E.clone()
    proj[:, :3, :4] = torch.matmul(K, E[:, :3, :4])
    proj = proj.unsqueeze(0).repeat(batch, 1, 1, 1)
    depth_values = (depth_min + interval * torch.arange(ndepth, dtype=torch.float32)).unsqueeze(0).repeat(batch, 1)
    return imgs, proj, depth_values
