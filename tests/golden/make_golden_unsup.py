#!/usr/bin/env python3
"""Generate tests/golden/g8_unsup_loss.npz by IMPORTING the reference's UnSupLoss (build container only).

    python tests/golden/make_golden_unsup.py

`jdacs/losses/unsup_loss.py` imports `config.py`, which parses sys.argv at import time; argv is
cleaned first.  Only tensors are stored: seeded inputs (images as float16-exact values), the loss, its three terms,
the gradient w.r.t. the depth map and the first view's warped image + mask."""
import os
import sys
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.ref_torch import synthetic_cameras  # shared synthetic camera definition (inputs only)

REF = "/root/reference"
sys.argv = ["x"]
sys.path.insert(0, os.path.join(REF, "jdacs"))
torch.set_num_threads(4)
from losses.unsup_loss import UnSupLoss  # noqa: E402
from losses.homography import inverse_warping  # noqa: E402


def textured_images(b, n, h, w, g):
    """smooth random textures (so that the photometric terms have a usable gradient), float16-exact"""
    low = torch.randn(b * n, 3, h // 8, w // 8, generator=g)
    img = F.interpolate(low, size=(h, w), mode="bicubic", align_corners=False) + 0.1 * torch.randn(b * n, 3, h, w, generator=g)
    return img.view(b, n, 3, h, w).half().float()


def make(name, b, n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    fh, fw = h // 4, w // 4
    imgs = textured_images(b, n, h, w, g)
    K, E = synthetic_cameras(n, fh, fw, w)
    cams = torch.zeros(b, n, 2, 4, 4)
    cams[:, :, 0] = E
    cams[:, :, 1, :3, :3] = K
    cams[1:, 1:, 0, :3, 3] *= 1.3           # batch items differ
    yy, xx = torch.meshgrid(torch.arange(fh, dtype=torch.float32), torch.arange(fw, dtype=torch.float32), indexing="ij")
    depth = 640.0 + 1.5 * xx - 2.0 * yy + 6.0 * torch.randn(b, fh, fw, generator=g)
    depth = depth.clone().requires_grad_(True)
    crit = UnSupLoss()
    loss = crit(imgs, cams, depth)
    loss.backward()
    with torch.no_grad():
        v1 = F.interpolate(imgs[:, 1], scale_factor=0.25, mode="bilinear").permute(0, 2, 3, 1)
        warped1, mask1 = inverse_warping(v1, cams[:, 0], cams[:, 1], depth.detach())
    out = dict(imgs=imgs.half().numpy(), cams=cams.numpy(), depth=depth.detach().numpy(), loss=loss.detach().numpy(),
               reconstr_loss=crit.reconstr_loss.detach().numpy(), ssim_loss=crit.ssim_loss.detach().numpy(),
               smooth_loss=crit.smooth_loss.detach().numpy(), grad_depth=depth.grad.numpy(),
               warped1=warped1.numpy(), mask1=mask1.numpy())
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-24s %7.1f KB  loss %.6f (reconstr %.6f ssim %.6f smooth %.6f) valid %.2f |grad| %.3e" % (
        name, os.path.getsize(path) / 1024, float(loss), float(crit.reconstr_loss), float(crit.ssim_loss),
        float(crit.smooth_loss), float(mask1.mean()), float(depth.grad.abs().mean())))


make("g8_unsup_loss", 2, 5, 64, 80, 21)
make("g8_unsup_loss_n4", 1, 4, 96, 128, 22)
