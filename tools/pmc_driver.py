#!/usr/bin/env python3
"""Runs each roofline kernel a few times at config-2 shapes (for rocprofv3 --pmc passes; see tools/gpu_round.sh pmc)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import mvs_amd  # noqa: F401
from mvs_amd import ops
from mvs_amd import synthetic as R

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, D, H, W, NS = 1, 32, 192, 128, 160, 2
K, E = R.synthetic_cameras(NS + 1, H, W, 4 * W)
P = E.clone()
P[:, :3, :4] = K @ E[:, :3, :4]
rt = [ops.relative_projection(P[s:s + 1], P[0:1]) for s in range(1, NS + 1)]
rot = torch.stack([r for r, _ in rt], 1).to(dev)
trans = torch.stack([t for _, t in rt], 1).to(dev)
feats = [F.avg_pool2d(torch.randn(B, C, H, W, generator=g), 3, 1, 1).to(dev).contiguous(memory_format=torch.channels_last)
         for _ in range(NS + 1)]
depth = (425 + 2.65 * torch.arange(D)).unsqueeze(0).to(dev)
w0 = (torch.randn(8, 32, 3, 3, 3, generator=g) * 0.05).to(dev)
for _ in range(3):
    fr = [f.clone().requires_grad_(True) for f in feats]
    var = ops.plane_sweep_variance(fr[0], fr[1:], rot, trans, depth)
    torch.autograd.grad(var, fr, torch.ones_like(var))
    if os.environ.get("MVS_PMC_SWEEP_ONLY"):
        continue
    with torch.no_grad():
        v = var.detach()
        lib = ops._lib_for(v)
        if os.environ.get("MVS_PMC_VARIANTS"):   # A/B of the Cout==8 kernel forms / tile orders (per-dispatch order in the summary)
            for k8, xcd in ((1, 0), (1, 1), (7, 0), (7, 1)):
                lib.call("mvs_set_tuning", b"k8", k8)
                lib.call("mvs_set_tuning", b"xcd", xcd)
                y0, _ = ops.conv3d_forward(v, w0, 1, False, want_stats=True)
                ops.conv3d_wgrad(v, y0, tuple(w0.shape), 1, False)
            lib.call("mvs_set_tuning", b"k8", 7)
            lib.call("mvs_set_tuning", b"xcd", 1)
        y0, _ = ops.conv3d_forward(v, w0, 1, False, want_stats=True)
        ops.conv3d_wgrad(v, y0, tuple(w0.shape), 1, False)
        ops.conv3d_dgrad(y0, w0, tuple(v.shape), 1, False)
torch.cuda.synchronize()
print("pmc driver done")
