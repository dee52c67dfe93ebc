#!/usr/bin/env python3
"""SURVEY 8(f)-1 measurement: UnSupLoss forward+backward at the BASELINE config-3 per-GPU shape (N = 5, 640x512 images,
loss at 160x128), fused HIP kernels vs the reference's op sequence (oracle/ref_torch.py, stock PyTorch-ROCm ops) on the
same GPU.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import mvs_amd  # noqa: F401
from mvs_amd.jdacs.losses.unsup_loss import UnSupLoss
from oracle import ref_torch as R

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(8)
b, n, h, w = 1, 5, 512, 640
imgs = (F.avg_pool2d(torch.randn(b * n, 3, h, w, generator=gen), 9, 1, 4).view(b, n, 3, h, w) * 4).to(dev)
K, E = R.synthetic_cameras(n, h // 4, w // 4, w)
cams = torch.zeros(b, n, 2, 4, 4)
cams[:, :, 0] = E
cams[:, :, 1, :3, :3] = K
cams = cams.to(dev)
depth = (600.0 + 60.0 * torch.rand(b, h // 4, w // 4, generator=gen)).to(dev).requires_grad_(True)
crit = UnSupLoss()


def run(fn, iters=50):
    for _ in range(5):
        depth.grad = None
        fn().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        depth.grad = None
        fn().backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


ms_hip = run(lambda: crit(imgs, cams, depth))
ms_ref = run(lambda: R.unsup_loss(imgs, cams, depth))
print(json.dumps({"what": "UnSupLoss fwd+bwd, B=1 N=5 640x512 (loss at 160x128)", "hip_ms": ms_hip, "reference_ops_ms": ms_ref,
                  "speedup": ms_ref / ms_hip}))

# ---- calDepthHypo at the BASELINE config-4 finest level (1152x864), fused kernel pair vs the oracle's fp64 op sequence
from mvs_amd.jdacs_ms.models import modules as M  # noqa: E402

hh, ww = 864, 1152
K2, E2 = R.synthetic_cameras(3, hh, ww, ww)
ref_in, src_in = K2.unsqueeze(0).to(dev), K2.view(1, 1, 3, 3).repeat(1, 2, 1, 1).to(dev)
ref_ex, src_ex = E2[0].unsqueeze(0).to(dev), E2[1:].unsqueeze(0).to(dev)
dep = (600.0 + 80.0 * torch.rand(1, hh, ww, generator=gen)).to(dev)


def run2(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


ms_a = run2(lambda: M.calDepthHypo(None, dep, ref_in, src_in, ref_ex, src_ex, None, None, 0))
ms_b = run2(lambda: R.cal_depth_hypo(dep, ref_in, src_in, ref_ex, src_ex))
err = float((M.calDepthHypo(None, dep, ref_in, src_in, ref_ex, src_ex, None, None, 0)
             - R.cal_depth_hypo(dep, ref_in, src_in, ref_ex, src_ex)).abs().max())
print(json.dumps({"what": "calDepthHypo, B=1 1152x864 (config 4 finest level)", "hip_ms": ms_a, "reference_ops_ms": ms_b,
                  "speedup": ms_b / ms_a, "max_abs_diff": err}))
