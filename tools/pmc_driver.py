#!/usr/bin/env python3
"""Runs each roofline kernel of one bench.py config a few times at that config's shapes (for rocprofv3 --pmc passes; bench.py's
pmc_traffic() and tools/gpu_round.sh pmc start it).  MVS_PMC_CONFIG = 2 (default) | 3 | 4 | 5, MVS_PMC_DTYPE = f32 | bf16."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import mvs_amd  # noqa: F401
from mvs_amd import ops
from mvs_amd import synthetic as R

CFG = int(os.environ.get("MVS_PMC_CONFIG", "2"))
DTYPE = os.environ.get("MVS_PMC_DTYPE", "f32")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
#        C   D    H     W    NS  per-pixel hypotheses
SHAPES = {2: (32, 192, 128, 160, 2, False), 3: (32, 192, 128, 160, 4, False),
          4: (16, 8, 864, 1152, 4, True), 5: (32, 256, 296, 400, 6, False)}
C, D, H, W, NS, PER_PIXEL = SHAPES[CFG]
B = 1
K, E = R.synthetic_cameras(NS + 1, H, W, 4 * W if CFG != 4 else W)
P = E.clone()
P[:, :3, :4] = K @ E[:, :3, :4]
rt = [ops.relative_projection(P[s:s + 1], P[0:1]) for s in range(1, NS + 1)]
rot = torch.stack([r for r, _ in rt], 1).to(dev)
trans = torch.stack([t for _, t in rt], 1).to(dev)
feats = [F.avg_pool2d(torch.randn(B, C, H, W, generator=g), 3, 1, 1).to(dev).contiguous(memory_format=torch.channels_last)
         for _ in range(NS + 1)]
if PER_PIXEL:   # CVP refine level: 8 hypotheses around a smooth per-pixel depth map
    base = 650 + 40 * F.avg_pool2d(torch.randn(B, 1, H, W, generator=g), 31, 1, 15)
    depth = (base + 2.0 * (torch.arange(D).view(1, D, 1, 1) - D // 2)).to(dev).contiguous()
else:
    depth = (425 + 2.65 * torch.arange(D)).unsqueeze(0).to(dev)
train = CFG in (2, 3)
for _ in range(3):
    if train:
        fr = [f.clone().requires_grad_(True) for f in feats]
        var = ops.plane_sweep_variance(fr[0], fr[1:], rot, trans, depth)
        torch.autograd.grad(var, fr, torch.ones_like(var))
    else:
        with torch.no_grad():
            var = ops.plane_sweep_variance(feats[0], feats[1:], rot, trans, depth,
                                           out_dtype=torch.bfloat16 if (CFG == 5 and DTYPE == "bf16") else torch.float32)
    if os.environ.get("MVS_PMC_SWEEP_ONLY"):
        continue
    with torch.no_grad():
        v = var.detach()
        if CFG in (2, 3):
            lib = ops._lib_for(v)
            w0 = (torch.randn(8, 32, 3, 3, 3, generator=g) * 0.05).to(dev)
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
        elif CFG == 4:
            w16 = (torch.randn(16, 16, 3, 3, 3, generator=g) * 0.05).to(dev)
            ops.conv3d_forward(v, w16, 1, False, want_stats=False)
            x64 = torch.randn(1, 64, D // 2, H // 2, W // 2, device=dev).contiguous(memory_format=torch.channels_last_3d)
            w64 = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).to(dev)
            ops.conv3d_forward(x64, w64, 1, False, want_stats=False)
            del x64
        elif CFG == 5 and DTYPE == "bf16":
            w0 = (torch.randn(8, 32, 3, 3, 3, generator=g) * 0.05).to(dev)
            ops.conv3d_forward_bf16(v, w0, 1, False, relu=True)
        elif CFG == 5:
            w0 = (torch.randn(8, 32, 3, 3, 3, generator=g) * 0.05).to(dev)
            ops.conv3d_forward(v, w0, 1, False, want_stats=False)
    del var
torch.cuda.synchronize()
print("pmc driver done (config %d, %s)" % (CFG, DTYPE))
