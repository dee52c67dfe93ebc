#!/usr/bin/env python3
"""Generate tests/golden/g3b_ms_homo_warping.npz by IMPORTING the reference's jdacs-ms `homo_warping`
(build container only; jdacs-ms/models/modules.py:62-104).

    python tests/golden/make_golden_ms_warp.py

Two cases (B=2 with different cameras per batch item; a ragged 13x19 map with planes that push samples outside the
image): inputs, warped volume, an upstream gradient and the gradient w.r.t. the source feature map.  Only tensors
are stored.  Separate from make_goldens.py so that its seeded RNG stream (and therefore the other fixtures) stays
bit-identical."""
import os
import sys
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.ref_torch import synthetic_cameras  # shared synthetic camera definition (inputs only)

REF = "/root/reference"
sys.argv = ["x"]
sys.path.insert(0, os.path.join(REF, "jdacs-ms"))
torch.set_num_threads(4)
torch.Tensor.cuda = lambda self, *a, **k: self  # modules.py:73 hard-codes .cuda()
from models import modules as msmod  # noqa: E402

out = {}
for tag, (b, c, d, h, w, seed, d0, dstep) in {"a": (2, 16, 6, 12, 16, 5, 450.0, 25.0),
                                             "b": (1, 16, 7, 13, 19, 6, 300.0, 110.0)}.items():
    g = torch.Generator().manual_seed(seed)
    K, E = synthetic_cameras(3, h, w, 4 * w)
    ref_in = K.unsqueeze(0).repeat(b, 1, 1).clone()
    src_in = K.unsqueeze(0).repeat(b, 1, 1).clone()
    src_in[:, 0, 0] *= 1.03
    ref_ex = E[0].unsqueeze(0).repeat(b, 1, 1).clone()
    src_ex = torch.stack([E[1 + (i % 2)] for i in range(b)], 0).clone()   # a different source pose per batch item
    src = torch.randn(b, c, h, w, generator=g).half().float().requires_grad_(True)
    planes = (d0 + dstep * torch.arange(d, dtype=torch.float32)).unsqueeze(0).repeat(b, 1)
    warped = msmod.homo_warping(src, ref_in, src_in, ref_ex, src_ex, planes)
    gup = torch.randn(warped.shape, generator=g).half().float()   # float16-exact: compresses well
    warped.backward(gup)
    for k, v in dict(src=src, ref_in=ref_in, src_in=src_in, ref_ex=ref_ex, src_ex=src_ex, planes=planes, warped=warped,
                     grad_out=gup, grad_src=src.grad).items():
        out[tag + "_" + k] = v.detach().numpy()
    print(tag, "zero fraction of the warped volume: %.3f" % float((warped == 0).float().mean()))
path = os.path.join(HERE, "g3b_ms_homo_warping.npz")
np.savez_compressed(path, **out)
print("g3b_ms_homo_warping %.1f KB" % (os.path.getsize(path) / 1024))
