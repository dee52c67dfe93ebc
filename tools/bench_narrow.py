#!/usr/bin/env python3
"""The narrow level-0 layers of the regulariser at BASELINE config 2, one at a time (HIP events, nothing beside them), under a list of
knob settings:  python tools/bench_narrow.py "conv_direct=0" "conv_direct=1" "conv_direct=3"
Each line: layer, median / min ms, and the HBM-side floor (bytes the call has to move / 8 TB/s)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import mvs_amd  # noqa: F401
from mvs_amd import _lib, ops


def timeit(fn, reps=30, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in evs)
    return ms[len(ms) // 2], ms[0]


def main():
    dev = torch.device("cuda:0")
    lib = _lib.get()
    settings = sys.argv[1:] or [""]
    g = torch.Generator().manual_seed(0)
    D, H, W = 192, 128, 160
    cl = torch.channels_last_3d
    x8 = torch.randn(1, 8, D, H, W, generator=g).to(dev).contiguous(memory_format=cl)
    x16 = torch.randn(1, 16, D // 2, H // 2, W // 2, generator=g).to(dev).contiguous(memory_format=cl)
    g1 = torch.randn(1, 1, D, H, W, generator=g).to(dev)
    w1 = (torch.randn(16, 8, 3, 3, 3, generator=g) * 0.1).to(dev)        # conv1: 8 -> 16 s2
    w11 = (torch.randn(16, 8, 3, 3, 3, generator=g) * 0.1).to(dev)       # conv11: transposed 16 -> 8 (weight [Cin=16][Cout=8])
    wp = (torch.randn(1, 8, 3, 3, 3, generator=g) * 0.1).to(dev)
    MB = 1e6
    x32 = torch.randn(1, 32, D, H, W, generator=g).to(dev).contiguous(memory_format=cl)
    w0 = (torch.randn(8, 32, 3, 3, 3, generator=g) * 0.1).to(dev)        # conv0: 32 -> 8
    w2 = (torch.randn(16, 16, 3, 3, 3, generator=g) * 0.1).to(dev)       # conv2: 16 -> 16 at level 1
    cases = [
        ("conv0 dgrad (8@L0 -> 32@L0)", lambda: ops.conv3d_dgrad(x8, w0, tuple(x32.shape), 1, False), (126 + 503) * MB),
        ("conv0 wgrad (32@L0 x 8@L0)", lambda: ops.conv3d_wgrad(x32, x8, tuple(w0.shape), 1, False), (126 + 503) * MB),
        ("conv2 fwd 16>16 @L1 (+stats)", lambda: ops.conv3d_forward(x16, w2, 1, False, want_stats=True), 63 * MB),
        ("conv2 dgrad + add", lambda: ops.conv3d_dgrad(x16, w2, tuple(x16.shape), 1, False, add=x16), 94.5 * MB),
        ("conv1 fwd 8>16 s2 (+stats)", lambda: ops.conv3d_forward(x8, w1, 2, False, want_stats=True), (126 + 31.5) * MB),
        ("conv1 dgrad (16@L1 -> 8@L0)", lambda: ops.conv3d_dgrad(x16, w1, tuple(x8.shape), 2, False), (126 + 31.5) * MB),
        ("conv1 dgrad + add", lambda: ops.conv3d_dgrad(x16, w1, tuple(x8.shape), 2, False, add=x8), (252 + 31.5) * MB),
        ("conv1 wgrad", lambda: ops.conv3d_wgrad(x8, x16, tuple(w1.shape), 2, False), (126 + 31.5) * MB),
        ("conv11 fwd T16>8 (+stats)", lambda: ops.conv3d_forward(x16, w11, 2, True, want_stats=True), (126 + 31.5) * MB),
        ("conv11 dgrad (8@L0 -> 16@L1)", lambda: ops.conv3d_dgrad(x8, w11, tuple(x16.shape), 2, True), (126 + 31.5) * MB),
        ("conv11 wgrad", lambda: ops.conv3d_wgrad(x16, x8, tuple(w11.shape), 2, True), (126 + 31.5) * MB),
        ("prob fwd 8>1", lambda: ops.conv3d_forward(x8, wp, 1, False), (126 + 15.7) * MB),
        ("prob dgrad 1>8", lambda: ops.conv3d_dgrad(g1, wp, tuple(x8.shape), 1, False), (126 + 15.7) * MB),
        ("prob wgrad", lambda: ops.conv3d_wgrad(x8, g1, tuple(wp.shape), 1, False), (126 + 15.7) * MB),
    ]
    only = os.environ.get("MVS_NARROW_ONLY", "")
    with torch.no_grad():
        for st in settings:
            pairs = [kv.split("=") for kv in st.split(",") if kv]
            for k, v in pairs:
                lib.call("mvs_set_tuning", k.encode(), int(v))
            print("---- %s" % (st or "defaults"), flush=True)
            for name, fn, nbytes in cases:
                if only and only not in name:
                    continue
                med, mn = timeit(fn)
                print("%-34s %7.3f ms (min %6.3f)   floor %5.3f ms   %4.1f%% of HBM peak" % (name, med, mn, nbytes / 8e12 * 1e3, 100 * nbytes / 8e12 * 1e3 / med), flush=True)
            for k, v in pairs:
                dflt = _lib.DEFAULT_TUNING.get(k)
                if dflt is not None:
                    lib.call("mvs_set_tuning", k.encode(), dflt)


if __name__ == "__main__":
    main()
