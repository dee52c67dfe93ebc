#!/usr/bin/env python3
"""FeatureNet's 2-D layers at BASELINE config 2 (3 views of 512x640 as one batch), per layer and per pass: csrc/conv2d.hip vs the
library (ATen -> MIOpen / CK), HIP events.  python tools/bench_conv2d.py [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import mvs_amd  # noqa: F401
from mvs_amd import ops

LAYERS = (("conv0", 3, 8, 3, 1, 512, 640), ("conv1", 8, 8, 3, 1, 512, 640), ("conv2", 8, 16, 5, 2, 512, 640), ("conv3", 16, 16, 3, 1, 256, 320),
          ("conv4", 16, 16, 3, 1, 256, 320), ("conv5", 16, 32, 5, 2, 256, 320), ("conv6", 32, 32, 3, 1, 128, 160), ("feature", 32, 32, 3, 1, 128, 160))


def timeit(fn, reps, warm=3):
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
    return ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    g = torch.Generator().manual_seed(0)
    bwd = torch.ops.aten.convolution_backward
    print("%-8s %-14s %10s %10s   (ms, median of %d)" % ("layer", "pass", "conv2d.hip", "library", args.reps))
    tot = {}
    keep = []
    for name, cin, cout, ks, st, h, w in LAYERS:
        x = torch.randn(3, cin, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, ks, ks, generator=g) * 0.1).to(dev)
        wcl = wt.contiguous(memory_format=torch.channels_last)
        pad = ks // 2
        y = torch.ops.aten.convolution(x, wcl, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1)
        gy = torch.randn_like(y)
        keep.append((name, x, gy, wcl, st))
        with torch.no_grad():
            rows = [("forward", lambda: ops.conv2d_forward(x, wt, None, st),
                     lambda: torch.ops.aten.convolution(x, wcl, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1))]
            if name != "conv0":
                rows.append(("input grad", lambda: ops.conv2d_dgrad(gy, wt, tuple(x.shape), st),
                             lambda: bwd(gy, x, wcl, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])))
            rows.append(("weight grad", lambda: ops.conv2d_wgrad(x, gy, tuple(wt.shape), st),
                         lambda: bwd(gy, x, wcl, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])))
            for what, ours, lib in rows:
                a, b = timeit(ours, args.reps), timeit(lib, args.reps)
                tot.setdefault(what, [0.0, 0.0])
                tot[what][0] += a
                tot[what][1] += b
                print("%-8s %-14s %10.4f %10.4f   %d>%d k%d s%d %dx%d" % (name, what, a, b, cin, cout, ks, st, h, w), flush=True)
    for what, (a, b) in tot.items():
        print("%-8s %-14s %10.4f %10.4f" % ("TOTAL", what, a, b))
    # all eight weight gradients as ONE launch + one reduction (ops.conv2d_wgrad_batch), whole batch and layer by layer
    xs, gys, ws, sts = [k[1] for k in keep], [k[2] for k in keep], [k[3] for k in keep], [k[4] for k in keep]
    with torch.no_grad():
        for budget in (1024, 1536, 2048, 3072, 4096):
            ops._lib_for(xs[0]).call("mvs_set_tuning", b"wgrad2d_batch", budget)
            ops._WGRAD_BATCH_PLANS.clear()
            print("BATCH    weight grad    %10.4f   (8 layers, one launch + one reduction, %d workgroups)"
                  % (timeit(lambda: ops.conv2d_wgrad_batch(xs, gys, ws, sts), args.reps), budget), flush=True)
        ops._lib_for(xs[0]).call("mvs_set_tuning", b"wgrad2d_batch", 2048)
        ops._WGRAD_BATCH_PLANS.clear()
        for i, k in enumerate(keep):
            print("%-8s batch-of-one   %10.4f" % (k[0], timeit(lambda: ops.conv2d_wgrad_batch(xs[i:i + 1], gys[i:i + 1], ws[i:i + 1], sts[i:i + 1]), args.reps)), flush=True)


if __name__ == "__main__":
    main()
