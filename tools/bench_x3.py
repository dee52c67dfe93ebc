#!/usr/bin/env python3
"""conv0's input gradient (8 -> 32 channels) at BASELINE config 2: the fp32-MFMA kernel against the opt-in split-bf16 kernel
(knob conv0_x3, csrc/conv3d_x3.hip): time (HIP events, nothing beside them) and error against an fp64 reference of the same op."""
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
    g = torch.Generator().manual_seed(0)
    cl = torch.channels_last_3d
    w0 = (torch.randn(8, 32, 3, 3, 3, generator=g) * 0.1).to(dev)
    # error: a volume fp64 can do (48 x 64 x 80), gradients with a wide dynamic range
    D, H, W = 48, 64, 80
    gy = (torch.randn(1, 8, D, H, W, generator=g) * torch.rand(1, 8, D, H, W, generator=g).pow(4) * 10).to(dev).contiguous(memory_format=cl)
    ref = torch.nn.grad.conv3d_input((1, 32, D, H, W), w0.double(), gy.double(), padding=1)
    for knob in (0, 1):
        lib.call("mvs_set_tuning", b"conv0_x3", knob)
        out = ops.conv3d_dgrad(gy, w0, (1, 32, D, H, W), 1, False)
        e = (out.double() - ref).abs()
        print("conv0_x3=%d  error vs fp64: max abs %.3e (scale %.1f), relative L1 %.3e, rms %.3e" % (
            knob, e.max().item(), ref.abs().max().item(), (e.sum() / ref.abs().sum()).item(), e.pow(2).mean().sqrt().item()))
    # time: config 2
    D, H, W = 192, 128, 160
    x8 = torch.randn(1, 8, D, H, W, generator=g).to(dev).contiguous(memory_format=cl)
    for knob in (0, 1, 0, 1):
        lib.call("mvs_set_tuning", b"conv0_x3", knob)
        med, mn = timeit(lambda: ops.conv3d_dgrad(x8, w0, (1, 32, D, H, W), 1, False))
        print("conv0_x3=%d  conv0 dgrad 1x192x128x160: median %.4f ms  min %.4f ms" % (knob, med, mn))
    # ---- forward (knob bit 1) ----
    import torch.nn.functional as F
    D, H, W = 48, 64, 80
    xs = (torch.randn(1, 32, D, H, W, generator=g) * torch.rand(1, 32, D, H, W, generator=g).pow(4) * 10).to(dev).contiguous(memory_format=cl)
    ref = F.conv3d(xs.double().cpu(), w0.double().cpu(), padding=1).to(dev)
    for knob in (0, 2):
        lib.call("mvs_set_tuning", b"conv0_x3", knob)
        out, slots = ops.conv3d_forward(xs, w0, 1, False, want_stats=True)
        e = (out.double() - ref).abs()
        st = slots.sum(0)
        print("conv0_x3=%d  forward error vs fp64: max abs %.3e (scale %.1f), relative L1 %.3e; statistics: sum %.2e, sum of squares %.2e (relative)" % (
            knob, e.max().item(), ref.abs().max().item(), (e.sum() / ref.abs().sum()).item(),
            ((st[0] - ref.sum((0, 2, 3, 4))).abs() / ref.abs().sum((0, 2, 3, 4))).max().item(),
            ((st[1] - ref.pow(2).sum((0, 2, 3, 4))).abs() / ref.pow(2).sum((0, 2, 3, 4))).max().item()))
    D, H, W = 192, 128, 160
    x32 = torch.randn(1, 32, D, H, W, generator=g).to(dev).contiguous(memory_format=cl)
    for knob in (0, 2, 0, 2):
        lib.call("mvs_set_tuning", b"conv0_x3", knob)
        med, mn = timeit(lambda: ops.conv3d_forward(x32, w0, 1, False, want_stats=True))
        print("conv0_x3=%d  conv0 forward (+statistics) 1x192x128x160: median %.4f ms  min %.4f ms" % (knob, med, mn))
    lib.call("mvs_set_tuning", b"conv0_x3", 1)
    x8s = x8[:, :, :4].contiguous(memory_format=cl)
    med, mn = timeit(lambda: ops.conv3d_dgrad(x8s, w0, (1, 32, 4, H, W), 1, False))
    print("conv0_x3=1  one tile per workgroup (1x4x128x160: prologue + 1 tile): median %.4f ms" % med)
    lib.call("mvs_set_tuning", b"conv0_x3", 0)


if __name__ == "__main__":
    main()
