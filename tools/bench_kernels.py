#!/usr/bin/env python3
"""Per-kernel timings at BASELINE config-2 shapes (HIP events, one process, interleaved A/B).
    python tools/bench_kernels.py [--reps 20]   -> table on stdout + gpurun_out/kernels.json"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import mvs_amd  # noqa: F401
from mvs_amd import _lib, ops
from mvs_amd import synthetic as R


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
    return ms[len(ms) // 2], ms[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.get()
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
    vox = D * H * W
    rows = []

    def add(name, fn, bound, amount):
        med, mn = timeit(fn, args.reps)
        if bound == "hbm":
            ach, peak, unit = amount / (med * 1e-3) / 1e9, 8000.0, "GB/s"
        else:
            ach, peak, unit = amount / (med * 1e-3) / 1e12, 157.3, "TFLOP/s"
        rows.append({"kernel": name, "ms_median": med, "ms_min": mn, "bound": bound, "achieved": ach, "unit": unit,
                     "frac": ach / peak})
        print("%-34s %8.3f ms (min %7.3f)  %9.1f %-8s %5.1f%% of %s peak" % (name, med, mn, ach, unit, 100 * ach / peak,
                                                                            bound), flush=True)

    k1_bytes = (NS + 1) * C * H * W * 4 + C * vox * 4
    skip_sweep = bool(os.environ.get("MVS_BENCH_SKIP_SWEEP"))
    with torch.no_grad():
        for variant, label in (((3, "cached8"),) if skip_sweep else ((0, "direct"), (2, "cached4"), (3, "cached8"), (4, "cached16"))):
            lib.call("mvs_set_tuning", b"sweep_fwd", variant)
            add("sweep_fwd[%s]" % label, lambda: ops.plane_sweep_variance(feats[0], feats[1:], rot, trans, depth), "hbm", k1_bytes)
        lib.call("mvs_set_tuning", b"dslab", 0)
        lib.call("mvs_set_tuning", b"tile_w", 0)
        lib.call("mvs_set_tuning", b"sweep_fwd", 3)
        var = ops.plane_sweep_variance(feats[0], feats[1:], rot, trans, depth)
        add("calibration: fill_ 503 MB (write only)", lambda: var.fill_(1.0), "hbm", C * vox * 4)
        tmp = torch.empty_like(var)
        add("calibration: copy_ 503 MB (read+write)", lambda: tmp.copy_(var), "hbm", 2 * C * vox * 4)
        del tmp
        var = ops.plane_sweep_variance(feats[0], feats[1:], rot, trans, depth)
    if os.environ.get('MVS_BENCH_SWEEP_ONLY'):
        return
    # backward of the sweep: per-wave windows (default) vs the round-1 view-pair kernel, depth-slab sizes, and N = 5
    def sweep_bwd_case(ns_, label):
        K5, E5 = R.synthetic_cameras(ns_ + 1, H, W, 4 * W)
        P5 = E5.clone()
        P5[:, :3, :4] = K5 @ E5[:, :3, :4]
        rt5 = [ops.relative_projection(P5[s:s + 1], P5[0:1]) for s in range(1, ns_ + 1)]
        rot5 = torch.stack([r for r, _ in rt5], 1).to(dev)
        trans5 = torch.stack([t for _, t in rt5], 1).to(dev)
        f5 = [F.avg_pool2d(torch.randn(B, C, H, W, generator=g), 3, 1, 1).to(dev).contiguous(memory_format=torch.channels_last)
              .requires_grad_(True) for _ in range(ns_ + 1)]
        v5 = ops.plane_sweep_variance(f5[0], f5[1:], rot5, trans5, depth)
        gv5 = torch.randn_like(v5)
        nbytes = C * vox * 4 + 2 * (ns_ + 1) * C * H * W * 4
        for variant, gd, pf, dslab in ((1, 2, 0, 0), (0, 0, 0, 0), (0, 2, 0, 0)):
            lib.call("mvs_set_tuning", b"sweep_bwd", variant)
            lib.call("mvs_set_tuning", b"bwd_dslab", dslab)
            lib.call("mvs_set_tuning", b"bwd_gd", gd)
            lib.call("mvs_set_tuning", b"bwd_pf", pf)
            add("sweep_bwd N=%d [%s%s]%s" % (ns_ + 1, "round-2 per-wave windows, %s%s" % ("gradient 2 planes ahead" if gd == 2 else "rotating gradient set",
                                                                                   ", 1 wave/SIMD (3-4 views)" if pf == 2 else "") if variant == 0
                                             else "round-1 view pairs + LDS atomics", ", dslab %d" % dslab if dslab else "", label),
                lambda: torch.autograd.grad(v5, f5, gv5, retain_graph=True), "hbm", nbytes)
        lib.call("mvs_set_tuning", b"sweep_bwd", _lib.DEFAULT_TUNING.get("sweep_bwd", 0))
        lib.call("mvs_set_tuning", b"bwd_dslab", 0)
        lib.call("mvs_set_tuning", b"bwd_gd", 2)
        lib.call("mvs_set_tuning", b"bwd_pf", 0)

    if not skip_sweep:
        sweep_bwd_case(NS, "")
        sweep_bwd_case(4, "")
    if os.environ.get('MVS_BENCH_BWD_ONLY'):
        return
    # conv0 family
    w0 = (torch.randn(8, 32, 3, 3, 3, generator=g) * 0.05).to(dev)
    fl0 = 2 * 27 * 32 * 8 * vox
    with torch.no_grad():
        y0, _ = ops.conv3d_forward(var, w0, 1, False)
        gy0 = torch.randn_like(y0)
        for k8, xcd, label in ((0, 1, "16x16x4 padded"), (7, 0, "4x4x1 broadcast operand, linear"), (7, 1, "4x4x1 broadcast operand, XCD bricks")):
            lib.call("mvs_set_tuning", b"k8", k8)
            lib.call("mvs_set_tuning", b"xcd", xcd)
            add("conv0 fwd 32>8 [%s]" % label, lambda: ops.conv3d_forward(var, w0, 1, False, want_stats=True), "mfma", fl0)
            add("conv0 wgrad [%s]" % label, lambda: ops.conv3d_wgrad(var, gy0, tuple(w0.shape), 1, False), "mfma", fl0)
        lib.call("mvs_set_tuning", b"k8", _lib.DEFAULT_TUNING["k8"])
        lib.call("mvs_set_tuning", b"xcd", _lib.DEFAULT_TUNING["xcd"])
        add("conv0 dgrad", lambda: ops.conv3d_dgrad(gy0, w0, tuple(var.shape), 1, False), "mfma", fl0)
        # L0 8-channel layers
        w1 = (torch.randn(16, 8, 3, 3, 3, generator=g) * 0.05).to(dev)
        add("conv1 fwd 8>16 s2", lambda: ops.conv3d_forward(y0, w1, 2, False, want_stats=True), "mfma", 2 * 27 * 8 * 16 * vox / 8)
        y1, _ = ops.conv3d_forward(y0, w1, 2, False)
        gy1 = torch.randn_like(y1)
        add("conv1 dgrad (TR2 16>8)", lambda: ops.conv3d_dgrad(gy1, w1, tuple(y0.shape), 2, False), "mfma", 2 * 27 * 8 * 16 * vox / 8)
        # the same with the skip summand and the BatchNorm backward statistics of conv0's block in the epilogue (round 4)
        st0 = torch.stack([y0.mean(dim=(0, 2, 3, 4)), 1.0 / y0.std(dim=(0, 2, 3, 4)), torch.ones(8, device=dev), torch.zeros(8, device=dev)]).contiguous()
        sl0 = torch.zeros((128, 2, 8), dtype=torch.float64, device=dev)
        add("conv1 dgrad + add", lambda: ops.conv3d_dgrad(gy1, w1, tuple(y0.shape), 2, False, add=gy0), "hbm", 3 * 8 * vox * 4)
        add("conv1 dgrad + add + bn stats", lambda: ops.conv3d_dgrad(gy1, w1, tuple(y0.shape), 2, False, add=gy0, bn=(y0, st0, sl0)),
            "hbm", 4 * 8 * vox * 4)
        add("conv1 wgrad", lambda: ops.conv3d_wgrad(y0, gy1, tuple(w1.shape), 2, False), "mfma", 2 * 27 * 8 * 16 * vox / 8)
        wp = (torch.randn(1, 8, 3, 3, 3, generator=g) * 0.05).to(dev)
        bp = torch.zeros(1, device=dev)
        add("prob fwd 8>1", lambda: ops.conv3d_forward(y0, wp, 1, False, shift=bp), "hbm", 9 * vox * 4)
        gp = torch.randn(1, 1, D, H, W, device=dev)
        add("prob dgrad 1>8", lambda: ops.conv3d_dgrad(gp, wp, tuple(y0.shape), 1, False), "hbm", 9 * vox * 4)
        add("prob dgrad 1>8 + bn stats", lambda: ops.conv3d_dgrad(gp, wp, tuple(y0.shape), 1, False, bn=(y0, st0, sl0)), "hbm", 17 * vox * 4)
        add("prob wgrad", lambda: ops.conv3d_wgrad(y0, gp, tuple(wp.shape), 1, False), "hbm", 9 * vox * 4)
        # L1 16>16
        x2 = torch.randn(1, 16, D // 2, H // 2, W // 2, device=dev).contiguous(memory_format=torch.channels_last_3d)
        w2 = (torch.randn(16, 16, 3, 3, 3, generator=g) * 0.05).to(dev)
        fl2 = 2 * 27 * 16 * 16 * vox / 8
        add("conv2 fwd 16>16 @L1", lambda: ops.conv3d_forward(x2, w2, 1, False, want_stats=True), "mfma", fl2)
        add("conv2 wgrad", lambda: ops.conv3d_wgrad(x2, x2, tuple(w2.shape), 1, False), "mfma", fl2)
        # BatchNorm passes on the L0 activation (statistic slots: finished in the apply kernels' prologues)
        gam, bet = torch.ones(8, device=dev), torch.zeros(8, device=dev)
        _, slf = ops.conv3d_forward(var, w0, 1, False, want_stats=True)
        add("bn_relu_fwd_slots L0 (8ch)", lambda: ops.bn_relu_fwd_slots(y0, slf, gam, bet, None, None, 1e-5, 0.1), "hbm", 2 * 8 * vox * 4)
        add("bn_relu_fwd_slots L0 + skip", lambda: ops.bn_relu_fwd_slots(y0, slf, gam, bet, None, None, 1e-5, 0.1, skip=gy0), "hbm", 3 * 8 * vox * 4)
        add("bn bwd reduce + apply L0", lambda: ops.bn_relu_bwd_slots(gy0, y0, st0, torch.zeros_like(sl0), False), "hbm", 5 * 8 * vox * 4)
        add("bn bwd apply L0 (statistics from the dgrad epilogue)", lambda: ops.bn_relu_bwd_slots(gy0, y0, st0, sl0, True), "hbm", 3 * 8 * vox * 4)
        x3 = torch.randn(1, 64, D // 8, H // 8, W // 8, device=dev).contiguous(memory_format=torch.channels_last_3d)
        _, sl3 = ops.conv3d_forward(x3, (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).to(dev), 1, False, want_stats=True)
        g64, b64 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
        add("bn_relu_fwd_slots L3 (64ch, 7.7k voxels)", lambda: ops.bn_relu_fwd_slots(x3, sl3, g64, b64, None, None, 1e-5, 0.1), "hbm", 2 * 64 * vox / 512 * 4)
        # soft-argmin
        lg = torch.randn(1, D, H, W, device=dev) * 3
        add("softargmin_conf fwd", lambda: ops.softargmin_conf(lg, depth), "hbm", vox * 4)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "kernels.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
