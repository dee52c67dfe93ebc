#!/usr/bin/env python3
"""K1 (warp + variance forward) alone at BASELINE config-2 / config-3 shapes under its launch knobs: python tools/bench_k1.py [NS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import mvs_amd  # noqa: F401
from mvs_amd import _lib, ops
from mvs_amd import synthetic as R

def timeit(fn, reps=40, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in evs)
    return ms[len(ms) // 2], ms[0]

def main():
    NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    lib = _lib.get()
    g = torch.Generator().manual_seed(0)
    B, C, D, H, W = 1, 32, 192, 128, 160
    K, E = R.synthetic_cameras(NS + 1, H, W, 4 * W)
    P = E.clone(); P[:, :3, :4] = K @ E[:, :3, :4]
    rt = [ops.relative_projection(P[s:s + 1], P[0:1]) for s in range(1, NS + 1)]
    rot = torch.stack([r for r, _ in rt], 1).to(dev); trans = torch.stack([t for _, t in rt], 1).to(dev)
    feats = [F.avg_pool2d(torch.randn(B, C, H, W, generator=g), 3, 1, 1).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(NS + 1)]
    depth = (425 + 2.65 * torch.arange(D)).unsqueeze(0).to(dev)
    nbytes = (NS + 1) * C * H * W * 4 + C * D * H * W * 4
    run = lambda: ops.plane_sweep_variance(feats[0], feats[1:], rot, trans, depth)
    if os.environ.get("MVS_K1_INTERLEAVED"):
        variants = [("default", {}), ("round 5: tile_w=8,fwd_dl=1,dslab=12", {"tile_w": 8, "fwd_dl": 1, "dslab": 12}), ("tile_w=8", {"tile_w": 8}), ("fwd_dl=1", {"fwd_dl": 1}),
                    ("dslab=12", {"dslab": 12}), ("dslab=24", {"dslab": 24})]
        acc = {n: [] for n, _ in variants}
        with torch.no_grad():
            timeit(run, 20)
            for rnd in range(6):
                for name, knobs in variants:
                    for k, v in knobs.items():
                        lib.call("mvs_set_tuning", k.encode(), v)
                    acc[name].append(timeit(run, 60, 3)[0])
                    for k in knobs:
                        lib.call("mvs_set_tuning", k.encode(), {"nt": 0, "dslab": 0, "tile_w": 0, "fwd_dl": 2}[k])
        for name, _ in variants:
            v = sorted(acc[name])
            print("%-28s median of 6 medians %.4f ms (%.4f .. %.4f) = %.3f of 8 TB/s" % (name, v[len(v) // 2], v[0], v[-1], nbytes / v[len(v) // 2] / 1e6 / 8000), flush=True)
        return
    cases = [("default", {})] + [("nt=1", {"nt": 1})] + [("dslab=%d" % d, {"dslab": d}) for d in (8, 12, 16, 24, 32, 48)] + \
            [("tile_w=%d" % t, {"tile_w": t}) for t in (4, 8, 16, 32)] + [("fwd_dl=%d" % d, {"fwd_dl": d}) for d in (0, 2)] + [("default again", {})]
    with torch.no_grad():
        for name, knobs in cases:
            for k, v in knobs.items():
                lib.call("mvs_set_tuning", k.encode(), v)
            med, mn = timeit(run)
            print("%-16s %.4f ms (min %.4f)  %.1f GB/s = %.3f of 8 TB/s" % (name, med, mn, nbytes / med / 1e6, nbytes / med / 1e6 / 8000), flush=True)
            for k in knobs:
                lib.call("mvs_set_tuning", k.encode(), {"nt": 0, "dslab": 0, "tile_w": 0, "fwd_dl": 2}[k])

if __name__ == "__main__":
    main()
