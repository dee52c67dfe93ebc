#!/usr/bin/env python3
"""Generate tests/golden/g10_geo_filter.npz by EXECUTING the reference's own geometric-consistency functions
(build container only; jdacs/eval.py:169-224 `reproject_with_depth`, `check_geometric_consistency`).

    python tests/golden/make_golden_geo.py

jdacs/eval.py cannot be imported as a module here (cv2, plyfile, torchvision, tensorboardX are absent and config.py
parses sys.argv), so the two function definitions are taken out of its syntax tree and executed in a namespace that holds
numpy and a `cv2` stand-in whose ONLY member is `remap` = oracle.geo_filter_np.remap_bilinear_cv (our restatement of
OpenCV's algorithm: the one unpinned piece, see that module's header).  Nothing of the reference's text is stored: the
fixture holds seeded inputs (depth maps rendered from a smooth synthetic surface seen by synthetic cameras, one view
perturbed, one with holes) and the functions' outputs, plus the aggregation of eval.py:385-388 over the source views."""
import ast
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.geo_filter_np import remap_bilinear_cv  # noqa: E402

SRC = "/root/reference/jdacs/eval.py"
tree = ast.parse(open(SRC).read())
wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("reproject_with_depth", "check_geometric_consistency")]
assert len(wanted) == 2
ns = {"np": np, "cv2": types.SimpleNamespace(remap=remap_bilinear_cv, INTER_LINEAR=1)}
exec(compile(ast.Module(body=wanted, type_ignores=[]), SRC, "exec"), ns)
check = ns["check_geometric_consistency"]


def rot(ax, ay):
    ax, ay = np.radians(ax), np.radians(ay)
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    return ry @ rx


def camera(v, h, w):
    K = np.array([[0.9 * w, 0, w / 2.0], [0, 0.9 * w, h / 2.0], [0, 0, 1]], dtype=np.float32)
    E = np.eye(4)
    if v:
        s = 1.0 if v % 2 else -1.0
        E[:3, :3] = rot(s * (1.5 + v), -s * (1.0 + 0.7 * v))
        E[:3, 3] = [s * (25.0 + 9 * v), -s * 8.0 * v, 3.0 * v]
    return K, E.astype(np.float32)


def surface(X, Y):
    """world depth of the synthetic surface z = f(x, y)"""
    return 600.0 + 0.12 * X + 0.05 * Y + 25.0 * np.sin(X / 70.0) * np.cos(Y / 90.0)


def render_depth(K, E, h, w, iters=20):
    """depth map of the surface seen by camera (K, E): fixed-point iteration on the ray parameter"""
    Ki = np.linalg.inv(K.astype(np.float64))
    Ei = np.linalg.inv(E.astype(np.float64))
    x, y = np.meshgrid(np.arange(w), np.arange(h))
    rays = Ki @ np.vstack((x.reshape(-1), y.reshape(-1), np.ones(h * w)))
    d = np.full(h * w, 600.0)
    for _ in range(iters):
        pw = Ei @ np.vstack((rays * d, np.ones(h * w)))
        d = d + (surface(pw[0], pw[1]) - pw[2])
    return d.reshape(h, w).astype(np.float32)


h, w, nviews = 48, 64, 5
rng = np.random.RandomState(0)
cams = [camera(v, h, w) for v in range(nviews)]
depths = [render_depth(K, E, h, w) for K, E in cams]
depths[2] = (depths[2] * (1.0 + 0.012 * rng.randn(h, w))).astype(np.float32)     # noisy view: fails the 1 % test in places
depths[3][10:20, 30:50] = 0.0                                                     # holes (sampled depth 0)
depths[4] = (depths[4] + 4.0).astype(np.float32)                                  # biased view
conf = rng.rand(h, w).astype(np.float32)
out = {"depth_ref": depths[0], "conf_ref": conf, "K": np.stack([c[0] for c in cams]), "E": np.stack([c[1] for c in cams]),
       "depth_src": np.stack(depths[1:])}
geo_sum = 0
reps = []
for v in range(1, nviews):
    mask, rep, xs, ys = check(depths[0], cams[0][0], cams[0][1], depths[v], cams[v][0], cams[v][1])
    out["mask%d" % v], out["reproj%d" % v], out["x_src%d" % v], out["y_src%d" % v] = mask, rep, xs, ys
    geo_sum = geo_sum + mask.astype(np.int32)                     # eval.py:379
    reps.append(rep)
    print("view %d consistent fraction %.3f" % (v, mask.mean()))
out["geo_count"] = geo_sum
out["depth_avg"] = (sum(reps) + depths[0]) / (geo_sum + 1)       # eval.py:385
out["final_mask"] = np.logical_and(conf > 0.8, geo_sum >= 3)     # eval.py:369,387-388
path = os.path.join(HERE, "g10_geo_filter.npz")
np.savez_compressed(path, **out)
print("g10_geo_filter %.1f KB; final mask fraction %.3f; depth_avg dtype %s" % (os.path.getsize(path) / 1024, out["final_mask"].mean(),
                                                                               out["depth_avg"].dtype))
