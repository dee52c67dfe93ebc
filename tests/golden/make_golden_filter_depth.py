#!/usr/bin/env python3
"""Generate tests/golden/g12_filter_depth.npz by EXECUTING the reference's own scan-level filter (build container only;
jdacs/eval.py:61-108 `read_camera_parameters`, `read_img`, `save_mask`, `read_pair_file`, :169-224 the geometric check, :340-447
`filter_depth`) on a small synthetic scan folder written to a temporary directory.

    python tests/golden/make_golden_filter_depth.py

jdacs/eval.py cannot be imported as a module here (cv2, plyfile, torchvision, tensorboardX are absent and config.py parses
sys.argv), so the function definitions are taken out of its syntax tree and executed in a namespace that holds numpy, PIL, the
reference's own `read_pfm` (jdacs/datasets/data_io.py, imported from its file), a `cv2` stand-in whose ONLY member is `remap` =
oracle.geo_filter_np.remap_bilinear_cv (parity unpinned: see that module's header) and `PlyElement` / `PlyData` stand-ins that
CAPTURE the structured vertex array instead of writing it (plyfile is not installed: the .ply container bytes are unpinned, the
vertex table that goes into it is pinned).  Nothing of the reference's text is stored: the fixture holds the scan folder's input
files as bytes (cameras, pair file, JPEG images, PFM depth / confidence maps -- all generated here) and what the reference's
function produced from them (mask PNG bytes, the vertex table)."""
import ast
import importlib.util
import io
import os
import sys
import tempfile
import types
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle.geo_filter_np import remap_bilinear_cv  # noqa: E402

SRC = "/root/reference/jdacs/eval.py"
spec = importlib.util.spec_from_file_location("ref_data_io", "/root/reference/jdacs/datasets/data_io.py")
ref_io = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_io)

tree = ast.parse(open(SRC).read())
names = ("read_camera_parameters", "read_img", "save_mask", "read_pair_file", "reproject_with_depth", "check_geometric_consistency",
         "filter_depth")
wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
assert len(wanted) == len(names)
captured = {}


class _PlyElement:
    @staticmethod
    def describe(arr, name):
        captured["vertex_all"] = arr.copy()
        captured["element"] = name
        return arr


class _PlyData:
    def __init__(self, els):
        self.els = els

    def write(self, filename):
        captured["plyfilename"] = filename


ns = {"np": np, "os": os, "Image": Image, "read_pfm": ref_io.read_pfm, "PlyElement": _PlyElement, "PlyData": _PlyData,
      "cv2": types.SimpleNamespace(remap=remap_bilinear_cv, INTER_LINEAR=1), "args": types.SimpleNamespace(display=False)}
exec(compile(ast.Module(body=wanted, type_ignores=[]), SRC, "exec"), ns)

# ---- a synthetic scan: 5 views of a smooth surface; the function hard-codes DTU's geometry (images resized to 1152x864,
# intrinsics scaled by 1/4 * (1152/1600, 864/1200), colours taken at [::4, ::4]) -> depth maps are 216 x 288 ----
H, W, NV = 216, 288, 4
rng = np.random.RandomState(3)


def rot(ax, ay):
    ax, ay = np.radians(ax), np.radians(ay)
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    return ry @ rx


def camera_file(v):
    """full-resolution (1600x1200) intrinsics as the cam.txt files hold them, and the extrinsics"""
    K = np.array([[2890.0, 0, 800.0], [0, 2890.0, 600.0], [0, 0, 1]])
    E = np.eye(4)
    if v:
        s = 1.0 if v % 2 else -1.0
        E[:3, :3] = rot(s * (1.0 + 0.8 * v), -s * (0.8 + 0.5 * v))
        E[:3, 3] = [s * (22.0 + 8 * v), -s * 7.0 * v, 2.5 * v]
    return K, E


def surface(X, Y):
    return 640.0 + 0.10 * X + 0.04 * Y + 20.0 * np.sin(X / 60.0) * np.cos(Y / 80.0)


def render_depth(K, E, iters=25):
    Ki, Ei = np.linalg.inv(K), np.linalg.inv(E)
    x, y = np.meshgrid(np.arange(W), np.arange(H))
    rays = Ki @ np.vstack((x.reshape(-1), y.reshape(-1), np.ones(H * W)))
    d = np.full(H * W, 640.0)
    for _ in range(iters):
        pw = Ei @ np.vstack((rays * d, np.ones(H * W)))
        d = d + (surface(pw[0], pw[1]) - pw[2])
    return d.reshape(H, W).astype(np.float32)


def write_pfm(path, arr):
    with open(path, "wb") as f:
        f.write(b"Pf\n%d %d\n-1.000000\n" % (arr.shape[1], arr.shape[0]))
        f.write(np.flipud(arr).astype("<f4").tobytes())


files = {}
with tempfile.TemporaryDirectory() as tmp:
    scan, out = os.path.join(tmp, "scan1"), os.path.join(tmp, "out")
    for d in ("cams", "images"):
        os.makedirs(os.path.join(scan, d))
    for d in ("depth_est", "confidence"):
        os.makedirs(os.path.join(out, d))
    lines = ["%d" % NV]
    for v in range(NV):
        others = [u for u in range(NV) if u != v]
        lines += ["%d" % v, "%d " % len(others) + " ".join("%d %.2f" % (u, 100.0 - u) for u in others)]
    files["pair.txt"] = ("\n".join(lines) + "\n").encode()
    open(os.path.join(scan, "pair.txt"), "wb").write(files["pair.txt"])
    for v in range(NV):
        K, E = camera_file(v)
        txt = "extrinsic\n" + "\n".join(" ".join("%.6f" % x for x in row) for row in E) + "\n\nintrinsic\n" + \
              "\n".join(" ".join("%.6f" % x for x in row) for row in K) + "\n\n425.0 2.5\n"
        files["cams/%08d_cam.txt" % v] = txt.encode()
        open(os.path.join(scan, "cams/%08d_cam.txt" % v), "wb").write(txt.encode())
        yy, xx = np.mgrid[0:300, 0:400]
        img = np.stack([127 + 100 * np.sin(xx / 23.0 + v), 127 + 100 * np.cos(yy / 17.0 - v), 60 + 0.4 * xx], -1).clip(0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="JPEG", quality=92)
        files["images/%08d.jpg" % v] = buf.getvalue()
        open(os.path.join(scan, "images/%08d.jpg" % v), "wb").write(buf.getvalue())
        Kq, _ = ns["read_camera_parameters"](os.path.join(scan, "cams/%08d_cam.txt" % v))     # the intrinsics the filter will use
        depth = render_depth(Kq.astype(np.float64), E)
        if v == 2:
            depth = (depth * (1.0 + 0.006 * rng.randn(H, W))).astype(np.float32)      # noisy view
        if v == 3:
            depth[40:80, 100:160] = 0.0                                               # holes
        conf = (0.55 + 0.45 * rng.rand(H, W)).astype(np.float32)
        # values exactly representable in half precision: the fixture (zlib) stores half the entropy
        depth, conf = depth.astype(np.float16).astype(np.float32), conf.astype(np.float16).astype(np.float32)
        for kind, arr in (("depth_est", depth), ("confidence", conf)):
            p = os.path.join(out, "%s/%08d.pfm" % (kind, v))
            write_pfm(p, arr)
            files["%s/%08d.pfm" % (kind, v)] = open(p, "rb").read()
    ns["filter_depth"](scan, out, os.path.join(out, "scan1.ply"))
    masks = {}
    for v in range(NV):
        for kind in ("photo", "geo", "final"):
            masks["mask/%08d_%s.png" % (v, kind)] = open(os.path.join(out, "mask/%08d_%s.png" % (v, kind)), "rb").read()

va = captured["vertex_all"]
assert captured["element"] == "vertex" and va.dtype.names == ("x", "y", "z", "red", "green", "blue")
xyz, rgb = np.stack([va["x"], va["y"], va["z"]], 1), np.stack([va["red"], va["green"], va["blue"]], 1)
# every 4th vertex in full + exact float64 / integer sums over all of them (the table is ~0.9 MB of incompressible floats)
res = {"nvert": np.array(len(va)), "xyz_every4": xyz[::4], "rgb_every4": rgb[::4], "xyz_sum": xyz.astype(np.float64).sum(0),
       "xyz_abs_sum": np.abs(xyz.astype(np.float64)).sum(0), "rgb_sum": rgb.astype(np.int64).sum(0),
       "file_names": np.array(sorted(files)), "mask_names": np.array(sorted(masks))}
for i, k in enumerate(sorted(files)):
    res["file_%d" % i] = np.frombuffer(files[k], dtype=np.uint8)
for i, k in enumerate(sorted(masks)):
    res["maskfile_%d" % i] = np.frombuffer(masks[k], dtype=np.uint8)
path = os.path.join(HERE, "g12_filter_depth.npz")
np.savez_compressed(path, **res)
print("g12_filter_depth %.1f KB; %d vertices from %d views" % (os.path.getsize(path) / 1024, len(va), NV))
