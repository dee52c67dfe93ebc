#!/usr/bin/env python3
"""Generate tests/golden/g11_gipuma_formats.npz by EXECUTING the reference's own format glue (build container only):
jdacs/fusion/depthfusion.py:33-241,341-363 (read_pfm, save_pfm, load_pfm, write_pfm, load_cam, read_gipuma_dmb, write_gipuma_dmb,
mvsnet_to_gipuma_dmb, mvsnet_to_gipuma_cam, fake_gipuma_normal, probability_filter) and jdacs/eval.py:110-123 (write_depth_img).

    python tests/golden/make_golden_fusion.py

Neither file can be imported as a module here (cv2, pylab, plyfile, torchvision are absent; config.py parses sys.argv), so the
function definitions are taken out of the syntax trees and executed in a namespace holding what they use (numpy, re, sys, os,
errno, struct.pack/unpack, PIL.Image).  Nothing of the reference's text is stored: the fixture holds seeded inputs (depth and
confidence maps, a DTU-style camera file generated here) and the BYTES the reference's functions wrote for them."""
import ast
import errno
import os
import re
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np
from struct import pack, unpack
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def take(path, names, ns):
    tree = ast.parse(open(path).read())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    # the LAST definition of a name wins in a module; keep source order
    assert {n.name for n in wanted} == set(names), sorted(set(names) - {n.name for n in wanted})
    exec(compile(ast.Module(body=wanted, type_ignores=[]), path, "exec"), ns)
    return ns


ns = {"np": np, "re": re, "sys": sys, "os": os, "pack": pack, "unpack": unpack, "shutil": shutil}
take("/root/reference/jdacs/fusion/depthfusion.py",
     ["read_pfm", "save_pfm", "load_pfm", "write_pfm", "load_cam", "read_gipuma_dmb", "write_gipuma_dmb", "mvsnet_to_gipuma_dmb",
      "mvsnet_to_gipuma_cam", "fake_gipuma_normal", "probability_filter"], ns)
ns2 = {"np": np, "os": os, "errno": errno, "Image": Image}
take("/root/reference/jdacs/eval.py", ["write_depth_img"], ns2)


def cam_txt(rng, with_depth_words=4):
    """A DTU-style camera file: 'extrinsic' + 16 numbers, 'intrinsic' + 9 numbers, then depth_min interval [ndepth [depth_max]]."""
    ang = rng.uniform(-0.2, 0.2, 3)
    cx, sx = np.cos(ang[0]), np.sin(ang[0]); cy, sy = np.cos(ang[1]), np.sin(ang[1]); cz, sz = np.cos(ang[2]), np.sin(ang[2])
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
         @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    t = rng.uniform(-300, 300, 3)
    E = np.eye(4); E[:3, :3] = R; E[:3, 3] = t
    K = np.array([[2892.33, 0, 823.205], [0, 2883.18, 619.071], [0, 0, 1]])
    lines = ["extrinsic"] + [" ".join(repr(float(v)) for v in row) for row in E] + ["", "intrinsic"] + \
            [" ".join(repr(float(v)) for v in row) for row in K] + ["", " ".join(["425.0", "2.5", "192", "933.8"][:with_depth_words])]
    return "\n".join(lines) + "\n"


def main():
    rng = np.random.default_rng(11)
    out = {}
    tmp = tempfile.mkdtemp(prefix="mvs_golden_fusion_")
    try:
        # ---- .dmb writer / reader, fake normals, pfm -> dmb ----
        h, w = 7, 9
        depth = (500 + 300 * rng.random((h, w))).astype(np.float32)
        depth[rng.random((h, w)) < 0.25] = 0          # holes: the fake normal is masked by depth > 0
        out["depth"] = depth
        ns["save_pfm"](os.path.join(tmp, "d.pfm"), depth)
        ns["mvsnet_to_gipuma_dmb"](os.path.join(tmp, "d.pfm"), os.path.join(tmp, "disp.dmb"))
        out["disp_dmb_bytes"] = np.frombuffer(open(os.path.join(tmp, "disp.dmb"), "rb").read(), dtype=np.uint8)
        ns["fake_gipuma_normal"](os.path.join(tmp, "disp.dmb"), os.path.join(tmp, "normals.dmb"))
        out["normals_dmb_bytes"] = np.frombuffer(open(os.path.join(tmp, "normals.dmb"), "rb").read(), dtype=np.uint8)
        out["disp_read_back"] = np.asarray(ns["read_gipuma_dmb"](os.path.join(tmp, "disp.dmb")), dtype=np.float32)
        out["normals_read_back"] = np.asarray(ns["read_gipuma_dmb"](os.path.join(tmp, "normals.dmb")), dtype=np.float32)
        colour = rng.random((h, w, 3)).astype(np.float32)
        out["colour"] = colour
        ns["write_gipuma_dmb"](os.path.join(tmp, "c.dmb"), colour)
        out["colour_dmb_bytes"] = np.frombuffer(open(os.path.join(tmp, "c.dmb"), "rb").read(), dtype=np.uint8)
        # ---- camera file -> load_cam, -> gipuma .P text ----
        for tag, nwords in (("a", 4), ("b", 2), ("c", 3)):
            txt = cam_txt(rng, nwords)
            out["cam_txt_" + tag] = np.frombuffer(txt.encode("ascii"), dtype=np.uint8)
            path = os.path.join(tmp, "cam_%s.txt" % tag)
            open(path, "w").write(txt)
            with open(path) as fh:
                out["cam_loaded_" + tag] = ns["load_cam"](fh)
            ns["mvsnet_to_gipuma_cam"](path, os.path.join(tmp, "cam_%s.P" % tag))
            out["cam_P_bytes_" + tag] = np.frombuffer(open(os.path.join(tmp, "cam_%s.P" % tag), "rb").read(), dtype=np.uint8)
        # ---- probability_filter over the 49 views the reference hard-codes ----
        scan = os.path.join(tmp, "scan")
        os.makedirs(os.path.join(scan, "depth_est"))
        os.makedirs(os.path.join(scan, "confidence"))
        d49 = (500 + 300 * rng.random((49, 4, 5))).astype(np.float32)
        p49 = rng.random((49, 4, 5)).astype(np.float32)
        for v in range(49):
            ns["save_pfm"](os.path.join(scan, "depth_est", "{:08d}.pfm".format(v)), d49[v])
            ns["save_pfm"](os.path.join(scan, "confidence", "{:08d}.pfm".format(v)), p49[v])
        ns["probability_filter"](scan, 0.8)
        out["pf_depth"], out["pf_prob"] = d49, p49
        out["pf_filtered_bytes"] = np.stack([np.frombuffer(open(os.path.join(scan, "depth_est", "{:08d}_prob_filtered.pfm".format(v)), "rb").read(),
                                                           dtype=np.uint8) for v in range(49)])
        # ---- depth PNG (eval.py:110-123) ----
        dimg = (400 + 700 * rng.random((12, 16))).astype(np.float32)   # (d - 500) / 2 spans < 0 and > 255: PIL's clipping on both ends
        out["png_depth"] = dimg
        png = os.path.join(tmp, "sub", "dir", "00000000.pfm.png")
        assert ns2["write_depth_img"](png, dimg) == 1
        out["png_bytes"] = np.frombuffer(open(png, "rb").read(), dtype=np.uint8)
        out["png_pixels"] = np.asarray(Image.open(png))
        out["pil_version"] = np.frombuffer(Image.__version__.encode() if hasattr(Image, "__version__") else b"", dtype=np.uint8)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, "g11_gipuma_formats.npz"), **out)
    print("wrote g11_gipuma_formats.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
