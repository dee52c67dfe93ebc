"""Drop-in for jdacs/fusion/depthfusion.py: MVSNet outputs -> gipuma folder layout -> fused point cloud.

Same names, arguments and FILE BYTES as the reference for the format glue --
  read_pfm / save_pfm / load_pfm / write_pfm        depthfusion.py:33-147 (= jdacs/datasets/data_io.py)
  load_cam                                          :150-186   camera txt -> [2,4,4]
  read_gipuma_dmb / write_gipuma_dmb                :101-137   int32 type=1, H, W, channels + float32 payload
  mvsnet_to_gipuma_dmb / mvsnet_to_gipuma_cam       :188-221   .pfm -> disp.dmb, cam txt -> "<view>.jpg.P" (K [R|t], 3x4 text)
  fake_gipuma_normal                                :224-241   constant normal (1,1,1)/sqrt(3) masked by depth > 0
  mvsnet_to_gipuma                                  :278-319   the folder layout fusibile expects (cams/, images/, 2333__<view>/)
  probability_filter                                :341-363   depth[prob < threshold] = 0 -> "<view>_prob_filtered.pfm"
-- and where the reference shells out to the `fusibile` CUDA executable (depth_map_fusion, :366-386), the fusion kernel runs HERE
on the MI355X through the C ABI (mvs_fusibile_fuse, csrc/fusibile.hip): run_fusibile() reads the folder the way the program's
runFusibile does (main.cpp:548-857), prepares the cameras like cameraGeometryUtils.h:194-440, launches one kernel per reference
camera, compacts the points (fusibile.cu:281-320) and writes `consistencyCheck-<timestamp>/final3d_model.ply` with the program's
byte layout (displayUtils.h:80-136).  Quirks kept: write_gipuma_dmb stores a 3-channel image channel-PLANAR while the program
reads normals.dmb as pixel-interleaved (fileIoUtils.h:213-248) -- the fake normals are read back exactly that way.

Host code is numpy (like the reference's); only the fusion kernel needs the GPU."""
import os
import re
import shutil
import time
from struct import pack, unpack

import numpy as np

from ..datasets.data_io import read_pfm, save_pfm


def load_pfm(filename):
    """depthfusion.py:140-142."""
    return np.array(read_pfm(filename)[0], dtype=np.float32)


def write_pfm(filename, image, scale=1):
    save_pfm(filename, image, scale)


def load_cam(file, interval_scale=1, max_depth=256):
    """depthfusion.py:150-186: `file` is an open text file; returns cam[2,4,4] float64 (extrinsic, intrinsic + depth range row)."""
    cam = np.zeros((2, 4, 4))
    words = file.read().split()
    for i in range(4):
        for j in range(4):
            cam[0][i][j] = words[4 * i + j + 1]
    for i in range(3):
        for j in range(3):
            cam[1][i][j] = words[3 * i + j + 18]
    if len(words) == 29:
        cam[1][3][0] = words[27]
        cam[1][3][1] = float(words[28]) * interval_scale
        cam[1][3][2] = 1100
        cam[1][3][3] = cam[1][3][0] + cam[1][3][1] * cam[1][3][2]
    elif len(words) == 30:
        cam[1][3][0] = words[27]
        cam[1][3][1] = float(words[28]) * interval_scale
        cam[1][3][2] = words[29]
        cam[1][3][3] = cam[1][3][0] + cam[1][3][1] * cam[1][3][2]
    elif len(words) == 31:
        cam[1][3][0] = words[27]
        cam[1][3][1] = float(words[28]) * interval_scale
        cam[1][3][2] = words[29]
        cam[1][3][3] = words[30]
    return cam


def read_gipuma_dmb(path):
    """depthfusion.py:101-113."""
    with open(path, "rb") as fid:
        unpack("<i", fid.read(4))
        height = unpack("<i", fid.read(4))[0]
        width = unpack("<i", fid.read(4))[0]
        channel = unpack("<i", fid.read(4))[0]
        array = np.fromfile(fid, np.float32)
    array = array.reshape((width, height, channel), order="F")
    return np.transpose(array, (1, 0, 2)).squeeze()


def write_gipuma_dmb(path, image):
    """depthfusion.py:116-137 (a [H,W,3] image is written channel-planar: the reference transposes to [3,H,W] first)."""
    image_shape = np.shape(image)
    width, height = image_shape[1], image_shape[0]
    channels = image_shape[2] if len(image_shape) == 3 else 1
    if len(image_shape) == 3:
        image = np.transpose(image, (2, 0, 1)).squeeze()
    with open(path, "wb") as fid:
        fid.write(pack("<i", 1))
        fid.write(pack("<i", height))
        fid.write(pack("<i", width))
        fid.write(pack("<i", channels))
        np.asarray(image).tofile(fid)


def mvsnet_to_gipuma_dmb(in_path, out_path):
    write_gipuma_dmb(out_path, load_pfm(in_path))


def mvsnet_to_gipuma_cam(in_path, out_path, max_depth=256):
    """depthfusion.py:198-221: K [R|t] (intrinsic with a zeroed 4th row times the extrinsic), 3 text rows + an empty line."""
    with open(in_path) as fh:
        cam = load_cam(fh, max_depth=max_depth)
    extrinsic = cam[0]
    intrinsic = cam[1]
    intrinsic[3][:] = 0
    projection_matrix = np.matmul(intrinsic, extrinsic)[0:3]
    with open(out_path, "w") as f:
        for i in range(3):
            for j in range(4):
                f.write(str(projection_matrix[i][j]) + " ")
            f.write("\n")
        f.write("\n")


def fake_gipuma_normal(in_depth_path, out_normal_path):
    """depthfusion.py:224-241."""
    depth_image = read_gipuma_dmb(in_depth_path)
    h, w = np.shape(depth_image)[0], np.shape(depth_image)[1]
    normal_image = np.tile(np.reshape(np.ones_like(depth_image), (h, w, 1)), [1, 1, 3]) / 1.732050808
    mask_image = np.float32(np.tile(np.reshape(np.squeeze(np.where(depth_image > 0, 1, 0)), (h, w, 1)), [1, 1, 3]))
    write_gipuma_dmb(out_normal_path, np.float32(np.multiply(normal_image, mask_image)))


def mvsnet_to_gipuma(scan_folder, scan, dtu_test_root, gipuma_point_folder, num_views=49):
    """depthfusion.py:278-319 (the reference hard-codes the 49 views of a DTU scan)."""
    image_folder = os.path.join(dtu_test_root, scan, "images")
    cam_folder = os.path.join(dtu_test_root, scan, "cams")
    depth_folder = os.path.join(scan_folder, "depth_est")
    gipuma_cam_folder = os.path.join(gipuma_point_folder, "cams")
    gipuma_image_folder = os.path.join(gipuma_point_folder, "images")
    for d in (gipuma_point_folder, gipuma_cam_folder, gipuma_image_folder):
        if not os.path.isdir(d):
            os.mkdir(d)
    for view in range(num_views):
        mvsnet_to_gipuma_cam(os.path.join(cam_folder, "{:08d}_cam.txt".format(view)),
                             os.path.join(gipuma_cam_folder, "{:08d}.jpg.P".format(view)))
    for view in range(num_views):
        shutil.copy(os.path.join(image_folder, "{:08d}.jpg".format(view)), os.path.join(gipuma_image_folder, "{:08d}.jpg".format(view)))
    gipuma_prefix = "2333__"
    for view in range(num_views):
        sub = os.path.join(gipuma_point_folder, gipuma_prefix + "{:08d}".format(view))
        if not os.path.isdir(sub):
            os.mkdir(sub)
        mvsnet_to_gipuma_dmb(os.path.join(depth_folder, "{:08d}_prob_filtered.pfm".format(view)), os.path.join(sub, "disp.dmb"))
        fake_gipuma_normal(os.path.join(sub, "disp.dmb"), os.path.join(sub, "normals.dmb"))


def probability_filter(scan_folder, prob_threshold, num_views=49):
    """depthfusion.py:341-363."""
    depth_folder = os.path.join(scan_folder, "depth_est")
    prob_folder = os.path.join(scan_folder, "confidence")
    for view in range(num_views):
        depth_map = load_pfm(os.path.join(depth_folder, "{:08d}.pfm".format(view)))
        prob_map = load_pfm(os.path.join(prob_folder, "{:08d}.pfm".format(view)))
        depth_map[prob_map < prob_threshold] = 0
        write_pfm(os.path.join(depth_folder, "{:08d}_prob_filtered.pfm".format(view)), depth_map)


# ------------------------------------------------------------------------------------------------------------------------
# the fusibile program, in process
# ------------------------------------------------------------------------------------------------------------------------
def read_p_file(path):
    """readPFileStrechaPmvs (fileIoUtils.h:83-110): up to 4 text rows of floats (CONTOUR lines skipped) -> P [3,4] float32."""
    P = np.zeros((4, 4), dtype=np.float32)
    i = 0
    with open(path) as fh:
        for line in fh:
            if i >= 4:
                break
            if "CONTOUR" in line:
                continue
            for j, tok in enumerate(line.split(" ")):
                tok = tok.strip()
                if tok and j < 4:
                    P[i, j] = np.float32(float(tok))
            i += 1
    return P[:3]


def fusibile_cameras(P_list):
    """Camera_cu fields of cameraGeometryUtils.h:310-440 for transformP = false, cam_scale = 1 -> (cams [V,32] float32, f):
    every P decomposed into K, R, C (decomposeProjectionMatrix); P rebuilt as K_0 [R | -R C] with the FIRST camera's K; M_inv;
    the camera centre from the minors of the rebuilt P.  Layout of a row: P[12], M_inv[9], P[:,3] [3], C[3], 5 unused.
    Two host-side conventions differ from cv::decomposeProjectionMatrix and are no-ops for the .P files the pipeline itself writes
    (write_gipuma_cam: K [R | t] with K[2,2] = 1, det R = +1): K is normalised by K[2,2] and (K, R) change sign together when
    det R < 0; a P file scaled by s != 1 would give the program's own decomposition K[2,2] = s.  Colours: the images are decoded by
    PIL, the program uses cv::imread -- for .jpg inputs the two decoders may differ by +-1 per channel (the tests use .png)."""
    from scipy.linalg import rq
    Ks, Rts = [], []
    for P in P_list:
        P = np.asarray(P, dtype=np.float32)
        K, R = rq(P[:, :3].astype(np.float64))
        sgn = np.diag(np.sign(np.diag(K)))
        K, R = K @ sgn, sgn @ R                      # positive diagonal of K
        if np.linalg.det(R) < 0:
            K, R = -K, -R
        K = K / K[2, 2]
        T = np.linalg.svd(P.astype(np.float64))[2][-1]          # P T = 0
        C = T[:3] / T[3]
        Ks.append(K.astype(np.float32))
        Rts.append(np.concatenate([R, (-(R @ C)).reshape(3, 1)], axis=1).astype(np.float32))
    K0 = Ks[0]
    cams = np.zeros((len(P_list), 32), dtype=np.float32)
    for i, Rt in enumerate(Rts):
        P = np.matmul(K0, Rt).astype(np.float32)
        minor = lambda cols: np.float32(np.linalg.det(P[:, cols].astype(np.float64)))
        Ch = np.array([minor([1, 2, 3]), -minor([0, 2, 3]), minor([0, 1, 3]), -minor([0, 1, 2])], dtype=np.float32)
        cams[i, 0:12] = P.reshape(-1)
        cams[i, 12:21] = np.linalg.inv(P[:, :3]).astype(np.float32).reshape(-1)
        cams[i, 21:24] = P[:, 3]
        cams[i, 24:27] = (Ch / Ch[3])[:3]
    return cams, float(K0[0, 0])


def fuse_depth_maps(normals_depths, images, cams, f, disp_thresh, normal_thresh, num_consistent, save_texture=True):
    """The program's GPU part (fusibile.cu:322-440): normals_depths [V,H,W,4] and images [V,H,W,4] (or None) as float32 torch tensors
    on the GPU, cams [V,32] -> points [N,12] float32 tensor (coord xyz 0, normal xyz 0, colour xyz 0), camera after camera, row-major,
    only points whose x, y and z are all non-zero (copy_point_cloud_to_host, :281-320)."""
    import torch

    from ... import _lib
    from ...ops import _p, _stream
    lib = _lib.get()
    nd = normals_depths
    if nd.device.type != lib.device_type or nd.dtype != torch.float32 or nd.dim() != 4 or nd.shape[-1] != 4:
        raise TypeError("fuse_depth_maps: normals_depths must be a float32 [V,H,W,4] tensor on a '%s' device" % lib.device_type)
    nd = nd.contiguous()
    V, H, W = nd.shape[:3]
    if images is not None:
        if tuple(images.shape) != (V, H, W, 4) or images.dtype != torch.float32 or images.device != nd.device:
            raise ValueError("fuse_depth_maps: images must be float32 %s on %s" % ((V, H, W, 4), nd.device))
        images = images.contiguous()
    cams_t = torch.as_tensor(np.ascontiguousarray(cams, dtype=np.float32)).to(nd.device)
    if tuple(cams_t.shape) != (V, 32):
        raise ValueError("fuse_depth_maps: cams must be [%d,32], got %s" % (V, tuple(cams_t.shape)))
    subset = torch.arange(V, dtype=torch.int32, device=nd.device)      # selectViews(..., viewSel=false): all views (main.cpp:719)
    out = torch.empty((H, W, 12), dtype=torch.float32, device=nd.device)
    pts = []
    for ref in range(V):
        lib.call("mvs_fusibile_fuse", _p(nd), _p(images), _p(cams_t), _p(subset), V, V, H, W, ref, float(f), float(disp_thresh),
                 float(normal_thresh), int(num_consistent), int(bool(save_texture)), _p(out), _stream(nd))
        flat = out.view(-1, 12)
        keep = (flat[:, 0] != 0) & (flat[:, 1] != 0) & (flat[:, 2] != 0)
        pts.append(flat[keep].clone())
    return torch.cat(pts, 0) if pts else torch.zeros((0, 12), dtype=torch.float32, device=nd.device)


def ply_bytes(points):
    """storePlyFileBinaryPointCloud (displayUtils.h:80-136): float x y z + uchar red green blue = (char)(int) colour[2], [1], [0]."""
    pts = np.asarray(points, dtype=np.float32)
    n = pts.shape[0]
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode("ascii")
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    xyz = pts[:, 0:3].copy()
    xyz[~np.isfinite(xyz).all(axis=1)] = 0
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    ch = lambda v: (np.trunc(v).astype(np.int64) & 0xFF).astype(np.uint8)
    rec["r"], rec["g"], rec["b"] = ch(pts[:, 10]), ch(pts[:, 9]), ch(pts[:, 8])
    return hdr + rec.tobytes()


def _read_dmb_raw(path):
    """readDmb / readDmbNormal (fileIoUtils.h:213-285): header type, h, w, nb + float payload reinterpreted as [h,w,nb] interleaved."""
    with open(path, "rb") as fid:
        typ, h, w, nb = unpack("<4i", fid.read(16))
        if typ != 1:
            raise ValueError("%s: only float .dmb files are supported" % path)
        data = np.fromfile(fid, np.float32, h * w * nb)
    return data.reshape(h, w, nb)


def _load_image_bgra(path, h, w):
    """cv::imread(IMREAD_COLOR) -> float [h,w,4] in OpenCV's b, g, r order + an (uninitialised in the program) alpha = 0."""
    from PIL import Image
    rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)
    if rgb.shape[0] != h or rgb.shape[1] != w:
        raise ValueError("%s: image is %dx%d, depth maps are %dx%d" % (path, rgb.shape[1], rgb.shape[0], w, h))
    out = np.zeros((h, w, 4), dtype=np.float32)
    out[..., 0], out[..., 1], out[..., 2] = rgb[..., 2], rgb[..., 1], rgb[..., 0]
    return out


def run_fusibile(point_folder, cam_folder, image_folder, disp_thresh, num_consistent, normal_thresh_deg=360.0, device="cuda:0",
                 timestamp=None):
    """What the executable does for `-input_folder point_folder/ -p_folder cam_folder/ -images_folder image_folder/` (main.cpp:548-857):
    sub-folders "<date>_<time>_<id>" starting with '2' hold disp.dmb / normals.dmb of view <id>; cameras `<id>.<ext>.P`; images
    `<id>.png|.jpg|.ppm`.  Returns the path of the written .ply."""
    import torch
    subs = sorted(d for d in os.listdir(point_folder) if os.path.isdir(os.path.join(point_folder, d)))
    ids, names = [], []
    for d in subs:
        if d.count("_") < 2 or d[0] != "2":
            continue
        first = d.find("_") + 1
        ident = d[d[first:].find("_") + first + 1:]
        for ext in (".png", ".jpg", ".ppm"):
            if os.access(os.path.join(image_folder, ident + ext), os.R_OK):
                ids.append((d, ident))
                names.append(ident + ext)
                break
    if not ids:
        raise RuntimeError("run_fusibile: no '<date>_<time>_<id>' result folders with a matching image under %s" % point_folder)
    Ps = [read_p_file(os.path.join(cam_folder, n + ".P")) for n in names]
    cams, f = fusibile_cameras(Ps)
    nds, imgs = [], []
    for (d, ident), n in zip(ids, names):
        depth = _read_dmb_raw(os.path.join(point_folder, d, "disp.dmb"))[..., 0]
        normals = _read_dmb_raw(os.path.join(point_folder, d, "normals.dmb"))
        h, w = depth.shape
        nds.append(np.concatenate([normals[..., :3], depth[..., None]], axis=2).astype(np.float32))
        imgs.append(_load_image_bgra(os.path.join(image_folder, n), h, w))
    dev = torch.device(device)
    pts = fuse_depth_maps(torch.from_numpy(np.stack(nds)).to(dev), torch.from_numpy(np.stack(imgs)).to(dev), cams, f, disp_thresh,
                          np.float32(normal_thresh_deg) * np.float32(np.pi) / np.float32(180.0), num_consistent, True)
    ts = timestamp or time.strftime("%Y%m%d-%H%M%S")
    out_dir = os.path.join(point_folder, "consistencyCheck-%s" % ts)
    os.makedirs(out_dir, exist_ok=True)
    ply = os.path.join(out_dir, "final3d_model.ply")
    with open(ply, "wb") as fh:
        fh.write(ply_bytes(pts.cpu().numpy()))
    return ply


def depth_map_fusion(point_folder, fusibile_exe_path, disp_thresh, num_consistent, device="cuda:0"):
    """depthfusion.py:366-386 with the program's arguments (depth_min 0.001, depth_max 100000, normal_thresh 360 degrees);
    `fusibile_exe_path` is accepted for signature compatibility and ignored: the fusion runs in process on the GPU."""
    return run_fusibile(point_folder, os.path.join(point_folder, "cams"), os.path.join(point_folder, "images"), disp_thresh,
                        int(num_consistent), 360.0, device)
