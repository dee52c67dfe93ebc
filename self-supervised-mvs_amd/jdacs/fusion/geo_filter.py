"""Drop-in for the geometric-consistency filter of jdacs/eval.py.

  check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src)
      -> mask, depth_reprojected, x2d_src, y2d_src                                        (eval.py:210-224)
  filter_depth_view(depth_ref, confidence, intrinsics_ref, extrinsics_ref, src_depths, src_intrinsics, src_extrinsics)
      -> photo_mask, geo_mask, final_mask, geo_count, depth_avg                           (eval.py:357-396, per reference view)

Same argument meaning and results as the reference's numpy / cv2 code, but the depth maps are torch tensors on the GPU
(straight from MVSNet.forward) and ALL source views of a reference view are handled by one HIP kernel
(csrc/geo_filter.hip).  The few 3x3 / 4x4 camera products are formed on the host in float32 exactly like the reference
does (np.linalg.inv / np.matmul of float32 arrays).

  filter_depth(scan_folder, out_folder, plyfilename)                                      (eval.py:340-447, the scan-level loop)
      pair file, cameras, images, PFM maps in; mask PNGs and the fused, coloured point cloud (.ply) out -- with the reference's
      hard-coded DTU geometry (read_camera_parameters / read_img, eval.py:61-82) and without its plyfile / cv2 dependencies."""
import ctypes as C
import os

import numpy as np
import torch

from ... import _lib
from ...ops import _p, _ptr_array, _stream


def _mats(K_ref, E_ref, K_srcs, E_srcs):
    f32 = lambda m: np.asarray(m.detach().cpu().numpy() if isinstance(m, torch.Tensor) else m, dtype=np.float32)
    K_ref, E_ref = f32(K_ref), f32(E_ref)
    out = [np.linalg.inv(K_ref).reshape(-1), K_ref.reshape(-1)]
    E_ref_inv = np.linalg.inv(E_ref)
    for K, E in zip(K_srcs, E_srcs):
        K, E = f32(K), f32(E)
        out += [np.matmul(E, E_ref_inv)[:3].reshape(-1), K.reshape(-1), np.linalg.inv(K).reshape(-1),
                np.matmul(E_ref, np.linalg.inv(E))[:3].reshape(-1)]
    return np.concatenate(out).astype(np.float64)


def geo_consistency(depth_ref, K_ref, E_ref, src_depths, K_srcs, E_srcs, pix_thresh=1.0, rel_thresh=0.01, want_views=False):
    """depth_ref [H,W] float32 GPU tensor, src_depths: list of V tensors [H,W] -> (count int32 [H,W], depth_sum fp32 [H,W]
    [, masks bool [V,H,W], reproj fp32 [V,H,W], xy_src fp32 [V,2,H,W]])."""
    lib = _lib.get()
    if depth_ref.device.type != lib.device_type or depth_ref.dtype != torch.float32 or depth_ref.dim() != 2:
        raise TypeError("geo_consistency: depth_ref must be a float32 [H,W] tensor on a '%s' device" % lib.device_type)
    h, w = depth_ref.shape
    v = len(src_depths)
    if not (v == len(K_srcs) == len(E_srcs)) or v < 1:
        raise ValueError("geo_consistency: need one intrinsics / extrinsics pair per source depth map")
    depth_ref = depth_ref.contiguous()
    srcs = []
    for d in src_depths:
        if tuple(d.shape) != (h, w) or d.dtype != torch.float32 or d.device != depth_ref.device:
            raise ValueError("geo_consistency: source depth maps must be float32 %s on %s" % ((h, w), depth_ref.device))
        srcs.append(d.contiguous())
    dev = depth_ref.device
    mats = torch.from_numpy(_mats(K_ref, E_ref, K_srcs, E_srcs)).to(dev)
    count = torch.empty((h, w), dtype=torch.int32, device=dev)
    dsum = torch.empty((h, w), dtype=torch.float32, device=dev)
    masks = torch.empty((v, h, w), dtype=torch.uint8, device=dev) if want_views else None
    reproj = torch.empty((v, h, w), dtype=torch.float32, device=dev) if want_views else None
    xy = torch.empty((v, 2, h, w), dtype=torch.float32, device=dev) if want_views else None
    lib.call("mvs_geo_consistency", _p(depth_ref), _ptr_array(srcs), _p(mats), v, h, w, float(pix_thresh), float(rel_thresh),
             _p(count), _p(dsum), _p(masks), _p(reproj), _p(xy), _stream(depth_ref))
    if want_views:
        return count, dsum, masks.bool(), reproj, xy
    return count, dsum


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """eval.py:210-224 for one source view -> mask [H,W] bool, depth_reprojected [H,W], x2d_src [H,W], y2d_src [H,W]."""
    _, _, masks, reproj, xy = geo_consistency(depth_ref, intrinsics_ref, extrinsics_ref, [depth_src], [intrinsics_src],
                                              [extrinsics_src], want_views=True)
    return masks[0], reproj[0], xy[0, 0], xy[0, 1]


def filter_depth_view(depth_ref, confidence, intrinsics_ref, extrinsics_ref, src_depths, src_intrinsics, src_extrinsics,
                      photo_thresh=0.8, min_views=3):
    """The per-reference-view body of filter_depth (eval.py:357-396): masks + the averaged depth map
    (sum of the consistent reprojected depths + the reference depth) / (count + 1), float64 like the reference's."""
    count, dsum = geo_consistency(depth_ref, intrinsics_ref, extrinsics_ref, src_depths, src_intrinsics, src_extrinsics)
    photo_mask = confidence > photo_thresh
    geo_mask = count >= min_views
    depth_avg = (dsum + depth_ref).double() / (count + 1).double()
    return {"photo_mask": photo_mask, "geo_mask": geo_mask, "final_mask": photo_mask & geo_mask, "geo_count": count,
            "depth_avg": depth_avg}


# ---- the scan-level loop of filter_depth (eval.py:340-447) and the small readers / writers around it (eval.py:61-108) ----------
def read_camera_parameters(filename):
    """eval.py:61-73: extrinsics 4x4 (lines 1-4), intrinsics 3x3 (lines 7-9) of a <view>_cam.txt, the intrinsics scaled to the
    quarter-resolution 1152x864 grid the reference hard-codes (x 1/4, x 1152/1600, x 864/1200)."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape((4, 4))
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape((3, 3))
    intrinsics[:2, :] /= 4
    intrinsics[0] *= 1152 / 1600
    intrinsics[1] *= 864 / 1200
    return intrinsics, extrinsics


def read_img(filename):
    """eval.py:77-82: the image resized to 1152x864 (PIL bilinear), float32 in 0..1."""
    from PIL import Image
    img = Image.open(filename).resize((1152, 864), Image.BILINEAR)
    return np.array(img, dtype=np.float32) / 255.


def save_mask(filename, mask):
    """eval.py:90-93: a boolean mask as an 8-bit 0 / 255 image."""
    from PIL import Image
    mask = np.asarray(mask)
    assert mask.dtype == np.bool_
    Image.fromarray(mask.astype(np.uint8) * 255).save(filename)


def read_pair_file(filename):
    """eval.py:97-107: [(ref_view, [src_view, ...]), ...] of a pair.txt."""
    data = []
    with open(filename) as f:
        num_viewpoint = int(f.readline())
        for _ in range(num_viewpoint):
            ref_view = int(f.readline().rstrip())
            src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
            data.append((ref_view, src_views))
    return data


def vertex_ply_bytes(xyz, rgb):
    """The bytes plyfile's PlyData([PlyElement.describe(vertex_all, 'vertex')]).write() produces for the reference's vertex table
    (eval.py:438-447): binary little-endian, float x y z + uchar red green blue.  (plyfile is not part of this image: the header
    text is written from its documented format -- the TABLE is pinned by tests/golden/g12, the container is not.)"""
    xyz, rgb = np.asarray(xyz, dtype="<f4"), np.asarray(rgb, dtype=np.uint8)
    n = xyz.shape[0]
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode("ascii")
    rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["red"], rec["green"], rec["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    return hdr + rec.tobytes()


def filter_depth(scan_folder, out_folder, plyfilename, device=None, photo_thresh=0.8, min_views=3, verbose=True):
    """eval.py:340-447.  For every reference view of <scan_folder>/pair.txt: photometric mask (confidence > 0.8), geometric mask
    (>= 3 source views reproject consistently: ONE HIP launch per reference view, csrc/geo_filter.hip), masks written to
    <out_folder>/mask/, the surviving pixels back-projected with the averaged depth into world coordinates and coloured from the
    image; all views' points written to `plyfilename`.  Returns (xyz float32 [n,3], rgb uint8 [n,3])."""
    from ..datasets.data_io import read_pfm
    lib = _lib.get()
    dev = torch.device(device if device is not None else ("cuda" if lib.device_type == "cuda" else "cpu"))
    pair_data = read_pair_file(os.path.join(scan_folder, "pair.txt"))
    cams, depths = {}, {}

    def cam(v):
        if v not in cams:
            cams[v] = read_camera_parameters(os.path.join(scan_folder, "cams/{:0>8}_cam.txt".format(v)))
        return cams[v]

    def depth_of(v):
        if v not in depths:   # every view's map goes to the device once, not once per pair it appears in
            d = np.ascontiguousarray(read_pfm(os.path.join(out_folder, "depth_est/{:0>8}.pfm".format(v)))[0], dtype=np.float32)
            depths[v] = torch.from_numpy(d).to(dev)
        return depths[v]

    vertexs, vertex_colors = [], []
    os.makedirs(os.path.join(out_folder, "mask"), exist_ok=True)
    for ref_view, src_views in pair_data:
        ref_intrinsics, ref_extrinsics = cam(ref_view)
        ref_img = read_img(os.path.join(scan_folder, "images/{:0>8}.jpg".format(ref_view)))
        confidence = np.ascontiguousarray(read_pfm(os.path.join(out_folder, "confidence/{:0>8}.pfm".format(ref_view)))[0], dtype=np.float32)
        res = filter_depth_view(depth_of(ref_view), torch.from_numpy(confidence).to(dev), ref_intrinsics, ref_extrinsics,
                                [depth_of(v) for v in src_views], [cam(v)[0] for v in src_views], [cam(v)[1] for v in src_views],
                                photo_thresh=photo_thresh, min_views=min_views)
        photo_mask, geo_mask, final_mask = (res[k].cpu().numpy() for k in ("photo_mask", "geo_mask", "final_mask"))
        depth_est_averaged = res["depth_avg"].cpu().numpy()
        save_mask(os.path.join(out_folder, "mask/{:0>8}_photo.png".format(ref_view)), photo_mask)
        save_mask(os.path.join(out_folder, "mask/{:0>8}_geo.png".format(ref_view)), geo_mask)
        save_mask(os.path.join(out_folder, "mask/{:0>8}_final.png".format(ref_view)), final_mask)
        if verbose:
            print("processing {}, ref-view{:0>2}, photo/geo/final-mask:{}/{}/{}".format(scan_folder, ref_view, photo_mask.mean(),
                                                                                        geo_mask.mean(), final_mask.mean()))
        height, width = depth_est_averaged.shape[:2]
        x, y = np.meshgrid(np.arange(0, width), np.arange(0, height))
        valid_points = final_mask
        x, y, depth = x[valid_points], y[valid_points], depth_est_averaged[valid_points]
        color = ref_img[::4, ::4, :][valid_points]                       # hard-coded for DTU, like the reference (eval.py:424)
        xyz_ref = np.matmul(np.linalg.inv(ref_intrinsics), np.vstack((x, y, np.ones_like(x))) * depth)
        xyz_world = np.matmul(np.linalg.inv(ref_extrinsics), np.vstack((xyz_ref, np.ones_like(x))))[:3]
        vertexs.append(xyz_world.transpose((1, 0)))
        vertex_colors.append((color * 255).astype(np.uint8))
    xyz = np.concatenate(vertexs, axis=0).astype(np.float32)
    rgb = np.concatenate(vertex_colors, axis=0)
    with open(plyfilename, "wb") as fh:
        fh.write(vertex_ply_bytes(xyz, rgb))
    if verbose:
        print("saving the final model to", plyfilename)
    return xyz, rgb
