"""Drop-in for the geometric-consistency filter of jdacs/eval.py.

  check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src)
      -> mask, depth_reprojected, x2d_src, y2d_src                                        (eval.py:210-224)
  filter_depth_view(depth_ref, confidence, intrinsics_ref, extrinsics_ref, src_depths, src_intrinsics, src_extrinsics)
      -> photo_mask, geo_mask, final_mask, geo_count, depth_avg                           (eval.py:357-396, per reference view)

Same argument meaning and results as the reference's numpy / cv2 code, but the depth maps are torch tensors on the GPU
(straight from MVSNet.forward) and ALL source views of a reference view are handled by one HIP kernel
(csrc/geo_filter.hip).  The few 3x3 / 4x4 camera products are formed on the host in float32 exactly like the reference
does (np.linalg.inv / np.matmul of float32 arrays).  Reading cameras / pair files and writing .ply stay the reference's."""
import ctypes as C

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
