"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference's geometric-consistency filter,
jdacs/eval.py:169-224 (`reproject_with_depth`, `check_geometric_consistency`) and the aggregation lines of
`filter_depth` (eval.py:372-396).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Pinning: tests/golden/make_golden_geo.py EXECUTES the reference's own two functions (their source is read from
/root/reference/jdacs/eval.py at generation time, never stored) and stores inputs + outputs as fixtures; this module
is checked against those fixtures (tests/test_oracle_golden.py).  ONE piece is unpinned: `cv2.remap`
(opencv-python is not installed in the build container and is no part of /root/reference; the reference's
environment pins no version).  `remap_bilinear_cv` restates OpenCV's published algorithm for
cv2.remap(src32f, map_x32f, map_y32f, cv2.INTER_LINEAR) with the default BORDER_CONSTANT 0 (imgproc/src/imgwarp.cpp,
RemapInvoker + remapBilinear): map coordinates are converted to fixed point with INTER_BITS = 5 (cvRound(x * 32),
round half to even), the four weights come from a float32 table ((1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx) and taps
outside the image read the border value 0.  The fixture generator hands the same function to the reference code as
`cv2.remap`, so everything AROUND the look-up is pinned and the look-up itself is "parity unpinned"."""
import numpy as np


def remap_bilinear_cv(src, map_x, map_y, interpolation=1, border_value=0.0):
    """cv2.remap(src float32 [H,W], map_x, map_y float32 [h,w], INTER_LINEAR) -> float32 [h,w]."""
    src = np.asarray(src, dtype=np.float32)
    h, w = src.shape
    mx = np.asarray(map_x, dtype=np.float32)
    my = np.asarray(map_y, dtype=np.float32)
    sxf = np.rint(mx * np.float32(32.0))
    syf = np.rint(my * np.float32(32.0))
    bad = ~(np.abs(sxf) < 1.0e9) | ~(np.abs(syf) < 1.0e9)
    sx = np.where(bad, 0, sxf).astype(np.int64)
    sy = np.where(bad, 0, syf).astype(np.int64)
    x0, y0 = sx >> 5, sy >> 5
    fx = ((sx & 31).astype(np.float32)) * np.float32(1.0 / 32.0)
    fy = ((sy & 31).astype(np.float32)) * np.float32(1.0 / 32.0)
    one = np.float32(1.0)
    w0, w1, w2, w3 = (one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx

    def tap(xx, yy):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok, v, np.float32(border_value)).astype(np.float32)
    out = tap(x0, y0) * w0 + tap(x0 + 1, y0) * w1 + tap(x0, y0 + 1) * w2 + tap(x0 + 1, y0 + 1) * w3
    return np.where(bad, np.float32(border_value), out).astype(np.float32)


def camera_products(K_ref, E_ref, K_src, E_src):
    """The float32 camera products of eval.py:176-205 exactly as numpy forms them there (float32 in, float32 out)."""
    K_ref, E_ref, K_src, E_src = (np.asarray(m, dtype=np.float32) for m in (K_ref, E_ref, K_src, E_src))
    return dict(Kr_inv=np.linalg.inv(K_ref), Kr=K_ref, Trs=np.matmul(E_src, np.linalg.inv(E_ref)), Ks=K_src,
                Ks_inv=np.linalg.inv(K_src), Tsr=np.matmul(E_ref, np.linalg.inv(E_src)))


def reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """eval.py:169-207."""
    m = camera_products(K_ref, E_ref, K_src, E_src)
    h, w = depth_ref.shape
    x_ref, y_ref = np.meshgrid(np.arange(0, w), np.arange(0, h))
    x_ref, y_ref = x_ref.reshape(-1), y_ref.reshape(-1)
    ones = np.ones_like(x_ref)
    xyz_ref = np.matmul(m["Kr_inv"], np.vstack((x_ref, y_ref, ones)) * depth_ref.reshape(-1))
    xyz_src = np.matmul(m["Trs"], np.vstack((xyz_ref, ones)))[:3]
    k = np.matmul(m["Ks"], xyz_src)
    xy_src = k[:2] / k[2:3]
    x_src = xy_src[0].reshape(h, w).astype(np.float32)
    y_src = xy_src[1].reshape(h, w).astype(np.float32)
    sampled = remap_bilinear_cv(depth_src, x_src, y_src)
    xyz_src = np.matmul(m["Ks_inv"], np.vstack((xy_src, ones)) * sampled.reshape(-1))
    xyz_rep = np.matmul(m["Tsr"], np.vstack((xyz_src, ones)))[:3]
    depth_rep = xyz_rep[2].reshape(h, w).astype(np.float32)
    k = np.matmul(m["Kr"], xyz_rep)
    xy = k[:2] / k[2:3]
    return depth_rep, xy[0].reshape(h, w).astype(np.float32), xy[1].reshape(h, w).astype(np.float32), x_src, y_src


def check_geometric_consistency(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, pix_thresh=1, rel_thresh=0.01):
    """eval.py:210-224 -> mask, depth_reprojected (0 where inconsistent), x2d_src, y2d_src."""
    h, w = depth_ref.shape
    x_ref, y_ref = np.meshgrid(np.arange(0, w), np.arange(0, h))
    depth_rep, x_rep, y_rep, x_src, y_src = reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src)
    dist = np.sqrt((x_rep - x_ref) ** 2 + (y_rep - y_ref) ** 2)
    rel = np.abs(depth_rep - depth_ref) / depth_ref
    mask = np.logical_and(dist < pix_thresh, rel < rel_thresh)
    depth_rep[~mask] = 0
    return mask, depth_rep, x_src, y_src


def filter_depth_view(depth_ref, conf_ref, K_ref, E_ref, src_depths, src_Ks, src_Es, photo_thresh=0.8, min_views=3):
    """The per-reference-view body of filter_depth (eval.py:357-396): photometric mask, geometric mask (at least
    `min_views` consistent source views), final mask and the averaged depth map."""
    photo_mask = conf_ref > photo_thresh
    geo_sum = 0
    deps = []
    for d, k, e in zip(src_depths, src_Ks, src_Es):
        m, dr, _, _ = check_geometric_consistency(depth_ref, K_ref, E_ref, d, k, e)
        geo_sum = geo_sum + m.astype(np.int32)
        deps.append(dr)
    depth_avg = (sum(deps) + depth_ref) / (geo_sum + 1)
    geo_mask = geo_sum >= min_views
    return dict(photo_mask=photo_mask, geo_mask=geo_mask, final_mask=np.logical_and(photo_mask, geo_mask), geo_count=geo_sum,
                depth_avg=depth_avg)
