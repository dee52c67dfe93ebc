"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- numpy restatement of the `fusibile` depth-map fusion program
(/root/reference/jdacs/fusion/fusibile, the GPL CUDA program jdacs/fusion/depthfusion.py:366-386 shells out to).

  fusibile_cameras(P_list)          cameraGeometryUtils.h:194-440 (getCameraParameters with transformP = false, cam_scale = 1):
                                    decompose every 3x4 projection matrix into K, R, C; rebuild P = K_0 [R | t] with the FIRST camera's
                                    K (the program assumes one K for all views, :348), M_inv = P[:, :3]^-1, the camera centre from the
                                    3x3 minors of P (:20-47), f = K_0[0, 0].
  tex2d_linear(t, x, y)             the CUDA texture unit's linear filter with unnormalised coordinates (main.cpp:491,541 create the
                                    textures with cudaFilterModeLinear): xB = x - 0.5, i = floor(xB), a = frac(xB) in 1.8 fixed point
                                    (round to nearest), clamped indices -- the published rule (CUDA C programming guide, "Texture
                                    Fetching / Linear Filtering").
  fuse_view(...)                    kernel `fusibile`, fusibile.cu:138-277, vectorised over the pixels of one reference camera.
  fuse_all(...)                     host loop fusibile.cu:416-421 + copy_point_cloud_to_host :281-320 (points whose three coordinates
                                    are all non-zero, row-major, camera after camera).
  ply_bytes(points, colours)        storePlyFileBinaryPointCloud, displayUtils.h:80-136.

Arithmetic is float32 in the reference's order of operations (numpy float32 ops, no fused multiply-add).

Parity status: **parity unpinned** for the two third-party pieces -- OpenCV's decomposeProjectionMatrix / Mat::inv (OpenCV is no
part of /root/reference and is not installed) and the texture unit's filtering (no CUDA device; the program also needs OpenCV +
CUDA to build: unbuildable here).  Everything else follows the reference line by line; the HIP kernel (csrc/fusibile.hip) is
compared with this module, and both with a synthetic scene whose fused points must lie on the surface that generated it."""
import numpy as np

F = np.float32


def rq3(M):
    """RQ decomposition M = K R with K upper triangular with a POSITIVE diagonal and R a rotation (what OpenCV's
    decomposeProjectionMatrix returns for a finite camera K [R | t])."""
    M = np.asarray(M, dtype=np.float64)
    # QR of the row-reversed transpose: P M = (P R_q^T P)(P Q^T)
    Pm = np.flipud(np.eye(3))
    q, r = np.linalg.qr((Pm @ M).T)
    K = Pm @ r.T @ Pm
    R = Pm @ q.T
    D = np.diag(np.sign(np.diag(K)))
    K, R = K @ D, D @ R
    if np.linalg.det(R) < 0:
        K, R = -K, -R
    return K, R


def camera_center_h(P):
    """getCameraCenter (cameraGeometryUtils.h:20-47): the 3x3 minors of P with alternating signs."""
    P = np.asarray(P, dtype=F)
    d = lambda cols: F(np.linalg.det(P[:, cols].astype(np.float64)))
    return np.array([d([1, 2, 3]), -d([0, 2, 3]), d([0, 1, 3]), -d([0, 1, 2])], dtype=F)


def fusibile_cameras(P_list):
    """-> dict(cams [V,32] float32 as include/mvs_hip.h describes them, f)."""
    Ks, Rs, ts = [], [], []
    for P in P_list:
        P = np.asarray(P, dtype=F)
        K, R = rq3(P[:, :3])
        K = K / K[2, 2]
        # decomposeProjectionMatrix returns the homogeneous centre T (P T = 0); C = T[:3] / T[3]; t = -R C (:326-327)
        _, _, vt = np.linalg.svd(P.astype(np.float64))
        T = vt[-1]
        C = T[:3] / T[3]
        Ks.append(K.astype(F))
        Rs.append(R.astype(F))
        ts.append((-(R @ C)).astype(F))
    K0 = Ks[0]
    cams = np.zeros((len(P_list), 32), dtype=F)
    for i, (R, t) in enumerate(zip(Rs, ts)):
        Rt = np.concatenate([R, t.reshape(3, 1)], axis=1).astype(F)
        P = (K0 @ Rt).astype(F)                                    # transformCamera :126-135 with transform = identity
        Minv = np.linalg.inv(P[:, :3]).astype(F)                   # :384
        Ch = camera_center_h(P)
        C = (Ch / Ch[3])[:3]                                       # :139-142
        cams[i, 0:12] = P.reshape(-1)
        cams[i, 12:21] = Minv.reshape(-1)
        cams[i, 21:24] = P[:, 3]
        cams[i, 24:27] = C
    return {"cams": cams, "f": F(K0[0, 0])}


def tex2d_linear(t, x, y):
    """t [H,W,4] float32; x, y float32 arrays (unnormalised texture coordinates) -> [...,4]."""
    h, w = t.shape[:2]
    xb, yb = (x - F(0.5)).astype(F), (y - F(0.5)).astype(F)
    fx, fy = np.floor(xb), np.floor(yb)
    a = (np.floor((xb - fx) * F(256.0) + F(0.5)) * F(1.0 / 256.0)).astype(F)
    b = (np.floor((yb - fy) * F(256.0) + F(0.5)) * F(1.0 / 256.0)).astype(F)
    i0 = np.clip(fx.astype(np.int64), 0, w - 1); i1 = np.clip(fx.astype(np.int64) + 1, 0, w - 1)
    j0 = np.clip(fy.astype(np.int64), 0, h - 1); j1 = np.clip(fy.astype(np.int64) + 1, 0, h - 1)
    one = F(1.0)
    w00, w10, w01, w11 = (one - a) * (one - b), a * (one - b), (one - a) * b, a * b
    t00, t10, t01, t11 = t[j0, i0], t[j0, i1], t[j1, i0], t[j1, i1]
    return (w00[..., None] * t00 + w10[..., None] * t10 + w01[..., None] * t01 + w11[..., None] * t11).astype(F)


def _get3d(cam, px, py, depth):
    """get3Dpoint_cu, fusibile.cu:57-66 (matvecmul4, config.h:177-188)."""
    Mi = cam[12:21]
    ptx = (depth * px.astype(F) - cam[21]).astype(F)
    pty = (depth * py.astype(F) - cam[22]).astype(F)
    ptz = (depth - cam[23]).astype(F)
    X = ((Mi[0] * ptx + Mi[1] * pty).astype(F) + Mi[2] * ptz).astype(F)
    Y = ((Mi[3] * ptx + Mi[4] * pty).astype(F) + Mi[5] * ptz).astype(F)
    Z = ((Mi[6] * ptx + Mi[7] * pty).astype(F) + Mi[8] * ptz).astype(F)
    return X, Y, Z


def fuse_view(nd, img, cams, subset, ref, f, depth_thresh, normal_thresh, num_consistent, save_texture=True):
    """nd [V,H,W,4], img [V,H,W,4] or None -> out [H,W,12] (coord xyz 0, normal xyz 0, colour xyz 0)."""
    nd = np.asarray(nd, dtype=F)
    V, H, W = nd.shape[:3]
    cams = np.asarray(cams, dtype=F)
    py, px = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    normal = nd[ref]
    depth = normal[..., 3]
    camr = cams[ref]
    Xx, Xy, Xz = _get3d(camr, px, py, depth)
    cX = [Xx.copy(), Xy.copy(), Xz.copy()]
    cn = [normal[..., 0].copy(), normal[..., 1].copy(), normal[..., 2].copy()]
    ct = [img[ref][..., k].astype(F).copy() for k in range(3)] if img is not None else [np.zeros((H, W), F) for _ in range(3)]
    count = np.zeros((H, W), dtype=np.int32)
    with np.errstate(all="ignore"):
        for cur in subset:
            if cur == ref:
                continue
            c = cams[cur]
            tx = (((c[0] * Xx + c[1] * Xy).astype(F) + c[2] * Xz).astype(F) + c[3]).astype(F)
            ty = (((c[4] * Xx + c[5] * Xy).astype(F) + c[6] * Xz).astype(F) + c[7]).astype(F)
            tz = (((c[8] * Xx + c[9] * Xy).astype(F) + c[10] * Xz).astype(F) + c[11]).astype(F)
            ptx, pty = (tx / tz).astype(F), (ty / tz).astype(F)
            inside = (ptx >= 0) & (ptx < F(W)) & (pty >= 0) & (pty < F(H))
            sx = np.where(inside, ptx, F(0)); sy = np.where(inside, pty, F(0))
            ndc = tex2d_linear(nd[cur], (sx + F(0.5)).astype(F), (sy + F(0.5)).astype(F))
            d = (camr[24:27] - c[24:27]).astype(F)
            baseline = np.sqrt((d[0] * d[0] + d[1] * d[1]).astype(F) + d[2] * d[2]).astype(F)
            depth_disp = (F(f) * baseline / tz).astype(F)
            nd_disp = (F(f) * baseline / ndc[..., 3]).astype(F)
            ok = inside & (np.abs(depth_disp - nd_disp) < F(depth_thresh))
            dot = ((ndc[..., 0] * normal[..., 0] + ndc[..., 1] * normal[..., 1]).astype(F) + ndc[..., 2] * normal[..., 2]).astype(F)
            angle = np.arccos(dot).astype(F)
            angle = np.where(np.isnan(angle), F(0), angle)
            ok = ok & (angle < F(normal_thresh))
            tpx, tpy = np.trunc(sx).astype(np.int64), np.trunc(sy).astype(np.int64)
            tX = _get3d(c, tpx, tpy, ndc[..., 3])
            for k in range(3):
                cX[k] = np.where(ok, (cX[k] + tX[k]).astype(F), cX[k])
                cn[k] = np.where(ok, (cn[k] + ndc[..., k]).astype(F), cn[k])
            if save_texture and img is not None:
                tc = tex2d_linear(np.asarray(img[cur], dtype=F), (sx + F(0.5)).astype(F), (sy + F(0.5)).astype(F))
                for k in range(3):
                    ct[k] = np.where(ok, (ct[k] + tc[..., k]).astype(F), ct[k])
            count += ok.astype(np.int32)
        div = (count.astype(F) + F(1.0)).astype(F)
        out = np.zeros((H, W, 12), dtype=F)
        keep = count >= num_consistent
        for k in range(3):
            out[..., k] = np.where(keep, (cX[k] / div).astype(F), F(0))
            out[..., 4 + k] = np.where(keep, (cn[k] / div).astype(F), F(0))
            out[..., 8 + k] = np.where(keep, (ct[k] / div).astype(F), F(0))
    return out, count


def compact(out):
    """copy_point_cloud_to_host (fusibile.cu:281-320): row-major, points whose x, y and z are ALL non-zero."""
    flat = out.reshape(-1, 12)
    keep = (flat[:, 0] != 0) & (flat[:, 1] != 0) & (flat[:, 2] != 0)
    return flat[keep]


def fuse_all(nd, img, cams, f, depth_thresh, normal_thresh, num_consistent, save_texture=True):
    V = nd.shape[0]
    subset = list(range(V))                     # selectViews(..., viewSel = false): every view (main.cpp:719)
    pts = [compact(fuse_view(nd, img, cams, subset, ref, f, depth_thresh, normal_thresh, num_consistent, save_texture)[0])
           for ref in range(V)]
    return np.concatenate(pts, axis=0) if pts else np.zeros((0, 12), F)


def ply_bytes(points):
    """storePlyFileBinaryPointCloud (displayUtils.h:80-136): x y z float32 + red green blue uchar = (char)(int) of colour
    components 2, 1, 0 (OpenCV's b, g, r order); non-finite coordinates become 0."""
    n = points.shape[0]
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode("ascii")
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    xyz = points[:, 0:3].astype(F).copy()
    bad = ~(np.isfinite(xyz).all(axis=1))
    xyz[bad] = 0
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    col = points[:, 8:11].astype(np.float32)
    as_char = lambda v: (np.trunc(v).astype(np.int64) & 0xFF).astype(np.uint8)
    rec["r"], rec["g"], rec["b"] = as_char(col[:, 2]), as_char(col[:, 1]), as_char(col[:, 0])
    return hdr + rec.tobytes()


# ---- a synthetic multi-view scene for the tests (a smooth surface z = f(x, y) seen by a few cameras) ---------------------------
def _rot(ax, ay):
    ax, ay = np.radians(ax), np.radians(ay)
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    return ry @ rx


def scene_surface(X, Y):
    return 600.0 + 0.12 * X + 0.05 * Y + 25.0 * np.sin(X / 70.0) * np.cos(Y / 90.0)


def synthetic_scene(nviews, h, w, seed=0, noisy_view=2, hole_view=3):
    """-> P_list (K [R|t], float32), normals_depths [V,h,w,4] (fake gipuma normals (1,1,1)/sqrt(3) where depth > 0, like
    depthfusion.py:224-241), images [V,h,w,4] (b, g, r, 0) float32, K, E lists.  One view is noisy and one has holes."""
    rng = np.random.RandomState(seed)
    Ps, nds, imgs, Ks, Es = [], [], [], [], []
    for v in range(nviews):
        K = np.array([[0.9 * w, 0, w / 2.0], [0, 0.9 * w, h / 2.0], [0, 0, 1]])
        E = np.eye(4)
        if v:
            s = 1.0 if v % 2 else -1.0
            E[:3, :3] = _rot(s * (1.5 + v), -s * (1.0 + 0.7 * v))
            E[:3, 3] = [s * (25.0 + 9 * v), -s * 8.0 * v, 3.0 * v]
        Ki, Ei = np.linalg.inv(K), np.linalg.inv(E)
        x, y = np.meshgrid(np.arange(w), np.arange(h))
        rays = Ki @ np.vstack((x.reshape(-1), y.reshape(-1), np.ones(h * w)))
        d = np.full(h * w, 600.0)
        for _ in range(25):
            pw = Ei @ np.vstack((rays * d, np.ones(h * w)))
            d = d + (scene_surface(pw[0], pw[1]) - pw[2])
        depth = d.reshape(h, w).astype(F)
        if v == noisy_view:
            depth = (depth * (1.0 + 0.02 * rng.randn(h, w))).astype(F)
        if v == hole_view:
            depth[h // 5:h // 2, w // 3:2 * w // 3] = 0
        n = np.where(depth[..., None] > 0, F(1.0 / 1.732050808), F(0)) * np.ones((1, 1, 3), F)
        nds.append(np.concatenate([n, depth[..., None]], axis=2).astype(F))
        img = np.zeros((h, w, 4), F)
        img[..., :3] = (rng.rand(h, w, 3) * 255).astype(F)
        imgs.append(img)
        Ps.append((K @ E[:3]).astype(F))
        Ks.append(K)
        Es.append(E)
    return Ps, np.stack(nds), np.stack(imgs), Ks, Es
