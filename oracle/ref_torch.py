"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- CPU restatement of the reference hot path.

Plain PyTorch fp32 ops, device agnostic, no ``.cuda()``.  Every function cites the reference
file:line (relative to /root/reference) it follows.  The bilinear sampler is written out by hand
(explicit floor / 4-tap gather / zero padding) instead of calling ``F.grid_sample`` so that the
sampling convention (App. A Q1 of SURVEY.md) is explicit; ``tests/test_oracle_vs_reference.py``
checks it against the reference's ``F.grid_sample`` call.

Parity status: pinned by fixtures generated from the imported reference (tests/golden/), because
the reference itself holds no golden vectors for this path.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------------
def relative_projection(src_proj: torch.Tensor, ref_proj: torch.Tensor):
    """rot [B,3,3], trans [B,3] of P_src * P_ref^-1  (jdacs/models/module.py:116-118)."""
    proj = torch.matmul(src_proj, torch.inverse(ref_proj))
    return proj[:, :3, :3].contiguous(), proj[:, :3, 3].contiguous()


def ms_projection(intrinsics: torch.Tensor, extrinsics: torch.Tensor) -> torch.Tensor:
    """[K*E[:3,:]; 0 0 0 1]  (jdacs-ms/models/modules.py:71-75, 221-226)."""
    top = torch.matmul(intrinsics, extrinsics[:, 0:3, :])
    last = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]], dtype=top.dtype, device=top.device)
    return torch.cat((top, last.repeat(top.shape[0], 1, 1)), 1)


def warp_pixel_coords(rot, trans, depth, height: int, width: int):
    """Source-view pixel coordinates of every (plane, ref pixel).

    rot [B,3,3], trans [B,3]; depth [B,D] (one value per plane, module.py:126) or [B,D,H,W]
    (per-pixel hypotheses, jdacs-ms/models/modules.py:239-240).  Returns px, py [B,D,H*W]
    (module.py:120-130: rot@(x,y,1) * d + t, then x/z, y/z -- no guard on z, App. A Q14).
    """
    b = rot.shape[0]
    dev = rot.device
    ys = torch.arange(0, height, dtype=rot.dtype, device=dev)  # fp32 in the reference; fp64 for truth tests
    xs = torch.arange(0, width, dtype=rot.dtype, device=dev)
    y = ys.view(height, 1).expand(height, width).reshape(-1)
    x = xs.view(1, width).expand(height, width).reshape(-1)
    xyz = torch.stack((x, y, torch.ones_like(x))).unsqueeze(0).expand(b, 3, -1)
    rot_xyz = torch.matmul(rot, xyz)  # [B,3,HW]
    nd = depth.shape[1]
    if depth.dim() == 2:
        dd = depth.view(b, 1, nd, 1)
    else:
        dd = depth.reshape(b, 1, nd, height * width)
    proj_xyz = rot_xyz.unsqueeze(2) * dd + trans.view(b, 3, 1, 1)
    px = proj_xyz[:, 0] / proj_xyz[:, 2]
    py = proj_xyz[:, 1] / proj_xyz[:, 2]
    return px, py


def to_sample_index(p: torch.Tensor, size: int, align_corners: bool) -> torch.Tensor:
    """Reference normalises with the align_corners=True formula (module.py:131-132) and then calls
    F.grid_sample with the *default* align_corners, which is False on torch >= 1.3 (App. A Q1).
    align_corners=False: ix = ((g + 1) * size - 1) / 2 with g = p / ((size-1)/2) - 1."""
    g = p / ((size - 1) / 2) - 1
    if align_corners:
        return (g + 1) / 2 * (size - 1)
    return ((g + 1) * size - 1) / 2


def bilinear_gather_zeros(fea: torch.Tensor, ix: torch.Tensor, iy: torch.Tensor) -> torch.Tensor:
    """4-tap bilinear sample with zero padding.  fea [B,C,H,W]; ix, iy [B,P] -> [B,C,P].
    Same arithmetic as ATen's CPU grid_sampler_2d (bilinear, zeros): w = ix - floor(ix), e = 1 - w."""
    b, c, h, w = fea.shape
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    wx = ix - x0
    ex = 1 - wx
    wy = iy - y0
    ey = 1 - wy
    flat = fea.reshape(b, c, h * w)
    out = torch.zeros(b, c, ix.shape[1], dtype=fea.dtype, device=fea.device)
    for dy, dx, wgt in ((0, 0, ey * ex), (0, 1, ey * wx), (1, 0, wy * ex), (1, 1, wy * wx)):
        xx = x0 + dx
        yy = y0 + dy
        ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
        idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).long()
        idx = torch.where(ok, idx, torch.zeros_like(idx))
        val = torch.gather(flat, 2, idx.unsqueeze(1).expand(b, c, -1))
        out = out + val * (wgt * ok.to(fea.dtype)).unsqueeze(1)
    return out


# Which bilinear sampler warp_features() uses.  "gather": the hand-written 4-tap gather above (the sampling convention spelled
# out; what the parity tests compare against).  "aten": F.grid_sample called exactly as the reference does
# (jdacs/models/module.py:131-136) -- the form bench.py TIMES as the CPU baseline / reference GPU path, so the baseline is the
# reference's own op and not a slower restatement (tests/test_oracle_golden.py asserts the two agree to 1e-6).
SAMPLER = "gather"


def set_sampler(name: str) -> str:
    """Select the sampler ("gather" | "aten"); returns the previous one."""
    global SAMPLER
    if name not in ("gather", "aten"):
        raise ValueError("sampler must be 'gather' or 'aten', got %r" % (name,))
    old, SAMPLER = SAMPLER, name
    return old


def grid_sample_aten(fea: torch.Tensor, px: torch.Tensor, py: torch.Tensor, nd: int, align_corners: bool) -> torch.Tensor:
    """The reference's own sampler call (module.py:131-136): normalise with the (size-1)/2 formula, stack to
    [B, D, H*W, 2], F.grid_sample(bilinear, zeros) on the [B, D*H, W, 2] view.  px, py [B,D,H*W] -> [B,C,D*H*W]."""
    b, c, h, w = fea.shape
    gx = px / ((w - 1) / 2) - 1
    gy = py / ((h - 1) / 2) - 1
    grid = torch.stack((gx, gy), dim=3)
    out = F.grid_sample(fea, grid.view(b, nd * h, w, 2), mode="bilinear", padding_mode="zeros", align_corners=align_corners)
    return out.view(b, c, nd * h * w)


def warp_features(src_fea, rot, trans, depth, align_corners: bool = False, sampler: Optional[str] = None) -> torch.Tensor:
    """[B,C,H,W] -> [B,C,D,H,W]; gradient flows to src_fea only (grid is built under no_grad,
    module.py:115).  sampler: None = the module-level SAMPLER."""
    b, c, h, w = src_fea.shape
    nd = depth.shape[1]
    with torch.no_grad():
        px, py = warp_pixel_coords(rot, trans, depth, h, w)
    if (sampler or SAMPLER) == "aten":
        return grid_sample_aten(src_fea, px, py, nd, align_corners).view(b, c, nd, h, w)
    with torch.no_grad():
        ix = to_sample_index(px, w, align_corners).reshape(b, -1)
        iy = to_sample_index(py, h, align_corners).reshape(b, -1)
    return bilinear_gather_zeros(src_fea, ix, iy).view(b, c, nd, h, w)


def homo_warping(src_fea, src_proj, ref_proj, depth_values, align_corners: bool = False):
    """jdacs/models/module.py:105-140."""
    with torch.no_grad():
        rot, trans = relative_projection(src_proj, ref_proj)
    return warp_features(src_fea, rot, trans, depth_values, align_corners)


def homo_warping_ms(src_feature, ref_in, src_in, ref_ex, src_ex, depth_hypos, align_corners: bool = False):
    """jdacs-ms/models/modules.py:62-104."""
    with torch.no_grad():
        rot, trans = relative_projection(ms_projection(src_in, src_ex), ms_projection(ref_in, ref_ex))
    return warp_features(src_feature, rot, trans, depth_hypos, align_corners)


def variance_from_warped(ref_fea, warped: Sequence[torch.Tensor], ms_alias: bool = False):
    """Variance aggregation.  ms_alias=False: jdacs/models/mvsnet.py:120-136
    (S = r + sum w, Q = r^2 + sum w^2).  ms_alias=True: jdacs-ms in-place alias quirk
    (network.py:114-116, modules.py:216-217): both running sums start from r^2 (App. A Q2)."""
    nd = warped[0].shape[2]
    n = len(warped) + 1
    r = ref_fea.unsqueeze(2).expand(-1, -1, nd, -1, -1)
    q = r * r
    s = q if ms_alias else r
    for wv in warped:
        s = s + wv
        q = q + wv * wv
    sm = s / n
    return q / n - sm * sm


def plane_sweep_variance(ref_fea, src_feas, rots, transs, depth, ms_alias=False, align_corners=False):
    """Fused statement of A1+A2 (or A4/A5): what the HIP kernel K1 computes in one pass."""
    warped = [warp_features(f, r, t, depth, align_corners) for f, r, t in zip(src_feas, rots, transs)]
    return variance_from_warped(ref_fea, warped, ms_alias)


# --------------------------------------------------------------------------------------------
# soft-argmin + confidence
# --------------------------------------------------------------------------------------------
def depth_regression(p, depth_values):
    """module.py:145-148 (depth_values [B,D] or [D]) / modules.py:330-331 ([B,D,H,W])."""
    if depth_values.dim() <= 2:
        depth_values = depth_values.view(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)


def photometric_confidence(prob_volume):
    """mvsnet.py:145-151 / network.py:183-189: sum of p over [idx-1, idx+2] (zero padded),
    idx = trunc(sum_d p_d * d)  (App. A Q7)."""
    with torch.no_grad():
        nd = prob_volume.shape[1]
        sum4 = 4 * F.avg_pool3d(F.pad(prob_volume.unsqueeze(1), pad=(0, 0, 0, 0, 1, 2)), (4, 1, 1),
                                stride=1, padding=0).squeeze(1)
        idx = depth_regression(prob_volume, torch.arange(nd, device=prob_volume.device,
                                                         dtype=torch.float)).long()
        return torch.gather(sum4, 1, idx.unsqueeze(1)).squeeze(1)


def softargmin_conf(logits, depth_values):
    """logits [B,D,H,W] -> depth, confidence, prob_volume (mvsnet.py:141-151)."""
    p = F.softmax(logits, dim=1)
    return depth_regression(p, depth_values), photometric_confidence(p), p


# --------------------------------------------------------------------------------------------
# networks (same parameter names / shapes as the reference so its state_dict loads unchanged)
# --------------------------------------------------------------------------------------------
class _CBR(nn.Module):
    """conv (no bias) -> BN -> ReLU; keys ``conv.*``/``bn.*`` (module.py:15-22, 35-42)."""

    def __init__(self, dims, cin, cout, k=3, stride=1, pad=1):
        super().__init__()
        conv_t, bn_t = (nn.Conv2d, nn.BatchNorm2d) if dims == 2 else (nn.Conv3d, nn.BatchNorm3d)
        self.conv = conv_t(cin, cout, k, stride=stride, padding=pad, bias=False)
        self.bn = bn_t(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


def _deconv3(cin, cout, stride, out_pad):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=out_pad,
                                            stride=stride, bias=False), nn.BatchNorm3d(cout), nn.ReLU())


class OracleFeatureNet(nn.Module):
    """jdacs/models/mvsnet.py:17-34."""

    def __init__(self):
        super().__init__()
        spec = [(3, 8, 3, 1, 1), (8, 8, 3, 1, 1), (8, 16, 5, 2, 2), (16, 16, 3, 1, 1), (16, 16, 3, 1, 1),
                (16, 32, 5, 2, 2), (32, 32, 3, 1, 1)]
        for i, (ci, co, k, s, p) in enumerate(spec):
            setattr(self, "conv%d" % i, _CBR(2, ci, co, k, s, p))
        self.feature = nn.Conv2d(32, 32, 3, 1, 1)

    def forward(self, x):
        for i in range(7):
            x = getattr(self, "conv%d" % i)(x)
        return self.feature(x)


class OracleCostRegNet(nn.Module):
    """jdacs/models/mvsnet.py:37-74 (skip adds come after the ReLU, App. A Q12)."""

    def __init__(self):
        super().__init__()
        for name, ci, co, s in (("conv0", 32, 8, 1), ("conv1", 8, 16, 2), ("conv2", 16, 16, 1),
                                ("conv3", 16, 32, 2), ("conv4", 32, 32, 1), ("conv5", 32, 64, 2),
                                ("conv6", 64, 64, 1)):
            setattr(self, name, _CBR(3, ci, co, 3, s, 1))
        self.conv7 = _deconv3(64, 32, 2, 1)
        self.conv9 = _deconv3(32, 16, 2, 1)
        self.conv11 = _deconv3(16, 8, 2, 1)
        self.prob = nn.Conv3d(8, 1, 3, stride=1, padding=1)

    def forward(self, x):
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        x = self.conv6(self.conv5(c4))
        x = c4 + self.conv7(x)
        x = c2 + self.conv9(x)
        x = c0 + self.conv11(x)
        return self.prob(x)


class OracleRefineNet(nn.Module):
    """jdacs/models/mvsnet.py:77-92."""

    def __init__(self):
        super().__init__()
        self.conv1 = _CBR(2, 4, 32)
        self.conv2 = _CBR(2, 32, 32)
        self.conv3 = _CBR(2, 32, 32)
        self.res = _CBR(2, 32, 1)

    def forward(self, img, depth_init):
        img = F.interpolate(img, scale_factor=0.25, mode="bilinear")
        d = depth_init.unsqueeze(1)
        return (d + self.res(self.conv3(self.conv2(self.conv1(torch.cat((img, d), 1)))))).squeeze(1)


class OracleMVSNet(nn.Module):
    """jdacs/models/mvsnet.py:95-161."""

    def __init__(self, refine=True, align_corners=False):
        super().__init__()
        self.refine = refine
        self.align_corners = align_corners
        self.feature = OracleFeatureNet()
        self.cost_regularization = OracleCostRegNet()
        if refine:
            self.refine_network = OracleRefineNet()

    def forward(self, imgs, proj_matrices, depth_values, return_intermediates=False):
        views = torch.unbind(imgs, 1)
        projs = torch.unbind(proj_matrices, 1)
        assert len(views) == len(projs), "Different number of images and projection matrices"
        feats = [self.feature(v) for v in views]
        warped = [homo_warping(f, p, projs[0], depth_values, self.align_corners)
                  for f, p in zip(feats[1:], projs[1:])]
        var = variance_from_warped(feats[0], warped, ms_alias=False)
        logits = self.cost_regularization(var).squeeze(1)
        depth, conf, prob = softargmin_conf(logits, depth_values)
        if self.refine:
            depth = self.refine_network(views[0], depth)
        out = {"depth": depth, "photometric_confidence": conf}
        if return_intermediates:
            out.update(variance=var, logits=logits, prob_volume=prob, features=feats)
        return out


def mvsnet_loss(depth_est, depth_gt, mask):
    """jdacs/models/mvsnet.py:164-166."""
    mask = mask > 0.5
    return F.smooth_l1_loss(depth_est[mask], depth_gt[mask], reduction="mean")


def abs_depth_error(depth_est, depth_gt, mask=None):
    """AbsDepthError_metrics, jdacs/utils.py:159-163 (the 'abs-depth L1' of BASELINE.json)."""
    if mask is None:
        mask = torch.ones_like(depth_gt, dtype=torch.bool)
    return torch.mean((depth_est[mask] - depth_gt[mask]).abs())


# ---- CVP-MVSNet (jdacs-ms) ------------------------------------------------------------------
def _conv_lrelu(cin, cout):
    """jdacs-ms/models/modules.py:15-19."""
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, 1, bias=True), nn.LeakyReLU(0.1))


class OracleFeaturePyramid(nn.Module):
    """jdacs-ms/models/network.py:16-41 (shared weights applied to a x0.5 image pyramid)."""
    _names = ("conv0aa", "conv0ba", "conv0bb", "conv0bc", "conv0bd", "conv0be", "conv0bf", "conv0bg", "conv0bh")
    _chan = (3, 64, 64, 64, 32, 32, 32, 16, 16, 16)

    def __init__(self):
        super().__init__()
        for i, n in enumerate(self._names):
            setattr(self, n, _conv_lrelu(self._chan[i], self._chan[i + 1]))

    def _trunk(self, img):
        for n in self._names:
            img = getattr(self, n)(img)
        return img

    def forward(self, img, scales=5):
        fp = [self._trunk(img)]
        for _ in range(scales - 1):
            img = F.interpolate(img, scale_factor=0.5, mode="bilinear", align_corners=None).detach()
            fp.append(self._trunk(img))
        return fp


class OracleCostRegNetMS(nn.Module):
    """jdacs-ms/models/network.py:44-74."""

    def __init__(self):
        super().__init__()
        for name, ci, co, s in (("conv0", 16, 16, 1), ("conv0a", 16, 16, 1), ("conv1", 16, 32, 2),
                                ("conv2", 32, 32, 1), ("conv2a", 32, 32, 1), ("conv3", 32, 64, 1),
                                ("conv4", 64, 64, 1), ("conv4a", 64, 64, 1)):
            setattr(self, name, _CBR(3, ci, co, 3, s, 1))
        self.conv5 = _deconv3(64, 32, 1, 0)
        self.conv6 = _deconv3(32, 16, 2, 1)
        self.prob0 = nn.Conv3d(16, 1, 3, stride=1, padding=1)

    def forward(self, x):
        c0 = self.conv0a(self.conv0(x))
        c2 = self.conv2a(self.conv2(self.conv1(c0)))
        c4 = self.conv4a(self.conv4(self.conv3(c2)))
        c5 = c2 + self.conv5(c4)
        c6 = c0 + self.conv6(c5)
        return self.prob0(c6).squeeze(1)


def condition_intrinsics(intrinsics, img_shape, fp_shapes):
    """jdacs-ms/models/modules.py:22-37."""
    outs = []
    for fs in fp_shapes:
        ratio = img_shape[2] / fs[2]
        k = intrinsics.clone()
        k[:, :2, :] = k[:, :2, :] / ratio
        outs.append(k)
    return torch.stack(outs).permute(1, 0, 2, 3)


def sweeping_depth_hypos(depth_min, depth_max, batch, nhyp=48):
    """jdacs-ms/models/modules.py:44-59, with the plane count made exact (App. A Q3): the
    reference uses torch.range(dmin, dmax, step) whose length depends on fp rounding; parity
    inputs use an exactly representable step so both give `nhyp` planes dmin + i*step."""
    step = (depth_max[0] - depth_min[0]) / (nhyp - 1)
    h = depth_min[0] + step * torch.arange(nhyp, dtype=torch.float32, device=depth_min.device)
    return h.unsqueeze(0).repeat(batch, 1)


def cal_depth_hypo(ref_depths, ref_in, src_in, ref_ex, src_ex, d=4, pixel_interval=1):
    """jdacs-ms/models/modules.py:107-206.  fp64 inside, source view 0 only, per-pixel interval
    collapsed to its mean (App. A Q4).  ref_depths [B,H,W]; src_in [B,nsrc,3,3]; src_ex [B,nsrc,4,4].
    Returns [B,2d,H,W] fp32."""
    nb, h, w = ref_depths.shape
    dev = ref_depths.device
    with torch.no_grad():
        ki = ref_in.double()
        ks = src_in[:, 0].double()
        ei = ref_ex.double()
        es = src_ex[:, 0].double()
        hyp = ref_depths.unsqueeze(1).repeat(1, 2 * d, 1, 1).double()
        for b in range(nb):
            # x-major pixel order (meshgrid([W],[H]) in the reference, modules.py:130-138)
            xx = torch.arange(0, w, device=dev).view(w, 1).expand(w, h).reshape(-1).double()
            yy = torch.arange(0, h, device=dev).view(1, h).expand(w, h).reshape(-1).double()
            X = torch.stack([xx, yy, torch.ones_like(xx)], 0)
            D1 = ref_depths[b].t().reshape(-1).double()
            D2 = D1 + 1
            one = torch.ones_like(xx).unsqueeze(0)

            def to_src(Dz):
                ray = torch.matmul(torch.inverse(ki[b]), X * Dz)
                wpt = torch.matmul(torch.inverse(ei[b]), torch.cat([ray, one], 0))
                cam = torch.matmul(es[b], wpt)[:3]
                pix = torch.matmul(ks[b], cam)
                z = pix[2].clone()
                return pix / z, z

            X1, X1_d = to_src(D1)
            X2, _ = to_src(D2)
            k = (X2[1] - X1[1]) / (X2[0] - X1[0])
            theta = torch.atan(k)
            X3 = X1 + torch.stack([torch.cos(theta) * pixel_interval, torch.sin(theta) * pixel_interval,
                                   torch.zeros_like(X1[2])], 0)
            A = torch.matmul(ki[b], ei[b][:3, :3])
            A = torch.matmul(A, torch.inverse(torch.matmul(ks[b], es[b][:3, :3])))
            t1 = X1_d * torch.matmul(A, X1)
            t2 = torch.matmul(A, X3)
            M1 = torch.cat([X.t().unsqueeze(2), t2.t().unsqueeze(2)], 2)[:, 1:, :]
            M2 = t1.t()[:, 1:]
            ans = torch.matmul(torch.inverse(M1), M2.unsqueeze(2))
            interval = torch.abs(ans[:, 0, 0]).mean()
            for lv in range(-d, d):
                hyp[b, lv + d] += lv * interval
        return hyp.float()


class OracleCVPMVSNet(nn.Module):
    """jdacs-ms/models/network.py:77-199."""

    def __init__(self, args, align_corners=False):
        super().__init__()
        self.featurePyramid = OracleFeaturePyramid()
        self.cost_reg_refine = OracleCostRegNetMS()
        self.args = args
        self.align_corners = align_corners

    def forward(self, ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max,
                return_intermediates=False):
        a = self.args
        ref_fp = self.featurePyramid(ref_img, a.nscale)
        src_fps = [self.featurePyramid(src_imgs[:, i], a.nscale) for i in range(a.nsrc)]
        ref_in_ms = condition_intrinsics(ref_in, ref_img.shape, [f.shape for f in ref_fp])
        src_in_ms = torch.stack([condition_intrinsics(src_in[:, i], ref_img.shape, [f.shape for f in src_fps[i]])
                                 for i in range(a.nsrc)]).permute(1, 0, 2, 3, 4)
        hypos = sweeping_depth_hypos(depth_min, depth_max, ref_img.shape[0])
        inter = {}
        warped = [homo_warping_ms(src_fps[i][-1], ref_in_ms[:, -1], src_in_ms[:, i, -1], ref_ex, src_ex[:, i],
                                  hypos, self.align_corners) for i in range(a.nsrc)]
        cost = variance_from_warped(ref_fp[-1], warped, ms_alias=True)
        logits = self.cost_reg_refine(cost)
        inter["cost_coarse"], inter["logits_coarse"] = cost, logits
        prob = F.softmax(logits, dim=1)
        depth = depth_regression(prob, hypos)
        ests = [depth]
        for level in range(a.nscale - 2, -1, -1):
            up = F.interpolate(depth[None, :], size=None, scale_factor=2, mode="bilinear",
                               align_corners=None).squeeze(0)
            hyp = cal_depth_hypo(up, ref_in_ms[:, level], src_in_ms[:, :, level], ref_ex, src_ex)
            cost = proj_cost(a.nsrc, ref_fp[level], [fp[level] for fp in src_fps], ref_in_ms[:, level],
                             src_in_ms[:, :, level], ref_ex, src_ex, hyp, self.align_corners)
            prob = F.softmax(self.cost_reg_refine(cost), dim=1)
            depth = depth_regression(prob, hyp)
            inter["hypos_l%d" % level], inter["cost_l%d" % level] = hyp, cost
            ests.append(depth)
        conf = photometric_confidence(prob)
        ests.reverse()
        out = {"depth_est_list": ests, "prob_confidence": conf}
        if return_intermediates:
            out["intermediates"] = inter
        return out


def proj_cost(nsrc, ref_feature, src_features, ref_in, src_in, ref_ex, src_ex, depth_hypos,
              align_corners=False):
    """jdacs-ms/models/modules.py:209-261: per-pixel hypotheses [B,D,H,W], alias quirk on.
    src_features: list of nsrc tensors at this level; src_in [B,nsrc,3,3]; src_ex [B,nsrc,4,4]."""
    warped = [homo_warping_ms(src_features[s], ref_in, src_in[:, s], ref_ex, src_ex[:, s], depth_hypos,
                              align_corners) for s in range(nsrc)]
    return variance_from_warped(ref_feature, warped, ms_alias=True)


def cvp_args(nsrc=2, nscale=2, mode="test"):
    return SimpleNamespace(nsrc=nsrc, nscale=nscale, mode=mode)


# --------------------------------------------------------------------------------------------
# synthetic DTU-shaped inputs; used by tests, smoke and bench (CPU leg)
# --------------------------------------------------------------------------------------------
# The generators themselves live in the package (mvs_amd.synthetic) so that bench.py and the measurement tools need not
# import this test-only module for their inputs; re-exported here for the tests.
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import mvs_amd  # noqa: E402,F401  (repo-root alias of self-supervised-mvs_amd/)
from mvs_amd.synthetic import _rot_xy, synthetic_cameras, synthetic_mvsnet_inputs  # noqa: E402,F401


# =============================================================================================
# the self-supervised loss on the path's output (jdacs/losses/unsup_loss.py:19-83).
# TEST INFRASTRUCTURE like everything in this file; pinned by tests/golden/g8_unsup_loss*.npz.
# =============================================================================================
def quarter_image(img):
    """[B,3,H,W] -> [B,H/4,W/4,3]: F.interpolate(scale_factor=0.25, bilinear) + permute (unsup_loss.py:36-37,53-54)."""
    return F.interpolate(img, scale_factor=0.25, mode="bilinear").permute(0, 2, 3, 1)


def unsup_view_transform(ref_cam, view_cam):
    """Per view: K_ref^-1 and the 3x4 matrix K_ref . [R_rel | t_rel] (homography.py:186-236).  The reference projects
    with the REFERENCE intrinsics (`intrinsic_mat_hom` is built from K_left, homography.py:231), kept as is."""
    R_l, t_l = ref_cam[:, 0, :3, :3], ref_cam[:, 0, :3, 3:4]
    R_r, t_r = view_cam[:, 0, :3, :3], view_cam[:, 0, :3, 3:4]
    K_l = ref_cam[:, 1, :3, :3]
    R_rel = R_r @ R_l.transpose(1, 2)
    t_rel = t_r - R_rel @ t_l
    return torch.inverse(K_l), K_l @ torch.cat([R_rel, t_rel], 2)        # [B,3,3], [B,3,4]


def unsup_inverse_warp(view_q, kinv, proj, depth):
    """view_q [B,h,w,3], depth [B,h,w] -> warped [B,h,w,3], mask [B,h,w,1] (homography.py:186-351).
    Quirks kept: z + 1e-10; the validity mask tests x0>=0, x1<=w-1, y0>=0, y0<=h-1 (not y1); the bilinear weights are
    formed with the CLAMPED x1 / y1 (homography.py:294-334)."""
    b, h, w, _ = view_q.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=depth.device),
                            torch.arange(w, dtype=torch.float32, device=depth.device), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(h * w, device=depth.device)], 0)      # [3,hw]
    cam = (kinv @ pix.unsqueeze(0)) * depth.reshape(b, 1, h * w)                                         # [B,3,hw]
    pc = proj[:, :, :3] @ cam + proj[:, :, 3:4]
    x = pc[:, 0] / (pc[:, 2] + 1e-10)
    y = pc[:, 1] / (pc[:, 2] + 1e-10)
    # (the reference normalises to [-1,1] and back, homography.py:270-272,292-293: identity up to rounding)
    x = ((x / (w - 1) * 2.0 - 1.0) + 1.0) * (w - 1.0) / 2.0
    y = ((y / (h - 1) * 2.0 - 1.0) + 1.0) * (h - 1.0) / 2.0
    x0 = torch.floor(x).int()
    y0 = torch.floor(y).int()
    x1, y1 = x0 + 1, y0 + 1
    mask = ((x0 >= 0) & (x1 <= w - 1) & (y0 >= 0) & (y0 <= h - 1)).float()
    x0c, x1c = x0.clamp(0, w - 1), x1.clamp(0, w - 1)
    y0c, y1c = y0.clamp(0, h - 1), y1.clamp(0, h - 1)
    flat = view_q.reshape(b, h * w, 3)

    def take(yy, xx):
        idx = (yy.long() * w + xx.long()).unsqueeze(-1).expand(-1, -1, 3)
        return torch.gather(flat, 1, idx)
    fx, fy = x1c.float() - x, y1c.float() - y
    out = ((fx * fy).unsqueeze(-1) * take(y0c, x0c) + (fx * (1.0 - fy)).unsqueeze(-1) * take(y1c, x0c)
           + ((1.0 - fx) * fy).unsqueeze(-1) * take(y0c, x1c) + ((1.0 - fx) * (1.0 - fy)).unsqueeze(-1) * take(y1c, x1c))
    return out.reshape(b, h, w, 3), mask.reshape(b, h, w, 1)


def unsup_reconstr_term(warped, ref, mask):
    """0.5 smooth-L1 of the masked images + 0.5 smooth-L1 of their forward differences (modules.py:80-90)."""
    wm, rm = warped * mask, ref * mask
    photo = F.smooth_l1_loss(wm, rm)
    gx = F.smooth_l1_loss(wm[:, :, 1:] - wm[:, :, :-1], rm[:, :, 1:] - rm[:, :, :-1])
    gy = F.smooth_l1_loss(wm[:, 1:] - wm[:, :-1], rm[:, 1:] - rm[:, :-1])
    return 0.5 * photo + 0.5 * (gx + gy)


def unsup_ssim_map(x, y, mask):
    """3x3 average-pool SSIM dissimilarity, mask pooled likewise (modules.py:17-52); NHWC in, NHWC out."""
    x, y, mask = x.permute(0, 3, 1, 2), y.permute(0, 3, 1, 2), mask.permute(0, 3, 1, 2)
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sx = F.avg_pool2d(x * x, 3, 1) - mu_x ** 2
    sy = F.avg_pool2d(y * y, 3, 1) - mu_y ** 2
    sxy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + 0.01 ** 2) * (2 * sxy + 0.03 ** 2)
    d = (mu_x ** 2 + mu_y ** 2 + 0.01 ** 2) * (sx + sy + 0.03 ** 2)
    return (F.avg_pool2d(mask, 3, 1) * torch.clamp((1 - n / d) / 2, 0, 1)).permute(0, 2, 3, 1)


def unsup_smoothness(depth, ref, lam):
    """image-aware first-order smoothness (modules.py:55-77): |d(p)-d(p+1)| * exp(-lam * mean_c |I(p)-I(p+1)|)."""
    d = depth.unsqueeze(-1)
    wx = torch.exp(-lam * (ref[:, :, :-1] - ref[:, :, 1:]).abs().mean(3, keepdim=True))
    wy = torch.exp(-lam * (ref[:, :-1] - ref[:, 1:]).abs().mean(3, keepdim=True))
    return ((d[:, :, :-1] - d[:, :, 1:]) * wx).abs().mean() + ((d[:, :-1] - d[:, 1:]) * wy).abs().mean()


def unsup_loss(imgs, cams, depth, smooth_lambda=1.0, return_terms=False):
    """UnSupLoss.forward (unsup_loss.py:24-83): 12 * mean(sum of the 3 smallest valid per-view reconstruction terms)
    + 6 * (SSIM of views 1, 2) + 0.18 * smoothness.  Note the per-view reconstruction term is a SCALAR broadcast
    over the pixels (+1e4 where the view is invalid) before the top-3 selection (unsup_loss.py:61-63,74-81)."""
    n = imgs.shape[1]
    ref = quarter_image(imgs[:, 0])
    vols, ssim = [], 0.0
    for v in range(1, n):
        kinv, proj = unsup_view_transform(cams[:, 0], cams[:, v])
        warped, mask = unsup_inverse_warp(quarter_image(imgs[:, v]), kinv, proj, depth)
        vols.append(unsup_reconstr_term(warped, ref, mask) + 1e4 * (1 - mask))
        if v < 3:
            ssim = ssim + unsup_ssim_map(ref, warped, mask).mean()
    smooth = unsup_smoothness(depth, ref, smooth_lambda)
    vol = torch.stack(vols).permute(1, 2, 3, 4, 0)
    top = -torch.topk(-vol, k=3, sorted=False)[0]
    top = top * (top < 1e4).float()
    reconstr = top.sum(-1).mean()
    total = 12 * reconstr + 6 * ssim + 0.18 * smooth
    return (total, reconstr, ssim, smooth) if return_terms else total
