"""Drop-in for jdacs-ms/models/modules.py (CVP-MVSNet helpers): same public names and argument
meaning, HIP kernels underneath, and no hard-coded ``.cuda()`` (App. A Q13).

  conv, conditionIntrinsics, calSweepingDepthHypo, homo_warping, calDepthHypo, proj_cost,
  ConvBnReLU3D, depth_regression, depth_regression_refine
"""
import torch
import torch.nn as nn

from ... import ops
from ...nn3d import ConvBnReLU3D, DeconvBnReLU3D, ProbConv3d  # noqa: F401

ALIGN_CORNERS = False  # see jdacs/models/module.py


def conv(in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1):
    """modules.py:15-19 (2-D feature pyramid block; stock PyTorch, not on the hot path)."""
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding, dilation, bias=True),
                         nn.LeakyReLU(0.1))


def conditionIntrinsics(intrinsics, img_shape, fp_shapes):
    """modules.py:22-37: scale the first two rows of K by each pyramid level's down-sampling ratio.
    Returns [B, nScale, 3, 3]."""
    outs = []
    for fp_shape in fp_shapes:
        k = intrinsics.clone()
        k[:, :2, :] = k[:, :2, :] / (img_shape[2] / fp_shape[2])
        outs.append(k)
    return torch.stack(outs).permute(1, 0, 2, 3)


def calSweepingDepthHypo(ref_in, src_in, ref_ex, src_ex, depth_min, depth_max, nhypothesis_init=48):
    """modules.py:44-59.  The reference builds the planes with torch.range(dmin, dmax, step), whose
    length depends on fp rounding (47 or 48, App. A Q3); here exactly `nhypothesis_init` planes
    dmin + i*step are generated (identical whenever the reference yields 48).  Like the reference,
    batch item 0's range is used for the whole batch."""
    assert nhypothesis_init % 2 == 0
    b = ref_in.shape[0]
    step = (depth_max[0] - depth_min[0]) / (nhypothesis_init - 1)
    planes = depth_min[0] + step * torch.arange(nhypothesis_init, dtype=torch.float32, device=depth_min.device)
    return planes.unsqueeze(0).repeat(b, 1).to(ref_in.device)


def _ms_proj(intrinsics, extrinsics):
    """[K @ E[:3,:]; 0 0 0 1]  (modules.py:71-75)."""
    top = torch.matmul(intrinsics, extrinsics[:, 0:3, :])
    last = torch.zeros(top.shape[0], 1, 4, dtype=top.dtype, device=top.device)
    last[:, 0, 3] = 1.0
    return torch.cat((top, last), 1)


def _ms_rot_trans(ref_in, src_in, ref_ex, src_ex):
    with torch.no_grad():
        return ops.relative_projection(_ms_proj(src_in, src_ex), _ms_proj(ref_in, ref_ex))


def homo_warping(src_feature, ref_in, src_in, ref_ex, src_ex, depth_hypos, align_corners=None):
    """modules.py:62-104: warp one source feature map [B,C,H,W] into the planes depth_hypos [B,D]."""
    rot, trans = _ms_rot_trans(ref_in, src_in, ref_ex, src_ex)
    ac = ALIGN_CORNERS if align_corners is None else align_corners
    return ops.HomoWarp.apply(src_feature, rot, trans, depth_hypos, ac)


def calDepthHypo(netArgs, ref_depths, ref_intrinsics, src_intrinsics, ref_extrinsics, src_extrinsics,
                 depth_min, depth_max, level):
    """modules.py:107-206: per-level depth hypotheses [B,8,H,W] = upsampled depth + k * interval,
    k=-4..3, where interval is the MEAN over pixels of the depth change that moves the projection
    into source view 0 by one pixel along the epipolar line (fp64 inside, fp32 out; App. A Q4).
    One fused kernel pair (csrc/depth_hypo.hip) instead of the reference's ~40 fp64 elementwise ops and
    H*W batched 2x2 torch.inverse per batch item; only the few 3x3 / 4x4 camera products stay host torch code."""
    with torch.no_grad():
        ki, ks = ref_intrinsics.double(), src_intrinsics[:, 0].double()
        ei, es = ref_extrinsics.double(), src_extrinsics[:, 0].double()
        inv = lambda m: torch.linalg.inv_ex(m).inverse            # same LU as torch.inverse, no device sync
        T = ks @ (es @ inv(ei))[:, :3, :]                          # src pixel (homogeneous) of a ref camera-frame point
        A = (ki @ ei[:, :3, :3]) @ inv(ks @ es[:, :3, :3])
        nb = ref_depths.shape[0]
        mats = torch.cat([inv(ki).reshape(nb, 9), T.reshape(nb, 12), A.reshape(nb, 9)], 1)
        return ops.depth_hypotheses(ref_depths.float(), mats)


def proj_cost(settings, ref_feature, src_feature, level, ref_in, src_in, ref_ex, src_ex, depth_hypos,
              align_corners=None):
    """modules.py:209-261: refine-level cost volume with PER-PIXEL hypotheses depth_hypos [B,D,H,W];
    src_feature is the list (per source) of lists (per level) of feature maps.  Includes the
    reference's in-place alias quirk (both running sums start from ref^2, App. A Q2).  One fused kernel."""
    nsrc = settings.nsrc
    with torch.no_grad():   # modules.py:71-80 for all source views: one launch
        rot, trans = ops.relative_projections([_ms_proj(src_in[:, s], src_ex[:, s]) for s in range(nsrc)], _ms_proj(ref_in, ref_ex), like=ref_feature)
    ac = ALIGN_CORNERS if align_corners is None else align_corners
    return ops.plane_sweep_variance(ref_feature, [src_feature[s][level] for s in range(nsrc)], rot, trans,
                                    depth_hypos, align_corners=ac, ms_alias=True)


def depth_regression(p, depth_values):
    """modules.py:324-327."""
    depth_values = depth_values.view(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)


def depth_regression_refine(prob_volume, depth_hypothesis):
    """modules.py:330-331."""
    return torch.sum(prob_volume * depth_hypothesis, 1)
