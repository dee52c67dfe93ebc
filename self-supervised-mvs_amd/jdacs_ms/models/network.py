"""Drop-in for jdacs-ms/models/network.py: CVPMVSNet(args).forward(ref_img, src_imgs, ref_in, src_in,
ref_ex, src_ex, depth_min, depth_max) -> {"depth_est_list": [finest..coarsest], "prob_confidence"}
with identical state_dict names.  Cost volumes, the shared 3-D regulariser and soft-argmin run in HIP."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from .modules import (ALIGN_CORNERS, ConvBnReLU3D, DeconvBnReLU3D, ProbConv3d, _ms_proj, calDepthHypo,
                      calSweepingDepthHypo, conditionIntrinsics, conv, proj_cost)


_PYRAMID_LAYERS = (("conv0aa", 3, 64), ("conv0ba", 64, 64), ("conv0bb", 64, 64), ("conv0bc", 64, 32), ("conv0bd", 32, 32),
                   ("conv0be", 32, 32), ("conv0bf", 32, 16), ("conv0bg", 16, 16), ("conv0bh", 16, 16))
_CVP_ENCODER = (("conv0", 16, 16, 1), ("conv0a", 16, 16, 1), ("conv1", 16, 32, 2), ("conv2", 32, 32, 1), ("conv2a", 32, 32, 1),
                ("conv3", 32, 64, 1), ("conv4", 64, 64, 1), ("conv4a", 64, 64, 1))


class FeaturePyramid(nn.Module):
    """network.py:16-41 -- 9 conv+LeakyReLU blocks, shared over a x0.5 image pyramid (stock PyTorch)."""

    def __init__(self):
        super().__init__()
        for name, cin, cout in _PYRAMID_LAYERS:
            setattr(self, name, conv(cin, cout, kernel_size=3, stride=1))

    # conv + bias + LeakyReLU of every block as ONE csrc/conv2d.hip pass (channels-last) instead of the stock
    # MIOpen convolution + activation.  ON by default: measured at BASELINE configs[3] (N=5, 1152x864, 3 levels) the whole
    # inference goes 47.2 -> 40.4 ms (profiles/r02_run10_*); MVS_HIP_PYRAMID=0 / FeaturePyramid.hip_conv = False restores MIOpen.
    hip_conv = os.environ.get("MVS_HIP_PYRAMID", "1") == "1"

    def _trunk(self, img):
        if self.hip_conv and img.is_cuda and img.dtype == torch.float32:
            x = img.contiguous(memory_format=torch.channels_last)
            for name, *_ in _PYRAMID_LAYERS:
                block = getattr(self, name)
                x = ops.Conv2dLReLUFn.apply(x, block[0].weight, block[0].bias, block[1].negative_slope)
            return x
        for name, *_ in _PYRAMID_LAYERS:
            img = getattr(self, name)(img)
        return img

    def forward(self, img, scales=5):
        levels = [self._trunk(img)]
        for _ in range(scales - 1):
            img = F.interpolate(img, scale_factor=0.5, mode='bilinear', align_corners=None).detach()
            levels.append(self._trunk(img))
        return levels


class CostRegNet(nn.Module):
    """network.py:44-74; one down-sampling, even D/H/W required; weights shared by all pyramid levels.
    conv5 is a STRIDE-1 transposed convolution, conv6 stride 2; both add their skip after the ReLU (network.py:71-72)."""

    def __init__(self):
        super().__init__()
        for name, cin, cout, stride in _CVP_ENCODER:
            setattr(self, name, ConvBnReLU3D(cin, cout, stride=stride, kernel_size=3, pad=1))
        self.conv5 = DeconvBnReLU3D(64, 32, stride=1)
        self.conv6 = DeconvBnReLU3D(32, 16, stride=2)
        self.prob0 = ProbConv3d(16)

    def forward(self, x):
        if x.dim() != 5 or x.shape[1] != 16:
            raise ValueError("CVP CostRegNet expects [B,16,D,H,W], got %s" % (tuple(x.shape),))
        if any(s % 2 for s in x.shape[2:]):
            raise ValueError("CVP CostRegNet needs even D,H,W, got %s" % (tuple(x.shape[2:]),))
        if self.training and x.dtype == torch.float32 and ops.FUSED_REGULARISER:
            names = [name for name, *_ in _CVP_ENCODER]
            blocks = []
            for i, (name, _, _, stride) in enumerate(_CVP_ENCODER):
                m = getattr(self, name)
                blocks.append((m.conv, m.bn, False, stride, i - 1, -1))
            blocks.append((self.conv5[0], self.conv5[1], True, 1, len(names) - 1, names.index("conv2a")))
            blocks.append((self.conv6[0], self.conv6[1], True, 2, len(names), names.index("conv0a")))
            return ops.unet_regulariser(x, blocks, self.prob0).squeeze(1)
        keep = {}
        for name, *_ in _CVP_ENCODER:
            x = getattr(self, name)(x)
            keep[name] = x
        x = self.conv5(x, skip=keep["conv2a"])
        x = self.conv6(x, skip=keep["conv0a"])
        return self.prob0(x).squeeze(1)


class CVPMVSNet(nn.Module):
    batch_views = os.environ.get("MVS_CVP_BATCH_VIEWS", "1") != "0"   # feature pyramid of all views in one batch

    def __init__(self, args, align_corners=ALIGN_CORNERS):
        super().__init__()
        self.featurePyramid = FeaturePyramid()
        self.cost_reg_refine = CostRegNet()
        self.args = args
        self.align_corners = align_corners

    def forward(self, ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max):
        with ops.slot_scope():   # one zero-filled arena for the BatchNorm statistic slots of all three regulariser passes
            return self._forward(ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max)

    def _forward(self, ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max):
        a = self.args
        depth_est_list = []
        # feature pyramids (stock PyTorch)
        if self.batch_views:
            # the pyramid has no batch statistics: all views as ONE batch (per-sample results are the same launches' tiles), so
            # the small pyramid levels fill the chip (a 216x288 level of one view is 243 workgroups on 256 CUs)
            nb = ref_img.shape[0]
            fps = self.featurePyramid(torch.cat([ref_img] + [src_imgs[:, i] for i in range(a.nsrc)], 0), a.nscale)
            ref_fp = [f[:nb] for f in fps]
            src_fps = [[f[(i + 1) * nb:(i + 2) * nb] for f in fps] for i in range(a.nsrc)]
        else:   # network.py:100-105: one pyramid call per view
            ref_fp = self.featurePyramid(ref_img, a.nscale)
            src_fps = [self.featurePyramid(src_imgs[:, i], a.nscale) for i in range(a.nsrc)]
        ref_in_ms = conditionIntrinsics(ref_in, ref_img.shape, [f.shape for f in ref_fp])
        src_in_ms = torch.stack([conditionIntrinsics(src_in[:, i], ref_img.shape, [f.shape for f in src_fps[i]])
                                 for i in range(a.nsrc)]).permute(1, 0, 2, 3, 4)

        # coarse level: 48 fronto-parallel planes, fused warp + variance (alias quirk on, network.py:114-137)
        depth_hypos = calSweepingDepthHypo(ref_in_ms[:, -1], src_in_ms[:, 0, -1], ref_ex, src_ex, depth_min, depth_max)
        with torch.no_grad():   # modules.py:71-80 for all source views: one launch
            rot, trans = ops.relative_projections([_ms_proj(src_in_ms[:, i, -1], src_ex[:, i]) for i in range(a.nsrc)],
                                                  _ms_proj(ref_in_ms[:, -1], ref_ex), like=ref_fp[-1])
        cost_volume = ops.plane_sweep_variance(ref_fp[-1], [fp[-1] for fp in src_fps], rot, trans, depth_hypos,
                                               align_corners=self.align_corners, ms_alias=True)
        cost_reg = self.cost_reg_refine(cost_volume)
        del cost_volume
        depth, conf = ops.softargmin_conf(cost_reg, depth_hypos)
        depth_est_list.append(depth)

        # refine along the pyramid (network.py:153-180)
        for level in range(a.nscale - 2, -1, -1):
            depth_up = F.interpolate(depth[None, :], size=None, scale_factor=2, mode='bilinear',
                                     align_corners=None).squeeze(0)
            depth_hypos = calDepthHypo(a, depth_up, ref_in_ms[:, level], src_in_ms[:, :, level], ref_ex, src_ex,
                                       depth_min, depth_max, level)
            cost_volume = proj_cost(a, ref_fp[level], src_fps, level, ref_in_ms[:, level], src_in_ms[:, :, level],
                                    ref_ex, src_ex[:, :], depth_hypos, align_corners=self.align_corners)
            cost_reg2 = self.cost_reg_refine(cost_volume)
            del cost_volume
            depth, conf = ops.softargmin_conf(cost_reg2, depth_hypos)
            depth_est_list.append(depth)

        depth_est_list.reverse()  # [0] is the finest scale (network.py:195)
        return {"depth_est_list": depth_est_list, "prob_confidence": conf}


def sL1_loss(depth_est, depth_gt, mask):
    """network.py:202-203."""
    return F.smooth_l1_loss(depth_est[mask], depth_gt[mask], reduction='mean')


def MSE_loss(depth_est, depth_gt, mask):
    """network.py:206-207."""
    return F.mse_loss(depth_est[mask], depth_gt[mask], reduction='mean')
