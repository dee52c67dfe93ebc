"""Drop-in for jdacs/models/mvsnet.py: MVSNet(refine).forward(imgs, proj_matrices, depth_values)
-> {"depth", "photometric_confidence"} with identical state_dict names.

Hot path (HIP): plane-sweep variance volume (mvsnet.py:120-136 + module.py:105-140), CostRegNet
(mvsnet.py:37-74), softmax + depth regression + confidence (mvsnet.py:141-151).
FeatureNet / RefineNet are 2-D CNNs outside the path and stay stock PyTorch (MIOpen)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...nn3d import batched_bn_counters
from .module import ALIGN_CORNERS, ConvBnReLU, ConvBnReLU3D, DeconvBnReLU3D, ProbConv3d


class FeatureNet(nn.Module):
    """mvsnet.py:17-34 -- 3 -> 32 channels at 1/4 resolution."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.conv0 = ConvBnReLU(3, 8, 3, 1, 1)
        self.conv1 = ConvBnReLU(8, 8, 3, 1, 1)
        self.conv2 = ConvBnReLU(8, 16, 5, 2, 2)
        self.conv3 = ConvBnReLU(16, 16, 3, 1, 1)
        self.conv4 = ConvBnReLU(16, 16, 3, 1, 1)
        self.conv5 = ConvBnReLU(16, 32, 5, 2, 2)
        self.conv6 = ConvBnReLU(32, 32, 3, 1, 1)
        self.feature = nn.Conv2d(32, 32, 3, 1, 1)

    def forward(self, x, groups=1):
        """groups: number of views stacked along the batch dim (per-view BatchNorm statistics are kept)."""
        x = self.conv1(self.conv0(x, groups), groups)
        x = self.conv4(self.conv3(self.conv2(x, groups), groups), groups)
        return self.feature(self.conv6(self.conv5(x, groups), groups))


class CostRegNet(nn.Module):
    """mvsnet.py:37-74: 3-D U-Net 32 -> 8 -> 16 -> 32 -> 64 -> ... -> 1; D, H, W divisible by 8."""

    def __init__(self):
        super().__init__()
        self.conv0 = ConvBnReLU3D(32, 8)
        self.conv1 = ConvBnReLU3D(8, 16, stride=2)
        self.conv2 = ConvBnReLU3D(16, 16)
        self.conv3 = ConvBnReLU3D(16, 32, stride=2)
        self.conv4 = ConvBnReLU3D(32, 32)
        self.conv5 = ConvBnReLU3D(32, 64, stride=2)
        self.conv6 = ConvBnReLU3D(64, 64)
        self.conv7 = DeconvBnReLU3D(64, 32, stride=2)
        self.conv9 = DeconvBnReLU3D(32, 16, stride=2)
        self.conv11 = DeconvBnReLU3D(16, 8, stride=2)
        self.prob = ProbConv3d(8)

    def forward(self, x):
        if x.dim() != 5 or x.shape[1] != 32:
            raise ValueError("CostRegNet expects [B,32,D,H,W], got %s" % (tuple(x.shape),))
        if any(s % 8 for s in x.shape[2:]):
            raise ValueError("CostRegNet needs D,H,W divisible by 8, got %s" % (tuple(x.shape[2:]),))
        conv0 = self.conv0(x)
        conv2 = self.conv2(self.conv1(conv0))
        conv4 = self.conv4(self.conv3(conv2))
        x = self.conv6(self.conv5(conv4))
        x = self.conv7(x, skip=conv4)    # conv4 + relu(bn(deconv(x)))   (mvsnet.py:70)
        x = self.conv9(x, skip=conv2)
        x = self.conv11(x, skip=conv0)
        return self.prob(x)


class RefineNet(nn.Module):
    """mvsnet.py:77-92 (off by default in the reference, config.py:48)."""

    def __init__(self):
        super().__init__()
        self.conv1 = ConvBnReLU(4, 32)
        self.conv2 = ConvBnReLU(32, 32)
        self.conv3 = ConvBnReLU(32, 32)
        self.res = ConvBnReLU(32, 1)

    def forward(self, img, depth_init):
        img = F.interpolate(img, scale_factor=0.25, mode='bilinear')
        depth_init = depth_init.unsqueeze(dim=1)
        concat = torch.cat((img, depth_init), dim=1)
        depth_residual = self.res(self.conv3(self.conv2(self.conv1(concat))))
        return (depth_init + depth_residual).squeeze(dim=1)


class MVSNet(nn.Module):
    def __init__(self, refine=True, align_corners=ALIGN_CORNERS, channels_last_features=True):
        super().__init__()
        self.refine = refine
        self.align_corners = align_corners
        # optional: run the (stock PyTorch) 2-D feature extractor in channels-last so MIOpen picks NHWC kernels
        # and the features arrive in the layout the plane-sweep kernel reads (no NCHW->NHWC transpose)
        self.channels_last_features = channels_last_features
        self.feature = FeatureNet()
        self.cost_regularization = CostRegNet()
        if self.refine:
            self.refine_network = RefineNet()

    def forward(self, imgs, proj_matrices, depth_values):
        with batched_bn_counters():
            return self._forward(imgs, proj_matrices, depth_values)

    def _forward(self, imgs, proj_matrices, depth_values):
        imgs = torch.unbind(imgs, 1)
        proj_matrices = torch.unbind(proj_matrices, 1)
        assert len(imgs) == len(proj_matrices), "Different number of images and projection matrices"

        # step 1. feature extraction (stock PyTorch)
        if self.channels_last_features:
            if not getattr(self, "_feature_cl", False):
                self.feature.to(memory_format=torch.channels_last)
                self._feature_cl = True
            # all views through the shared-weight extractor as ONE batch (3x fewer launches, no per-view
            # gradient accumulation); BatchNorm keeps the reference's per-view statistics (grouped BN kernels)
            stacked = torch.cat(imgs, 0).contiguous(memory_format=torch.channels_last)
            features = list(self.feature(stacked, groups=len(imgs)).chunk(len(imgs), 0))
        else:
            features = [self.feature(img) for img in imgs]
        ref_feature, src_features = features[0], features[1:]
        ref_proj, src_projs = proj_matrices[0], proj_matrices[1:]

        # step 2. homography warp + variance cost volume: ONE fused HIP kernel, volume written once
        with torch.no_grad():
            rt = [ops.relative_projection(p, ref_proj) for p in src_projs]
            rot = torch.stack([r for r, _ in rt], 1)
            trans = torch.stack([t for _, t in rt], 1)
        volume_variance = ops.plane_sweep_variance(ref_feature, src_features, rot, trans, depth_values,
                                                   align_corners=self.align_corners, ms_alias=False)

        # step 3. cost volume regularisation (MFMA implicit-GEMM convs)
        cost_reg = self.cost_regularization(volume_variance).squeeze(1)

        # step 4. softmax over depth + soft-argmin + photometric confidence (one kernel)
        depth, photometric_confidence = ops.softargmin_conf(cost_reg, depth_values)

        if not self.refine:
            return {"depth": depth, "photometric_confidence": photometric_confidence}
        refined_depth = self.refine_network(imgs[0], depth)
        return {"depth": refined_depth, "photometric_confidence": photometric_confidence}


def mvsnet_loss(depth_est, depth_gt, mask):
    """mvsnet.py:164-166: mean smooth-L1 over the pixels with mask > 0.5.  Same value and gradient as the
    reference's boolean-index form, written without `tensor[mask]` (which launches nonzero + a host sync)."""
    m = (mask > 0.5).to(depth_est.dtype)
    per_pixel = F.smooth_l1_loss(depth_est, depth_gt, reduction='none')
    return (per_pixel * m).sum() / m.sum()
