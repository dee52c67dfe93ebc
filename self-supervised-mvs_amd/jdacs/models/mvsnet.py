"""Drop-in for jdacs/models/mvsnet.py: MVSNet(refine).forward(imgs, proj_matrices, depth_values)
-> {"depth", "photometric_confidence"} with identical state_dict names.

Hot path (HIP): plane-sweep variance volume (mvsnet.py:120-136 + module.py:105-140), CostRegNet
(mvsnet.py:37-74), softmax + depth regression + confidence (mvsnet.py:141-151).
FeatureNet / RefineNet are 2-D CNNs next to the path: they keep the reference's parameter names and run through
csrc/conv2d.hip where that measured faster (eval: folded BatchNorm; training: ops.FeatureExtractorFn) and the library elsewhere."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...nn3d import batched_bn_counters, count_batch
from .module import ALIGN_CORNERS, ConvBnReLU, ConvBnReLU3D, DeconvBnReLU3D, ProbConv3d, conv2d_maybe_hip, hip_conv2d_serves


# (name, Cin, Cout, kernel, stride, pad) in registration order == the reference's (same seed => same init)
_FEATURE_LAYERS = (("conv0", 3, 8, 3, 1, 1), ("conv1", 8, 8, 3, 1, 1), ("conv2", 8, 16, 5, 2, 2), ("conv3", 16, 16, 3, 1, 1),
                   ("conv4", 16, 16, 3, 1, 1), ("conv5", 16, 32, 5, 2, 2), ("conv6", 32, 32, 3, 1, 1))
# (name, Cin, Cout, stride) of the encoder, then (name, Cin, Cout, skip source) of the decoder
_REG_ENCODER = (("conv0", 32, 8, 1), ("conv1", 8, 16, 2), ("conv2", 16, 16, 1), ("conv3", 16, 32, 2), ("conv4", 32, 32, 1),
                ("conv5", 32, 64, 2), ("conv6", 64, 64, 1))
_REG_DECODER = (("conv7", 64, 32, "conv4"), ("conv9", 32, 16, "conv2"), ("conv11", 16, 8, "conv0"))


class FeatureNet(nn.Module):
    """mvsnet.py:17-34 -- 2-D CNN, 3 -> 32 channels at 1/4 resolution; kernel choice per mode and layer: DESIGN.md section 7."""

    one_node = os.environ.get("MVS_FEATURE_ONE_NODE", "1") != "0"   # training: the extractor as one autograd node (ops.FeatureExtractorFn)

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        for name, cin, cout, k, stride, pad in _FEATURE_LAYERS:
            setattr(self, name, ConvBnReLU(cin, cout, k, stride, pad))
        self.feature = nn.Conv2d(32, 32, 3, 1, 1)

    def forward(self, x, groups=1):
        """groups: number of views stacked along the batch dim (per-view BatchNorm statistics are kept)."""
        blocks = [getattr(self, name) for name, *_ in _FEATURE_LAYERS]
        packed = [None] * len(blocks)
        if (blocks[0].hip_train_forward_serves(x) and all(hip_conv2d_serves(m.conv, x) for m in blocks)
                and all(m.block_trains_with_batch_statistics() for m in blocks)):
            f = self.feature
            closing_ok = (f.bias is not None and f.kernel_size == (3, 3) and f.stride == (1, 1) and f.padding == (1, 1)
                          and f.dilation == (1, 1) and f.groups == 1)      # what FeatureExtractorFn's closing convolution assumes
            if (self.one_node and closing_ok and not any(m.split_bwd or m.hip_wgrad for m in blocks)):
                # the whole extractor as ONE autograd node (ops.FeatureExtractorFn): same kernels, a third of the host work
                cfg, params = [], []
                for m in blocks:
                    for _ in range(groups):
                        count_batch(m.bn, True)
                    cfg.append((m.conv.stride[0], m.conv.padding[0], float(m.bn.eps), float(m.bn.momentum), m._hip_dgrad()))
                    params += [m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var]
                params += [self.feature.weight, self.feature.bias]
                return ops.FeatureExtractorFn.apply(x, groups, tuple(cfg), *params)
            # training through csrc/conv2d.hip: the weight images of all blocks in ONE launch (channels-last parameters read as they are)
            packed = ops.pack_conv2d_weights([m.conv.weight for m in blocks], [m.conv.stride[0] for m in blocks], x)
        for m, ws in zip(blocks, packed):
            x = m(x, groups, ws)
        # inference: the closing convolution through csrc/conv2d.hip like the folded blocks before it
        hip = ConvBnReLU.hip_conv or (ConvBnReLU.fold_eval and not self.training and not torch.is_grad_enabled())
        return conv2d_maybe_hip(self.feature, x) if hip else self.feature(x)


class CostRegNet(nn.Module):
    """mvsnet.py:37-74: 3-D U-Net 32 -> 8 -> 16 -> 32 -> 64 -> ... -> 1; D, H, W divisible by 8.  Decoder blocks
    add their skip AFTER the ReLU (mvsnet.py:70-72), fused into the block's BatchNorm-apply kernel."""

    def __init__(self):
        super().__init__()
        for name, cin, cout, stride in _REG_ENCODER:
            setattr(self, name, ConvBnReLU3D(cin, cout, stride=stride))
        for name, cin, cout, _ in _REG_DECODER:
            setattr(self, name, DeconvBnReLU3D(cin, cout, stride=2))
        self.prob = ProbConv3d(8)

    def conv_weights(self):
        """the convolution weights whose gradients the fused node computes on the side stream (ops.tail_join_views)"""
        ws = [getattr(self, name).conv.weight for name, *_ in _REG_ENCODER] + [getattr(self, name)[0].weight for name, *_ in _REG_DECODER]
        return ws + [self.prob.weight]

    def forward(self, x, tail=None):
        """tail: ops.tail_join_views(self.conv_weights()) made at the START of the model's forward pass (MVSNet._forward): the join of
        the side-stream weight gradients then happens at the end of the backward pass, safely (ops.DeferredJoinFn)."""
        if x.dim() != 5 or x.shape[1] != 32:
            raise ValueError("CostRegNet expects [B,32,D,H,W], got %s" % (tuple(x.shape),))
        if x.dtype == torch.bfloat16 and self.training:
            raise RuntimeError("CostRegNet: bf16 activations are the eval-mode inference path")
        if any(s % 8 for s in x.shape[2:]):
            raise ValueError("CostRegNet needs D,H,W divisible by 8, got %s" % (tuple(x.shape[2:]),))
        if self.training and x.dtype == torch.float32 and ops.FUSED_REGULARISER:
            # the whole U-Net as one autograd node (ops.UNetRegulariserFn): same kernels, skip gradients summed in the dgrad
            # epilogues, weight gradients on a side stream
            order = [name for name, *_ in _REG_ENCODER] + [name for name, *_ in _REG_DECODER]
            blocks = []
            for i, (name, _, _, stride) in enumerate(_REG_ENCODER):
                m = getattr(self, name)
                blocks.append((m.conv, m.bn, False, stride, i - 1, -1))
            for j, (name, _, _, skip) in enumerate(_REG_DECODER):
                m = getattr(self, name)
                blocks.append((m[0], m[1], True, 2, len(_REG_ENCODER) + j - 1, order.index(skip)))
            return ops.unet_regulariser(x, blocks, self.prob, tail)
        keep = {}
        for name, *_ in _REG_ENCODER:
            x = getattr(self, name)(x)
            keep[name] = x
        for name, _, _, skip in _REG_DECODER:
            x = getattr(self, name)(x, skip=keep[skip])
        return self.prob(x)


class RefineNet(nn.Module):
    """mvsnet.py:77-92 (off by default in the reference, config.py:48): residual on the 1/4-resolution image + depth."""

    def __init__(self):
        super().__init__()
        for name, cin, cout in (("conv1", 4, 32), ("conv2", 32, 32), ("conv3", 32, 32), ("res", 32, 1)):
            setattr(self, name, ConvBnReLU(cin, cout))

    def forward(self, img, depth_init):
        small = F.interpolate(img, scale_factor=0.25, mode='bilinear')
        d = depth_init.unsqueeze(1)
        x = torch.cat((small, d), dim=1)
        for name in ("conv1", "conv2", "conv3", "res"):
            x = getattr(self, name)(x)
        return (d + x).squeeze(1)


class MVSNet(nn.Module):
    def __init__(self, refine=True, align_corners=ALIGN_CORNERS, channels_last_features=True):
        super().__init__()
        self.refine = refine
        self.align_corners = align_corners
        # optional: run the (stock PyTorch) 2-D feature extractor in channels-last so MIOpen picks NHWC kernels
        # and the features arrive in the layout the plane-sweep kernel reads (no NCHW->NHWC transpose)
        self.channels_last_features = channels_last_features
        # torch.bfloat16: eval-mode inference stores the cost volume and the regulariser's activations in bf16 (fp32
        # accumulation, fp32 logits / soft-argmin) -- BASELINE configs[4]; training and the default are fp32
        self.storage_dtype = torch.float32
        self.feature = FeatureNet()
        if channels_last_features:
            # once, at construction: parameter storage must never be re-allocated inside forward() (an optimiser or a
            # flat parameter/gradient bucket built before the first step would silently stop tracking the weights)
            self.feature.to(memory_format=torch.channels_last)
        self.cost_regularization = CostRegNet()
        if self.refine:
            self.refine_network = RefineNet()

    def forward(self, imgs, proj_matrices, depth_values):
        with batched_bn_counters(), ops.slot_scope():
            return self._forward(imgs, proj_matrices, depth_values)

    def _forward(self, imgs, proj_matrices, depth_values):
        # the tail node of the regulariser's weight gradients FIRST: the autograd engine runs ready nodes newest first, so the node
        # created before everything else of this forward pass is the one that runs last in the backward pass (ops.DeferredJoinFn)
        tail = None
        if self.training and torch.is_grad_enabled() and imgs.is_cuda and self.storage_dtype == torch.float32 and ops.FUSED_REGULARISER:
            tail = ops.tail_join_views(self.cost_regularization.conv_weights())
        imgs_in = imgs
        imgs = torch.unbind(imgs, 1)
        proj_matrices = torch.unbind(proj_matrices, 1)
        assert len(imgs) == len(proj_matrices), "Different number of images and projection matrices"

        # step 1. feature extraction
        if self.channels_last_features:
            # all views through the shared-weight extractor as ONE batch (3x fewer launches, no per-view
            # gradient accumulation); BatchNorm keeps the reference's per-view statistics (grouped BN kernels)
            # (view-major [N*B,3,H,W]: for one sample per GPU a plain view of the input, so the only copy is the layout change)
            stacked = imgs_in.transpose(0, 1).reshape(-1, *imgs_in.shape[2:]).contiguous(memory_format=torch.channels_last)
            features = list(self.feature(stacked, groups=len(imgs)).chunk(len(imgs), 0))
        else:
            features = [self.feature(img) for img in imgs]
        ref_feature, src_features = features[0], features[1:]
        ref_proj, src_projs = proj_matrices[0], proj_matrices[1:]

        # step 2. homography warp + variance cost volume: ONE fused HIP kernel, volume written once
        with torch.no_grad():
            rot, trans = ops.relative_projections(src_projs, ref_proj, like=ref_feature)   # module.py:116-118 for all source views, one launch
        if self.storage_dtype == torch.bfloat16 and (self.training or torch.is_grad_enabled()):
            raise RuntimeError("MVSNet.storage_dtype = bfloat16 is the inference path: call .eval() and run under torch.no_grad()")
        volume_variance = ops.plane_sweep_variance(ref_feature, src_features, rot, trans, depth_values,
                                                   align_corners=self.align_corners, ms_alias=False,
                                                   out_dtype=self.storage_dtype)

        # step 3. cost volume regularisation (MFMA implicit-GEMM convs)
        cost_reg = self.cost_regularization(volume_variance, tail).squeeze(1)

        # step 4. softmax over depth + soft-argmin + photometric confidence (one kernel)
        depth, photometric_confidence = ops.softargmin_conf(cost_reg, depth_values)

        if not self.refine:
            return {"depth": depth, "photometric_confidence": photometric_confidence}
        refined_depth = self.refine_network(imgs[0], depth)
        return {"depth": refined_depth, "photometric_confidence": photometric_confidence}


_FUSED_LOSS_MAX_PIXELS = 1 << 19      # 512 k pixels: ~25 us for the one-workgroup kernel, about what the torch launches cost


def mvsnet_loss(depth_est, depth_gt, mask):
    """mvsnet.py:164-166: mean smooth-L1 over the pixels with mask > 0.5.  fp32 maps on the GPU: one HIP launch forward, one backward
    (ops.MaskedSmoothL1); anything else: the same value and gradient from torch ops, written without `tensor[mask]` (which
    launches nonzero + a host sync)."""
    # the fused kernel is ONE workgroup walking the map (built for the 20 k pixels of a quarter-resolution map, where the torch
    # formulation is 14 launches of pure latency); a full-resolution batch (millions of pixels) is a bandwidth problem that one CU
    # streams slowly -- there the torch ops are the faster form.  Broadcastable gt / mask are expanded, as the torch
    # formulation always accepted them.
    if depth_est.is_cuda and depth_est.dtype == torch.float32 and depth_est.numel() <= _FUSED_LOSS_MAX_PIXELS:
        if depth_gt.shape != depth_est.shape or mask.shape != depth_est.shape:
            depth_gt, mask = torch.broadcast_to(depth_gt, depth_est.shape), torch.broadcast_to(mask, depth_est.shape)
        return ops.MaskedSmoothL1.apply(depth_est, depth_gt, mask)
    m = (mask > 0.5).to(depth_est.dtype)
    per_pixel = F.smooth_l1_loss(depth_est, depth_gt, reduction='none')
    return (per_pixel * m).sum() / m.sum()
