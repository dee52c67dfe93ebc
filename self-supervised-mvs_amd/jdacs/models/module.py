"""Drop-in for jdacs/models/module.py: same public names, HIP kernels underneath.

  homo_warping(src_fea, src_proj, ref_proj, depth_values) -> [B,C,D,H,W]   (module.py:105-140)
  depth_regression(p, depth_values)                        -> [B,H,W]      (module.py:145-148)
  ConvBnReLU3D                                                             (module.py:35-42)
  ConvBnReLU (2-D, FeatureNet/RefineNet: stock PyTorch, not on the hot path; module.py:15-22)
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...nn3d import ConvBnReLU3D, DeconvBnReLU3D, ProbConv3d, count_batch  # noqa: F401

ALIGN_CORNERS = False  # what the reference's F.grid_sample call does on torch >= 1.3


def hip_conv2d_serves(conv: nn.Conv2d, x) -> bool:
    """csrc/conv2d.hip serves the feature extractors' shapes: 3x3 s1 p1 or 5x5 s2 p2, <= 32 or exactly 64 channels, fp32 on the GPU."""
    k, s, p = conv.kernel_size, conv.stride, conv.padding
    wide = lambda c: c <= 32 or c == 64
    return (x.is_cuda and x.dtype == torch.float32 and conv.groups == 1 and conv.dilation == (1, 1)
            and wide(conv.in_channels) and wide(conv.out_channels) and (k == (3, 3) or max(conv.in_channels, conv.out_channels) <= 32)
            and ((k, s, p) == ((3, 3), (1, 1), (1, 1)) or (k, s, p) == ((5, 5), (2, 2), (2, 2))))


def conv2d_maybe_hip(conv: nn.Conv2d, x):
    """nn.Conv2d through csrc/conv2d.hip when it is one of the feature extractors' shapes; the stock module otherwise."""
    return ops.Conv2dFn.apply(x, conv.weight, conv.bias, conv.stride[0]) if hip_conv2d_serves(conv, x) else conv(x)


class ConvBnReLU(nn.Module):
    """module.py:15-22.  In training inside FeatureNet the block is part of ONE autograd node (ops.FeatureExtractorFn; round 6: every
    convolution, input gradient, weight gradient and BatchNorm pass through csrc/conv2d.hip + csrc/bn.hip, two C calls per step); in eval
    mode under no_grad BatchNorm is folded into the convolution (``fold_eval``).  Used stand-alone (a user's own block, a channel count the
    kernels do not serve, RefineNet's 1-channel output layer) the convolution is the stock module's and BatchNorm2d + ReLU run through this
    library's BatchNorm kernels when the activation is channels-last on the GPU with C in {4, 8, 16, 32, 64} (``hip_bn``)."""
    hip_bn = True
    # SURVEY 8(f)-3, first cut: the convolution itself through csrc/conv2d.hip instead of MIOpen.  Parity-tested (CPU
    # emulation of the kernels), not yet measured on the GPU -> off unless MVS_HIP_FEATURE=1 / ConvBnReLU.hip_conv = True.
    hip_conv = os.environ.get("MVS_HIP_FEATURE", "0") == "1"
    # with ops.set_async_wgrad(True): weight gradient of the library's 2-D convolution on the side stream (Conv2dSplitBwdFn).
    # Opt-in.  Round 3 switched it off after a GPU comparison of two equivalent FeatureNet paths disagreed (conv0.conv.weight, 60 %
    # relative L1) with side-stream weight gradients; round 4 found the cause (ops._maybe_on_side_stream: the library's gradient
    # came back in another memory layout than the parameter, so AccumulateGrad cloned it on the main stream before the join) and
    # fixed it there; tests/test_gpu_parity.py::test_featurenet_training_hip_forward_with_fused_statistics now runs both modes.
    split_bwd = os.environ.get("MVS_SPLIT_CONV2D_BWD", "0") == "1"
    # Inference (eval mode, no autograd): BatchNorm's running statistics folded into the convolution's weights and bias, ReLU in
    # the same csrc/conv2d.hip pass -- no separate normalisation pass over the activation (jdacs/eval.py:143 runs the model
    # in eval mode under no_grad).  MVS_FOLD_EVAL_BN=0 keeps convolution and BatchNorm apart.
    fold_eval = os.environ.get("MVS_FOLD_EVAL_BN", "1") != "0"
    # training: forward convolution through csrc/conv2d.hip with BatchNorm's statistics summed in its epilogue, backward through the
    # library (its weight gradient is 3x faster than conv2d.hip's).  Default since round 4 (5 interleaved A/B pairs of the config-2
    # step: 6.186 -> 6.142 ms, profiles/r04_run1_*); MVS_HIP_FEATURE_FWD=0 restores the library's forward.
    hip_fwd_train = os.environ.get("MVS_HIP_FEATURE_FWD", "1") == "1"

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)

    # backward kernels of the training path above, chosen per layer from measurements at BASELINE config 2 (3 views of 512x640:
    # profiles/r04_run9_conv2d_layers.log): csrc/conv2d.hip's input gradient wins on the 3x3 stride-1 layers with <= 16 channels
    # (8->8: 0.035 vs 0.080 ms, 16->16: 0.026 vs 0.039), the library's on the 5x5 stride-2 layers and at 32 channels; the weight
    # gradient stays the library's unless MVS_HIP_FEATURE_WGRAD=1.  MVS_HIP_FEATURE_DGRAD=0 keeps the library's everywhere.
    hip_dgrad_auto = os.environ.get("MVS_HIP_FEATURE_DGRAD", "1") == "1"
    hip_wgrad = os.environ.get("MVS_HIP_FEATURE_WGRAD", "0") == "1"

    def _hip_dgrad(self) -> bool:
        c = self.conv
        return self.hip_dgrad_auto and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.in_channels * c.out_channels <= 256

    def hip_train_forward_serves(self, x) -> bool:
        """the training path of this block through csrc/conv2d.hip with fused BatchNorm statistics (hip_fwd_train) applies to x"""
        return (self.block_trains_with_batch_statistics() and x.is_cuda and torch.is_grad_enabled() and hip_conv2d_serves(self.conv, x)
                and x.is_contiguous(memory_format=torch.channels_last))

    def block_trains_with_batch_statistics(self) -> bool:
        """What the one-node extractor (ops.FeatureExtractorFn) assumes of EVERY block, not only the first: train mode on the block
        AND its BatchNorm (a block put in .eval() for frozen-statistics fine-tuning must normalise with its running statistics, as the
        reference's module would), an affine BatchNorm with tracked running statistics and an exponential moving average (momentum
        None = cumulative average is the stock module's business), a bias-free ungrouped convolution, a channel count the BatchNorm
        kernels serve.  Instance attributes win over the class-level switches."""
        bn, c = self.bn, self.conv
        return (self.hip_fwd_train and not self.hip_conv and self.hip_bn and self.training and bn.training and bn.affine
                and bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None
                and c.bias is None and c.groups == 1 and c.dilation == (1, 1) and c.out_channels in (4, 8, 16, 32, 64))

    def forward(self, x, groups=1, packed_ws=None):
        """groups > 1: x holds `groups` equal batch chunks that the reference would pass through this block one
        after the other (the views of a sample); BatchNorm statistics / running-stat updates stay per chunk.
        packed_ws: this block's forward weight image, already written (ops.pack_conv2d_weights; FeatureNet packs all its blocks
        in one launch)."""
        if (self.fold_eval and not self.training and not torch.is_grad_enabled() and self.bn.track_running_stats
                and self.bn.running_mean is not None and self.bn.affine and hip_conv2d_serves(self.conv, x)):
            w, b = self._folded()
            return ops.conv2d_forward(x, w, b, self.conv.stride[0], negative_slope=0.0)
        if self.hip_conv:
            y = conv2d_maybe_hip(self.conv, x)
        elif ((ops._ASYNC_WGRAD and self.split_bwd or self.hip_fwd_train) and x.is_cuda and self.training and torch.is_grad_enabled()
              and self.conv.bias is None and self.conv.groups == 1 and self.conv.dilation == (1, 1)):
            # opt-in side-stream weight gradients (ops.set_async_wgrad): the library convolution with its backward issued as two calls
            hip_fwd = (self.hip_fwd_train and hip_conv2d_serves(self.conv, x) and x.is_contiguous(memory_format=torch.channels_last))
            if hip_fwd and self.hip_bn and self.conv.out_channels in (4, 8, 16, 32, 64) and self.bn.momentum is not None:
                # convolution + BatchNorm statistics in one launch, then finalize + apply: no statistics pass over the activation
                y, slots = ops.Conv2dSplitBwdFn.apply(x, self.conv.weight, self.conv.stride, self.conv.padding, True, True, groups,
                                                      self.split_bwd, packed_ws, self._hip_dgrad(), self.hip_wgrad)
                for _ in range(groups):
                    count_batch(self.bn, self.training)
                return ops.BnReLUFn.apply(y, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var, True,
                                          self.bn.eps, self.bn.momentum, groups, slots)
            y = ops.Conv2dSplitBwdFn.apply(x, self.conv.weight, self.conv.stride, self.conv.padding, hip_fwd, False, 1, self.split_bwd)
        else:
            y = self.conv(x)
        bn = self.bn
        # the BatchNorm kernels serve 4/8/16/32/64 channels and an exponential moving average; anything else (e.g. 12 or 48
        # channels in a user's own ConvBnReLU, or momentum=None = cumulative average) takes the stock modules
        if (self.hip_bn and y.is_cuda and y.dtype == torch.float32 and y.shape[1] in (4, 8, 16, 32, 64) and bn.momentum is not None
                and y.is_contiguous(memory_format=torch.channels_last)):
            for _ in range(groups):
                count_batch(bn, self.training)
            momentum = bn.momentum
            return ops.BnReLUFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, bn.eps,
                                      momentum, groups)
        if groups > 1:
            return torch.cat([F.relu(bn(c)) for c in y.chunk(groups, 0)], 0)
        return F.relu(bn(y), inplace=True)


def _fold_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """(w', b') with  conv(x, w') + b' == bn_eval(conv(x, w) + bias)."""
    scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    if conv.bias is not None:
        shift = shift + conv.bias * scale
    return (conv.weight * scale.view(-1, 1, 1, 1)).contiguous(), shift.contiguous()


def _folded(self):
    # recomputed only when one of the five tensors changed (in-place updates bump ._version; load_state_dict copies in place)
    srcs = (self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
    key = tuple((t.data_ptr(), t._version) for t in srcs) + (self.bn.eps, None if self.conv.bias is None else self.conv.bias._version)
    cache = self.__dict__.get("_fold_cache")
    if cache is None or cache[0] != key:
        with torch.no_grad():
            cache = (key, _fold_bn(self.conv, self.bn))
        self.__dict__["_fold_cache"] = cache
    return cache[1]


def _drop_fold(self):
    """forget the folded weights (call after changing parameters through `.data`, which does not bump their version counters)"""
    self.__dict__.pop("_fold_cache", None)


def _train(self, mode=True):
    self.__dict__.pop("_fold_cache", None)
    return nn.Module.train(self, mode)


def _apply_and_drop(self, fn, *args, **kwargs):
    self.__dict__.pop("_fold_cache", None)          # .to() / .cuda() / .half(): new parameter tensors
    return nn.Module._apply(self, fn, *args, **kwargs)


def _getstate(self):
    state = dict(self.__dict__)                      # deepcopy / pickle: the cache is derived data
    state.pop("_fold_cache", None)
    return state


ConvBnReLU._folded = _folded
ConvBnReLU.refold = _drop_fold
ConvBnReLU.train = _train
ConvBnReLU._apply = _apply_and_drop
ConvBnReLU.__getstate__ = _getstate


def homo_warping(src_fea, src_proj, ref_proj, depth_values, align_corners=None):
    """Warp one source feature map into every depth plane of the reference view."""
    with torch.no_grad():
        rot, trans = ops.relative_projection(src_proj, ref_proj)
    ac = ALIGN_CORNERS if align_corners is None else align_corners
    return ops.HomoWarp.apply(src_fea, rot, trans, depth_values, ac)


def depth_regression(p, depth_values):
    """Expectation of the depth hypotheses under p [B,D,H,W] (an already-normalised probability
    volume).  Tiny elementwise host op; the fused softmax+regression kernel is ops.softargmin_conf."""
    if depth_values.dim() <= 2:
        depth_values = depth_values.view(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)
