"""Building blocks of the 3-D regularisers, with the reference's parameter names.

The stock ``nn.Conv3d`` / ``nn.ConvTranspose3d`` / ``nn.BatchNorm3d`` modules are kept ONLY as
parameter / buffer containers (identical ``state_dict`` keys, shapes and default initialisation as
jdacs/models/module.py:35-42 and jdacs/models/mvsnet.py:48-63, so reference checkpoints load
unchanged); their ``forward`` is never called -- the arithmetic runs in the HIP kernels.
"""
import torch
import torch.nn as nn

from . import ops


_DEFERRED = None  # {id: [tensor, count]} while a model forward batches the num_batches_tracked increments


class batched_bn_counters:
    """Context manager: BatchNorm ``num_batches_tracked += 1`` of every block executed inside is applied as ONE
    multi-tensor launch at exit instead of one tiny kernel per BatchNorm layer (31 per MVSNet step)."""

    def __enter__(self):
        global _DEFERRED
        self.outer = _DEFERRED
        if _DEFERRED is None:
            _DEFERRED = {}
        return self

    def __exit__(self, *exc):
        global _DEFERRED
        if self.outer is None:
            pending, _DEFERRED = _DEFERRED, None
            if pending and exc[0] is None:
                torch._foreach_add_([t for t, _ in pending.values()], [n for _, n in pending.values()])
        return False


def count_batch(bn, training: bool):
    """num_batches_tracked bookkeeping of nn.BatchNorm*.forward (torch/nn/modules/batchnorm.py)."""
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if _DEFERRED is None:
            bn.num_batches_tracked.add_(1)
        else:
            ent = _DEFERRED.setdefault(id(bn.num_batches_tracked), [bn.num_batches_tracked, 0])
            ent[1] += 1


def _require_eval(module, what):
    if module.training:
        raise RuntimeError("mvs_amd: %s got bf16 activations in train mode; bf16 storage is the eval-mode inference path "
                           "(BatchNorm folded into the convolution)" % what)


def _bn_step(bn: nn.BatchNorm3d, training: bool):
    if bn.momentum is None:
        raise ValueError("mvs_amd: BatchNorm3d(momentum=None) (cumulative moving average) is not supported by the fused kernels; "
                         "the reference uses the default momentum 0.1 (jdacs/models/module.py:39)")
    count_batch(bn, training)
    return bn.momentum


class ConvBnReLU3D(nn.Module):
    """jdacs/models/module.py:35-42 (and jdacs-ms/models/modules.py:285-292).  ``skip`` (optional) is
    added after the ReLU, which is how the U-Nets use it (mvsnet.py:70-72)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        if kernel_size != 3 or pad != 1 or stride not in (1, 2):
            raise ValueError("mvs_amd ConvBnReLU3D supports kernel_size=3, pad=1, stride 1|2 (the reference's use)")
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm3d(out_channels)
        self.stride = stride

    def forward(self, x, skip=None):
        bn = self.bn
        if x.dtype == torch.bfloat16:
            _require_eval(self, "ConvBnReLU3D")
            return ops.conv_bn_relu3d_eval_bf16(x, self.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, skip,
                                                self.stride, False, bn.eps)
        momentum = _bn_step(bn, self.training)
        return ops.ConvBnReLU3dFn.apply(x, self.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                        skip, self.stride, False, self.training, bn.eps, momentum)


class DeconvBnReLU3D(nn.Sequential):
    """nn.Sequential(ConvTranspose3d(k3, p1, output_padding=stride-1, bias=False), BatchNorm3d, ReLU)
    with keys ``0.weight`` / ``1.*`` (jdacs/models/mvsnet.py:48-61; jdacs-ms/models/network.py:55-64)."""

    def __init__(self, in_channels, out_channels, stride=2):
        if stride not in (1, 2):
            raise ValueError("stride must be 1 or 2")
        super().__init__(
            nn.ConvTranspose3d(in_channels, out_channels, kernel_size=3, padding=1, output_padding=stride - 1,
                               stride=stride, bias=False),
            nn.BatchNorm3d(out_channels),
            nn.ReLU(inplace=True))
        self.stride = stride

    def forward(self, x, skip=None):
        bn = self[1]
        if x.dtype == torch.bfloat16:
            _require_eval(self, "DeconvBnReLU3D")
            return ops.conv_bn_relu3d_eval_bf16(x, self[0].weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, skip,
                                                self.stride, True, bn.eps)
        momentum = _bn_step(bn, self.training)
        return ops.ConvBnReLU3dFn.apply(x, self[0].weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, skip,
                                        self.stride, True, self.training, bn.eps, momentum)


class ProbConv3d(nn.Conv3d):
    """nn.Conv3d(C, 1, 3, stride=1, padding=1) with bias (mvsnet.py:63); keys ``weight`` / ``bias``."""

    def __init__(self, in_channels):
        super().__init__(in_channels, 1, 3, stride=1, padding=1)

    def forward(self, x):
        if x.dtype == torch.bfloat16:   # inference path: bf16 activations in, fp32 logits out
            return ops.conv3d_forward_bf16(x, self.weight, 1, False, shift=self.bias, out_f32=True)
        return ops.ConvBias3dFn.apply(x, self.weight, self.bias)
