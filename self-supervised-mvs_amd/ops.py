"""torch <-> libmvs_hip.so glue: functional wrappers and autograd Functions for the hot path.

Host code is PyTorch (device memory, streams, autograd graph); all arithmetic on the path happens in
the HIP kernels behind the C ABI (include/mvs_hip.h).  Tensors keep the reference's LOGICAL shapes
([B,C,H,W], [B,C,D,H,W]) but are physically channels-last (torch.channels_last / channels_last_3d),
which is the layout the kernels index.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (OP_CONV_DGRAD, OP_CONV_FWD, OP_CONV_WGRAD, OP_CONVT_DGRAD, OP_CONVT_FWD, OP_CONVT_WGRAD)

CL2 = torch.channels_last
CL3 = torch.channels_last_3d


def _lib_for(t: torch.Tensor) -> _lib.MvsLib:
    lib = _lib.get()
    if t.device.type != lib.device_type:
        raise RuntimeError("mvs_amd: tensor on %s but the HIP library serves '%s' devices; the hot path has no "
                           "CPU fallback" % (t.device, lib.device_type))
    if t.dtype != torch.float32:
        raise TypeError("mvs_amd: fp32 tensors required, got %s (bf16 storage exists for the inference path only: "
                        "MVSNet.storage_dtype / ops.conv3d_forward_bf16)" % t.dtype)
    return lib


def _lib_for_bf16(t: torch.Tensor) -> _lib.MvsLib:
    lib = _lib.get()
    if t.device.type != lib.device_type:
        raise RuntimeError("mvs_amd: tensor on %s but the HIP library serves '%s' devices; the hot path has no "
                           "CPU fallback" % (t.device, lib.device_type))
    if t.dtype != torch.bfloat16:
        raise TypeError("mvs_amd: bf16 tensor expected, got %s" % t.dtype)
    return lib


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _ptr_array(ts: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr


def as_cl2(t: torch.Tensor) -> torch.Tensor:
    # (the layout test costs a tenth of a no-op .contiguous() call; these run ~300 times per training step)
    return t if t.is_contiguous(memory_format=CL2) else t.contiguous(memory_format=CL2)


def as_cl3(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous(memory_format=CL3) else t.contiguous(memory_format=CL3)


def empty_cl3(b, c, d, h, w, like: torch.Tensor) -> torch.Tensor:
    return torch.empty((b, c, d, h, w), dtype=torch.float32, device=like.device, memory_format=CL3)


# ------------------------------------------------------------------------------------------------
# plane sweep (K1/K2)
# ------------------------------------------------------------------------------------------------
def relative_projection(src_proj: torch.Tensor, ref_proj: torch.Tensor):
    """rot [B,3,3] / trans [B,3] of src_proj @ inverse(ref_proj) -- host torch code, the reference's lines
    (jdacs/models/module.py:116-118).  ``linalg.inv_ex`` is the same LU inverse as ``torch.inverse`` without the
    host-side singularity check, i.e. without a device synchronisation (and therefore capturable in a hipGraph);
    a singular matrix yields inf/nan instead of an exception."""
    proj = torch.matmul(src_proj, torch.linalg.inv_ex(ref_proj).inverse)
    return proj[:, :3, :3], proj[:, :3, 3]


def relative_projections(src_projs, ref_proj: torch.Tensor, like: Optional[torch.Tensor] = None):
    """rot [B,NS,3,3] / trans [B,NS,3] of src_projs[s] @ inverse(ref_proj) for ALL source views in one launch
    (``mvs_relative_projection``: fp64 Gauss-Jordan + product per (sample, view)) instead of one LU inverse + matmul + slices
    per view (jdacs/models/module.py:116-118 runs once per source view: ~10 tiny launches, 91 us of the training step).
    ``like``: the feature map the result will be used with -- cameras that live elsewhere (CPU-resident cameras next to GPU
    features: the reference's `.cuda()` calls are the caller's business) are copied to its device first, so the result is
    always where the plane-sweep kernel reads it."""
    ns = len(src_projs)
    b = ref_proj.shape[0]
    if tuple(ref_proj.shape) != (b, 4, 4) or any(tuple(p.shape) != (b, 4, 4) for p in src_projs):
        raise ValueError("relative_projections: need NS x [B,4,4] and [B,4,4], got %s and %s"
                         % ([tuple(p.shape) for p in src_projs], tuple(ref_proj.shape)))
    dev = like.device if like is not None else ref_proj.device
    if ref_proj.device != dev or any(p.device != dev for p in src_projs):
        ref_proj, src_projs = ref_proj.to(dev), [p.to(dev) for p in src_projs]
    # the kernel computes in fp64 from fp32 inputs; float64 / half matrices are cast first (the reference's lines are dtype
    # agnostic).  Matrices (and features) that do not live on the library's device -- the CPU emulation build aside, that is a
    # CPU tensor handed to the GPU build -- take the reference's own host lines and stay where they are.
    if dev.type != _lib.get().device_type:
        rts = [relative_projection(p, ref_proj) for p in src_projs]
        return torch.stack([r for r, _ in rts], 1).float(), torch.stack([t for _, t in rts], 1).float()
    src = torch.stack([p.to(torch.float32) for p in src_projs], 1).contiguous()
    ref = ref_proj.to(torch.float32).contiguous()
    lib = _lib_for(ref)
    rot = torch.empty((b, ns, 3, 3), dtype=torch.float32, device=ref.device)
    trans = torch.empty((b, ns, 3), dtype=torch.float32, device=ref.device)
    lib.call("mvs_relative_projection", _p(src), _p(ref), b, ns, _p(rot), _p(trans), _stream(ref))
    return rot, trans


def _depth_arg(depth: torch.Tensor, b: int, h: int, w: int):
    if depth.dim() == 2:
        return depth.contiguous(), 0
    if depth.dim() == 4 and depth.shape[0] == b and depth.shape[2] == h and depth.shape[3] == w:
        return depth.contiguous(), 1
    raise ValueError("depth hypotheses must be [B,D] or [B,D,H,W], got %s" % (tuple(depth.shape),))


class PlaneSweepVariance(torch.autograd.Function):
    """var[B,C,D,H,W] = variance over {ref, warped sources}.  Differentiable
    w.r.t. the feature maps only, like the reference (grid built under no_grad)."""

    @staticmethod
    def forward(ctx, depth, rot, trans, align_corners, ms_alias, ref, *srcs):
        lib = _lib_for(ref)
        b, c, h, w = ref.shape
        n = len(srcs) + 1
        ref_c = as_cl2(ref)
        srcs_c = [as_cl2(s) for s in srcs]
        for s in srcs_c:
            if s.shape != ref_c.shape:
                raise ValueError("source feature map shape %s != reference %s" % (tuple(s.shape), tuple(ref.shape)))
        depth_c, per_pixel = _depth_arg(depth, b, h, w)
        nd = depth_c.shape[1]
        rot_c = rot.reshape(b, n - 1, 9).contiguous().float()
        trans_c = trans.reshape(b, n - 1, 3).contiguous().float()
        var = empty_cl3(b, c, nd, h, w, ref)
        lib.call("mvs_plane_sweep_variance_fwd", _p(ref_c), _ptr_array(srcs_c), _p(rot_c), _p(trans_c), _p(depth_c),
                 per_pixel, b, n, c, nd, h, w, int(align_corners), int(ms_alias), _p(var), _stream(ref),
                 tag="sweep_fwd:N%d:C%d:%dx%dx%dx%d" % (n, c, b, nd, h, w))
        ctx.save_for_backward(ref_c, depth_c, rot_c, trans_c, *srcs_c)
        ctx.cfg = (per_pixel, int(align_corners), int(ms_alias))
        return var

    @staticmethod
    def backward(ctx, gvar):
        ref_c, depth_c, rot_c, trans_c, *srcs_c = ctx.saved_tensors
        per_pixel, align_corners, ms_alias = ctx.cfg
        lib = _lib_for(ref_c)
        b, c, h, w = ref_c.shape
        n = len(srcs_c) + 1
        nd = depth_c.shape[1]
        g = as_cl3(gvar)
        # ONE zero-filled [N,B,H,W,C] buffer (one fill launch instead of N), handed out as channels-last [B,C,H,W] views
        gall = torch.zeros((n, b, h, w, c), dtype=torch.float32, device=ref_c.device)
        gref = gall[0].permute(0, 3, 1, 2)
        gsrcs = [gall[i + 1].permute(0, 3, 1, 2) for i in range(n - 1)]
        lib.call("mvs_plane_sweep_variance_bwd", _p(g), _p(ref_c), _ptr_array(srcs_c), _p(rot_c), _p(trans_c),
                 _p(depth_c), per_pixel, b, n, c, nd, h, w, align_corners, ms_alias, _p(gref), _ptr_array(gsrcs),
                 _stream(ref_c), tag="sweep_bwd:N%d:C%d:%dx%dx%dx%d" % (n, c, b, nd, h, w))
        return (None, None, None, None, None, gref, *gsrcs)


def plane_sweep_variance(ref, srcs, rot, trans, depth, align_corners=False, ms_alias=False, out_dtype=torch.float32):
    """ref [B,C,H,W]; srcs list of [B,C,H,W]; rot [B,N-1,3,3]; trans [B,N-1,3]; depth [B,D]|[B,D,H,W].
    out_dtype=torch.bfloat16: the inference path's bf16 volume (no gradient; BASELINE configs[4])."""
    if out_dtype == torch.bfloat16:
        return plane_sweep_variance_bf16(ref, srcs, rot, trans, depth, align_corners, ms_alias)
    if out_dtype != torch.float32:
        raise TypeError("plane_sweep_variance: out_dtype must be float32 or bfloat16, got %s" % out_dtype)
    return PlaneSweepVariance.apply(depth, rot, trans, align_corners, ms_alias, ref, *srcs)


def plane_sweep_variance_bf16(ref, srcs, rot, trans, depth, align_corners=False, ms_alias=False):
    """Forward only: variance volume [B,C,D,H,W] stored in bf16 (channels_last_3d), computed in fp32 from fp32 features."""
    lib = _lib_for(ref)
    if torch.is_grad_enabled() and (ref.requires_grad or any(s.requires_grad for s in srcs)):
        raise RuntimeError("mvs_amd: the bf16 cost volume is an inference path (call it under torch.no_grad())")
    b, c, h, w = ref.shape
    n = len(srcs) + 1
    ref_c = as_cl2(ref)
    srcs_c = [as_cl2(s) for s in srcs]
    depth_c, per_pixel = _depth_arg(depth, b, h, w)
    nd = depth_c.shape[1]
    rot_c = rot.reshape(b, n - 1, 9).contiguous().float()
    trans_c = trans.reshape(b, n - 1, 3).contiguous().float()
    var = torch.empty((b, c, nd, h, w), dtype=torch.bfloat16, device=ref.device, memory_format=CL3)
    lib.call("mvs_plane_sweep_variance_fwd_bf16", _p(ref_c), _ptr_array(srcs_c), _p(rot_c), _p(trans_c), _p(depth_c), per_pixel, b,
             n, c, nd, h, w, int(align_corners), int(ms_alias), _p(var), _stream(ref),
             tag="sweep_fwd_bf16:N%d:C%d:%dx%dx%dx%d" % (n, c, b, nd, h, w))
    return var


class HomoWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, rot, trans, depth, align_corners):
        lib = _lib_for(src)
        b, c, h, w = src.shape
        src_c = as_cl2(src)
        depth_c, per_pixel = _depth_arg(depth, b, h, w)
        nd = depth_c.shape[1]
        rot_c = rot.reshape(b, 9).contiguous().float()
        trans_c = trans.reshape(b, 3).contiguous().float()
        out = empty_cl3(b, c, nd, h, w, src)
        lib.call("mvs_homo_warp_fwd", _p(src_c), _p(rot_c), _p(trans_c), _p(depth_c), per_pixel, b, c, nd, h, w,
                 int(align_corners), _p(out), _stream(src))
        ctx.save_for_backward(src_c, rot_c, trans_c, depth_c)
        ctx.cfg = (per_pixel, int(align_corners))
        return out

    @staticmethod
    def backward(ctx, gout):
        src_c, rot_c, trans_c, depth_c = ctx.saved_tensors
        per_pixel, align_corners = ctx.cfg
        lib = _lib_for(src_c)
        b, c, h, w = src_c.shape
        g = as_cl3(gout)
        gsrc = torch.zeros_like(src_c, memory_format=CL2)
        lib.call("mvs_homo_warp_bwd", _p(g), _p(src_c), _p(rot_c), _p(trans_c), _p(depth_c), per_pixel, b, c,
                 depth_c.shape[1], h, w, align_corners, _p(gsrc), _stream(src_c))
        return gsrc, None, None, None, None


# ------------------------------------------------------------------------------------------------
# 3-D convolution family + BatchNorm (K3-K8)
# ------------------------------------------------------------------------------------------------
def _ws_floats(lib, op, b, d, h, w, cin, cout, stride):
    nbytes = lib.raw("mvs_conv3d_workspace_bytes", op, b, d, h, w, cin, cout, stride)
    if nbytes < 0:
        raise ValueError("mvs_conv3d_workspace_bytes: bad op %d" % op)
    return (nbytes + 15) // 16 * 4        # floats, a multiple of 16 bytes


def _ws(lib, op, b, d, h, w, cin, cout, stride, like):
    return torch.empty(_ws_floats(lib, op, b, d, h, w, cin, cout, stride), dtype=torch.float32, device=like.device)


def _ctag(kind, cin, cout, stride, b, d, h, w):
    # formatted by _lib.MvsLib.call only when a KernelTimer is attached ("%s:%d>%d:s%d:%dx%dx%dx%d")
    return ("%s:%d>%d:s%d:%dx%dx%dx%d", kind, cin, cout, stride, b, d, h, w)


def _out_dims(d, h, w, stride, transposed):
    if transposed:
        return (d * stride, h * stride, w * stride)
    if stride == 1:
        return (d, h, w)
    return ((d - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1)


# ---- BatchNorm statistic slots (csrc/bn.hip) ----------------------------------------------------------------------------
# A train-mode BatchNorm needs, per statistics group, `nslots` zeroed rows [2][C] of fp64 accumulators for its forward
# statistics and the same again for its backward statistics.  Inside a `slot_scope()` (MVSNet / CVPMVSNet.forward open one)
# every layer cuts its rows out of ONE zero-filled arena -- one fill launch per model forward instead of one per layer; the
# backward rows are cut at forward time too and stay zero until the layer's backward uses them (a second backward through the
# same graph gets fresh rows).  The arena is allocated inside the scope, so a captured hipGraph re-zeroes it on every replay.
# Outside a scope each request is its own torch.zeros.
import threading

_SLOT_TLS = threading.local()
_ARENA_DOUBLES = 1 << 18          # 2 MB


class slot_scope:
    def __enter__(self):
        st = getattr(_SLOT_TLS, "state", None)
        if st is None:
            st = _SLOT_TLS.state = {"depth": 0, "arenas": {}}
        st["depth"] += 1
        return self

    def __exit__(self, *exc):
        st = _SLOT_TLS.state
        st["depth"] -= 1
        if st["depth"] == 0:
            st["arenas"].clear()
        return False


def bn_nslots(lib, c: int) -> int:
    n = lib.raw("mvs_bn_slots", c)
    if n <= 0:
        raise ValueError("mvs_amd BatchNorm kernels serve 4/8/16/32/64 channels, got %d" % c)
    return n


def stat_slots(like: torch.Tensor, groups: int, nslots: int, c: int, pieces: int = 1):
    """`pieces` zero-filled fp64 tensors [groups, nslots, 2, c] on like's device."""
    n = groups * nslots * 2 * c
    st = getattr(_SLOT_TLS, "state", None)
    if st is None or st["depth"] == 0:
        return list(torch.zeros((pieces, groups, nslots, 2, c), dtype=torch.float64, device=like.device).unbind(0))
    key = (like.device.type, like.device.index)
    ent = st["arenas"].get(key)
    if ent is None or ent[1] + pieces * n > ent[0].numel():
        ent = st["arenas"][key] = [torch.zeros(max(_ARENA_DOUBLES, pieces * n), dtype=torch.float64, device=like.device), 0]
    off = ent[1]
    ent[1] = off + pieces * n
    return list(ent[0][off:off + pieces * n].view(pieces, groups, nslots, 2, c).unbind(0))


def conv3d_forward(x, weight, stride=1, transposed=False, scale=None, shift=None, skip=None, relu=False,
                   want_stats=False, slots=None, packed_ws=None):
    """Raw C-ABI call.  x [B,Cin,D,H,W] (channels_last_3d).  Returns (y, slots|None).  slots (or want_stats: allocated here):
    zeroed fp64 [nslots, 2, Cout] that receives the BatchNorm statistics (sum, sum of squares per channel) of the raw output,
    spread over nslots rows; packed_ws: this op's weight image, already written (pack_conv3d_weights)."""
    lib = _lib_for(x)
    x = as_cl3(x)
    b, cin, d, h, w = x.shape
    wt = weight.contiguous()
    cout = wt.shape[1] if transposed else wt.shape[0]
    if (wt.shape[0] if transposed else wt.shape[1]) != cin or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError("weight shape %s does not match %d input channels / 3x3x3" % (tuple(wt.shape), cin))
    op = OP_CONVT_FWD if transposed else OP_CONV_FWD
    od, oh, ow = _out_dims(d, h, w, stride, transposed)
    y = empty_cl3(b, cout, od, oh, ow, x)
    ws = _ws(lib, op, b, d, h, w, cin, cout, stride, x) if packed_ws is None else packed_ws
    nslots = 0
    if slots is None and want_stats:
        slots = stat_slots(x, 1, bn_nslots(lib, cout) if cout in (4, 8, 16, 32, 64) else 8, cout, 1)[0][0]
    if slots is not None:
        if slots.dtype != torch.float64 or slots.shape[-1] != cout or slots.shape[-2] != 2 or not slots.is_contiguous():
            raise ValueError("statistic slots must be contiguous float64 [.., nslots, 2, %d], got %s %s" % (cout, slots.dtype, tuple(slots.shape)))
        nslots = slots.shape[-3]
    if skip is not None:
        skip = as_cl3(skip)
        if skip.shape != y.shape:
            raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(y.shape)))
    lib.call("mvs_convT3d_fwd" if transposed else "mvs_conv3d_fwd", _p(x), _p(wt), _p(y), _p(ws), b, d, h, w, cin,
             cout, stride, _p(scale), _p(shift), _p(skip), int(relu), _p(slots), nslots, int(packed_ws is not None), _stream(x),
             tag=_ctag("fwdT" if transposed else "fwd", cin, cout, stride, b, d, h, w))
    return y, slots


_PACK_PLANS = {}   # (op, shape) tuple of a list of layers -> (workspace sizes, ctypes op / shape arrays): the same every training step


def pack_conv3d_weights(items, like):
    """items: list of (op, weight, (B, D, H, W, Cin, Cout, stride)) -- forward / input-gradient ops with the FORWARD op's shape.
    ONE launch writes all their weight images; returns the list of workspace tensors (views of one buffer) to hand to
    conv3d_forward / conv3d_dgrad as ``packed_ws``."""
    lib = _lib_for(like)
    n = len(items)
    key = tuple((op, shape) for op, _, shape in items)
    plan = _PACK_PLANS.get(key)
    if plan is None:
        if len(_PACK_PLANS) > 64:
            _PACK_PLANS.clear()
        sizes = [_ws_floats(lib, op, *shape) for op, _, shape in items]
        plan = _PACK_PLANS[key] = (sizes, (C.c_int * n)(*[op for op, _, _ in items]),
                                   (C.c_int * (7 * n))(*[int(v) for _, _, shape in items for v in shape]))
    sizes, ops_arr, shp_arr = plan
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=like.device)
    views = list(torch.split(buf, sizes))
    ws_ = [wt if wt.is_contiguous() else wt.contiguous() for _, wt, _ in items]
    lib.call("mvs_conv3d_pack_weights_batch", n, ops_arr, _ptr_array(ws_), _ptr_array(views), shp_arr, _stream(like))
    return views


def conv3d_forward_bf16(x, weight, stride=1, transposed=False, scale=None, shift=None, skip=None, relu=False, out_f32=False):
    """Inference-only C-ABI call: x bf16 [B,Cin,D,H,W] (channels_last_3d), fp32 weight -> y bf16 (fp32 if out_f32) with the
    folded BatchNorm / bias / ReLU / skip epilogue.  fp32 accumulation on the bf16 MFMA."""
    lib = _lib_for_bf16(x)
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        raise RuntimeError("mvs_amd: the bf16 regulariser is an inference path (call it under torch.no_grad())")
    x = as_cl3(x)
    b, cin, d, h, w = x.shape
    wt = weight.detach().contiguous().float()
    cout = wt.shape[1] if transposed else wt.shape[0]
    if (wt.shape[0] if transposed else wt.shape[1]) != cin or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError("weight shape %s does not match %d input channels / 3x3x3" % (tuple(wt.shape), cin))
    nbytes = lib.raw("mvs_conv3d_bf16_workspace_bytes", cin, cout, stride, int(transposed))
    if nbytes < 0:
        raise ValueError("conv3d bf16: unsupported channels %d -> %d" % (cin, cout))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    od, oh, ow = _out_dims(d, h, w, stride, transposed)
    y = torch.empty((b, cout, od, oh, ow), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device, memory_format=CL3)
    if skip is not None:
        if skip.dtype != torch.bfloat16 or skip.shape != y.shape:
            raise ValueError("skip must be bf16 with the output's shape %s, got %s %s" % (tuple(y.shape), skip.dtype, tuple(skip.shape)))
        skip = as_cl3(skip)
    scale_c = None if scale is None else scale.contiguous()      # (locals: pointers handed to the C ABI stay valid until the call is made)
    shift_c = None if shift is None else shift.contiguous()
    lib.call("mvs_conv3d_bf16_fwd", _p(x), _p(wt), _p(y), _p(ws), b, d, h, w, cin, cout, stride, int(transposed),
             _p(scale_c), _p(shift_c), _p(skip),
             int(relu), int(out_f32), _stream(x), tag=_ctag("fwdT_bf16" if transposed else "fwd_bf16", cin, cout, stride, b, d, h, w))
    return y


def conv_bn_relu3d_eval_bf16(x, weight, gamma, beta, running_mean, running_var, skip, stride, transposed, eps):
    """Eval-mode ConvBnReLU3D / deconv block on bf16 activations: BatchNorm folded into the conv epilogue."""
    lib = _lib_for_bf16(x)
    cout = weight.shape[1] if transposed else weight.shape[0]
    scale = torch.empty(cout, dtype=torch.float32, device=x.device)
    shift = torch.empty_like(scale)
    lib.call("mvs_bn_eval_affine", _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), cout, _p(scale), _p(shift),
             _stream(x))
    return conv3d_forward_bf16(x, weight, stride, transposed, scale=scale, shift=shift, skip=skip, relu=True)


def conv3d_dgrad(gy, weight, in_shape, stride=1, transposed=False, add=None, bn=None, packed_ws=None):
    """Input gradient of conv3d / conv_transpose3d; ``add`` (same shape as the result, channels_last_3d) is summed into it in the
    kernel's epilogue (the second gradient contribution of a tensor with two consumers: the U-Net skips).  ``bn`` = (raw, stats,
    slots): the result is the COMPLETE output gradient of the BatchNorm+ReLU block whose raw output is ``raw`` (stats [4,C]); its
    backward statistics are added into the zeroed fp64 ``slots`` [.., nslots, 2, C] by the same epilogue."""
    lib = _lib_for(gy)
    gy = as_cl3(gy)
    b, cin, d, h, w = in_shape
    wt = weight.contiguous()
    cout = wt.shape[1] if transposed else wt.shape[0]
    op = OP_CONVT_DGRAD if transposed else OP_CONV_DGRAD
    gx = empty_cl3(b, cin, d, h, w, gy)
    ws = _ws(lib, op, b, d, h, w, cin, cout, stride, gy) if packed_ws is None else packed_ws
    tag = _ctag("dgradT" if transposed else "dgrad", cin, cout, stride, b, d, h, w)
    if add is not None:
        add = as_cl3(add)
        if tuple(add.shape) != tuple(in_shape):
            raise ValueError("conv3d_dgrad: summand shape %s != input shape %s" % (tuple(add.shape), tuple(in_shape)))
    raw = stats = slots = None
    nslots = 0
    if bn is not None:
        raw, stats, slots = bn
        raw = as_cl3(raw)
        if tuple(raw.shape) != tuple(in_shape) or tuple(stats.shape) != (4, cin) or slots.dtype != torch.float64 or \
                slots.shape[-1] != cin or slots.shape[-2] != 2:
            raise ValueError("conv3d_dgrad: BatchNorm operands do not match the input shape %s" % (tuple(in_shape),))
        nslots = slots.shape[-3]
    lib.call("mvs_convT3d_dgrad" if transposed else "mvs_conv3d_dgrad", _p(gy), _p(wt), _p(add), _p(gx), _p(ws), b, d, h, w,
             cin, cout, stride, _p(raw), _p(stats), _p(slots), nslots, int(packed_ws is not None), _stream(gy), tag=tag)
    return gx


def conv3d_wgrad(x, gy, weight_shape, stride=1, transposed=False, on_stream=None):
    """on_stream: a torch stream OTHER than the current one to enqueue the kernels on (the caller has made it wait for the producers
    of x and gy).  The output and the workspace are allocated from the current stream's pool -- no stream switch on the host, which
    costs more than the launch -- and handed to `on_stream` with record_stream, so the allocator does not recycle them early."""
    lib = _lib_for(x)
    x = as_cl3(x)
    gy = as_cl3(gy)
    b, cin, d, h, w = x.shape
    cout = weight_shape[1] if transposed else weight_shape[0]
    op = OP_CONVT_WGRAD if transposed else OP_CONV_WGRAD
    gw = torch.empty(tuple(weight_shape), dtype=torch.float32, device=x.device)
    ws = _ws(lib, op, b, d, h, w, cin, cout, stride, x)
    lib.call("mvs_convT3d_wgrad" if transposed else "mvs_conv3d_wgrad", _p(x), _p(gy), _p(gw), _p(ws), b, d, h, w,
             cin, cout, stride, _stream(x) if on_stream is None else on_stream.cuda_stream,
             tag=_ctag("wgradT" if transposed else "wgrad", cin, cout, stride, b, d, h, w), tstream=on_stream)
    if on_stream is not None:
        for ten in (x, gy, gw, ws):
            ten.record_stream(on_stream)
    return gw


# ---- weight gradients on a side stream (opt-in) -----------------------------------------------------------------
# In the backward pass a layer's weight gradient is a leaf of the dependency graph: nothing needs it before the optimiser
# step, while the input gradient feeds the next layer's backward.  The deep U-Net levels launch fewer workgroups than
# the chip has CUs and every weight gradient ends in two tiny reduction kernels, so running them on a second HIP stream
# lets them fill the gaps of the main chain.  The side stream forks from the main stream when the output gradient is
# ready and is joined by ONE end-of-backward callback (autograd engine), so every consumer AFTER backward() sees
# finished gradients.
#
# That is only correct if nothing reads the gradient DURING the backward pass, which this module cannot prove in
# general: DDP reducer hooks sit on the AccumulateGrad node (invisible from the tensor), nn.DataParallel replicas reduce
# mid-backward.  So it is OFF by default (MVS_ASYNC_WGRAD=1 or set_async_wgrad(True) switches it on -- bench.py does,
# its step reads gradients only after backward()), and even when on a weight takes the synchronous path unless it is a
# contiguous leaf without tensor / post-accumulate hooks, with no existing .grad (accumulation kernel on the main stream)
# and a single forward use in this graph (several uses: the engine sums the contributions mid-backward).
_ASYNC_WGRAD = os.environ.get("MVS_ASYNC_WGRAD", "0") == "1"
# Bookkeeping is per DEVICE, not per thread: the autograd engine runs backward nodes on its own worker threads and the
# final callback on the thread that called backward(), so thread-local state would not connect them; one-thread-per-GPU
# callers (nn.DataParallel style) touch disjoint entries.
_SIDE_STREAMS = {}      # device index -> [side streams] (created once), handed out round-robin
_SIDE_NEXT = {}         # device index -> next pool slot
_N_SIDE = max(1, int(os.environ.get("MVS_WGRAD_STREAMS", "1")))


def set_wgrad_streams(n: int) -> None:
    """Number of side streams the weight gradients are dealt to round-robin (default 1).  The deep U-Net levels' weight gradients
    are launches of 48-250 workgroups on a 256-CU chip and independent of each other: on several streams they run next to each
    other instead of one after the other."""
    global _N_SIDE
    _N_SIDE = max(1, int(n))


_SIDE_PRIORITY = os.environ.get("MVS_SIDE_PRIORITY", "default")   # "low": the lowest stream priority the device offers


def set_side_stream_priority(which: str) -> None:
    """"low": the side streams are created with the lowest priority of the device (torch.cuda.Stream.priority_range()), so the
    main stream's kernels -- the critical path -- are dispatched first and the weight gradients fill what is left; "default": the
    priority of an ordinary stream.  Drops the existing pool (new streams are made on demand)."""
    global _SIDE_PRIORITY
    if which not in ("low", "default", "high"):
        raise ValueError("side stream priority must be 'low', 'default' or 'high'")
    _SIDE_PRIORITY = which
    for idx, pool in _SIDE_STREAMS.items():
        for sd in pool:
            sd.synchronize()
    _SIDE_STREAMS.clear()


def _new_side_stream(dev):
    if _SIDE_PRIORITY == "low":
        try:
            least = torch.cuda.Stream.priority_range()[0]     # (least, greatest): numerically larger = lower priority
            return torch.cuda.Stream(device=dev, priority=least)
        except Exception:
            pass
    if _SIDE_PRIORITY == "high":
        try:
            return torch.cuda.Stream(device=dev, priority=torch.cuda.Stream.priority_range()[1])
        except Exception:
            pass
    return torch.cuda.Stream(device=dev)


def _side_stream(dev):
    idx = dev.index
    pool = _SIDE_STREAMS.setdefault(idx, [])
    while len(pool) < _N_SIDE:
        pool.append(_new_side_stream(dev))
    k = _SIDE_NEXT.get(idx, 0) % _N_SIDE
    _SIDE_NEXT[idx] = k + 1
    return pool[k]


JOIN_TRACE = None        # diagnostics (bench.py --step-events): a list -> (main-stream event, side-stream event) recorded right before each join


# Tensors the side stream is still reading / writing when the node that allocated them returns (deferred join): kept ALIVE here until
# the join instead of being handed to Tensor.record_stream.  record_stream makes the caching allocator poll an event before it may
# reuse the block; whether the block is free when the next request comes then depends on timing, the pool fragments (a 500 MB block
# split for a 126 MB request, the next 500 MB request then needs a fresh hipMalloc) and keeps GROWING: 6-18 hipMalloc calls inside
# bench.py's 20 timed steps, 10-21 GB reserved for a 1.6 GB step, launch-thread stalls of 16-29 ms each
# (profiles/r06_run20_allocator_before.txt).  Freed at the join, the blocks go back to the main stream's pool in stream order: the
# same blocks serve the same requests every step.  Keyed by (device, main stream): the join of one host thread's stream must not
# release what another thread's side-stream work still uses.
_HELD = {}


def _hold(idx, main, tensors) -> None:
    _HELD.setdefault((idx, main.cuda_stream), []).extend(tensors)


def _join_side(main, idx) -> None:
    for sd in _SIDE_STREAMS.get(idx, ()):
        if JOIN_TRACE is not None:
            em, es = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            em.record(main)
            es.record(sd)
            JOIN_TRACE.append((em, es))
        main.wait_stream(sd)
    _HELD.pop((idx, main.cuda_stream), None)     # everything enqueued on `main` from here on is ordered behind the side streams' work
_WEIGHT_USES = {}       # device index -> {weight data_ptr: forward uses whose backward node has not run yet}
_WEIGHT_MULTI = {}      # device index -> {weight data_ptr} that had > 1 outstanding use at some point (until all of them have run)
_BWD_OPEN = {}          # device index -> [main stream, side stream used?] while a backward pass with our nodes is running
_DEFER_JOIN = os.environ.get("MVS_WGRAD_DEFER_JOIN", "0") == "1"
_WGRAD_FORK_EARLY = int(os.environ.get("MVS_WGRAD_FORK_EARLY", "0"))   # 0: after the block's input gradient, 1: before it (not conv0), 2: always before


def set_async_wgrad(flag: bool, defer_join=None) -> None:
    """Weight gradients on a side stream: the per-layer Functions (off by default: unsafe under gradient hooks) AND the fused
    regulariser node (on by default: joined before its gradients are returned).
    defer_join=True (opt-in, MVS_WGRAD_DEFER_JOIN=1): the fused node hands its weight gradients to autograd WITHOUT waiting for the
    side stream and the join happens once, at the end of the backward pass -- conv0's weight gradient (0.6 ms, the last one
    forked) then runs under the plane-sweep backward and the 2-D extractor's backward instead of holding the main stream.  Only for
    training loops that read gradients after backward() returns (no DDP / DataParallel gradient hooks, no pre-existing .grad to
    accumulate into): bench.py's loop is one."""
    global _ASYNC_WGRAD, _ASYNC_WGRAD_FUSED, _DEFER_JOIN
    _ASYNC_WGRAD = bool(flag)
    _ASYNC_WGRAD_FUSED = bool(flag)
    if defer_join is not None:
        _DEFER_JOIN = bool(defer_join)


def _note_weight_use(weight: torch.Tensor) -> None:
    if _ASYNC_WGRAD and weight.is_cuda:
        idx = weight.device.index
        if idx in _BWD_OPEN:
            # a forward while a backward pass is still "open": that pass died before its end-of-backward callback ran
            # (an exception in a later node).  Close it here so that state never leaks into the next step.
            _end_of_backward(idx)
        uses = _WEIGHT_USES.setdefault(idx, {})
        n = uses[weight.data_ptr()] = uses.get(weight.data_ptr(), 0) + 1
        if n > 1:
            _WEIGHT_MULTI.setdefault(idx, set()).add(weight.data_ptr())


def reset_weight_uses() -> None:
    """Forget the outstanding forward uses (per weight) the side-stream rules count.  A forward pass whose weight-gradient
    backward never runs (a train-mode validation pass without no_grad, autograd.grad w.r.t. inputs only, an exception) leaves a
    count behind and that weight then stays on the synchronous path -- safe, but slower; a training loop that does such passes
    calls this at the start of an iteration (no backward pass may be open)."""
    if _BWD_OPEN:
        raise RuntimeError("reset_weight_uses() inside a backward pass")
    _WEIGHT_USES.clear()
    _WEIGHT_MULTI.clear()


def _weight_use_done(idx: int, ptr: int) -> None:
    """One backward node of this weight has taken its decision.  The counts are kept per OUTSTANDING use, not reset per backward
    pass: with two graphs built before either backward (forward A, forward B, A.backward(), B.backward()) a reset at the end of A's
    pass made graph B's shared weights (CVP's regulariser) look single-use.  A weight stays in the multi-use set until
    every outstanding use has run; a forward that never gets a backward only ever makes the weight take the synchronous path."""
    uses = _WEIGHT_USES.get(idx)
    if uses is None or ptr not in uses:
        return
    uses[ptr] -= 1
    if uses[ptr] <= 0:
        del uses[ptr]
        _WEIGHT_MULTI.get(idx, set()).discard(ptr)


def _end_of_backward(idx: int) -> None:
    ent = _BWD_OPEN.pop(idx, None)
    if ent is not None and ent[1]:
        _join_side(ent[0], idx)


# Round 6: the deferred join as the SAFE library default.  `set_async_wgrad(True, defer_join=True)` (round 5, bench-only) hands the
# regulariser's weight gradients to autograd while the side stream is still writing them and joins once at the end of the backward
# pass -- unsafe under anything that READS a gradient when AccumulateGrad fires (DDP / DataParallel hooks are C++ hooks no Python test
# can see), so the library default joined inside the node and conv0's weight gradient (0.6 ms) held the main stream (5.05 vs 4.75 ms).
# The tail node makes the late join safe: an identity autograd node on the regulariser's convolution weights, CREATED FIRST in the
# model's forward (MVSNet._forward), whose backward joins the side streams and only then passes the gradients on to AccumulateGrad.
# The autograd engine runs ready nodes in order of creation, newest first, so the tail node runs when nothing newer is left -- after the
# plane-sweep backward and the 2-D extractor's backward, i.e. at the end of the pass -- and every gradient any hook ever sees is finished.
# (If the engine ran it earlier, the join would simply come earlier: correctness does not depend on the order, only the overlap does.)
TAIL_JOIN = os.environ.get("MVS_TAIL_JOIN", "1") != "0"


class DeferredJoinFn(torch.autograd.Function):
    """identity on a list of weights; backward: join the device's weight-gradient side streams, then hand the gradients on"""

    @staticmethod
    def forward(ctx, *ws):
        return tuple(w.view_as(w) for w in ws)

    @staticmethod
    def backward(ctx, *gs):
        for g in gs:
            if g is not None and g.is_cuda:
                idx = g.device.index
                _join_side(torch.cuda.current_stream(g.device), idx)
                ent = _BWD_OPEN.get(idx)
                if ent is not None:
                    ent[1] = False                # joined: the end-of-backward callback has nothing left to wait for
                break
        return gs


def tail_join_views(weights):
    """{id(weight): view} through ONE DeferredJoinFn node for the weights that need a gradient (call it before anything else of the
    forward pass builds autograd nodes); {} when the tail join does not apply (CPU tensors, no gradient, switched off)."""
    ws = [w for w in weights if w.requires_grad and w.is_cuda]
    if not (TAIL_JOIN and _ASYNC_WGRAD_FUSED and ws and torch.is_grad_enabled()):
        return {}
    views = DeferredJoinFn.apply(*ws)
    for v in views:
        v._mvs_tail_join = True
    return {id(w): v for w, v in zip(ws, views)}


def _async_safe(weight: torch.Tensor) -> bool:
    return (weight.is_leaf and weight.grad is None and weight.is_contiguous()
            and not getattr(weight, "_backward_hooks", None)
            and not getattr(weight, "_post_accumulate_grad_hooks", None))


def _wgrad_maybe_async(x, gy, weight, stride, transposed):
    lib = _lib_for(x)
    if not (_ASYNC_WGRAD and x.is_cuda):
        return conv3d_wgrad(x, gy, tuple(weight.shape), stride, transposed)
    idx = x.device.index
    main = torch.cuda.current_stream(x.device)
    ent = _BWD_OPEN.get(idx)
    if ent is None:
        # first weight gradient of this backward pass: ONE callback closes the pass (joins the side stream if it was used,
        # resets the per-graph use counts) whichever path the individual layers take
        ent = _BWD_OPEN[idx] = [main, False]
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _end_of_backward(idx))
    ok = _async_safe(weight) and weight.data_ptr() not in _WEIGHT_MULTI.get(idx, ())
    _weight_use_done(idx, weight.data_ptr())
    if not ok:
        return conv3d_wgrad(x, gy, tuple(weight.shape), stride, transposed)
    side = _side_stream(x.device)
    x, gy = as_cl3(x), as_cl3(gy)                # (a layout copy, if one is needed, runs on the main stream: before the fork)
    side.wait_stream(main)                       # gy was produced on the main stream
    gw = conv3d_wgrad(x, gy, tuple(weight.shape), stride, transposed, on_stream=side)
    ent[1] = True
    return gw


def bn_relu_fwd_slots(x, slots, gamma, beta, running_mean, running_var, eps, momentum, skip=None, relu=True, groups=1):
    """y = relu(BatchNorm_train(x)) (+ skip) from the statistic slots of x (written by the kernel that produced x); -> (y, stats
    [groups, 4, C]: mean, invstd, scale, shift).  x channels-last [B,C,...]; the groups are equal chunks of the batch."""
    lib = _lib_for(x)
    fmt = CL2 if x.dim() == 4 else CL3
    x = x.contiguous(memory_format=fmt)
    if skip is not None:
        skip = skip.contiguous(memory_format=fmt)
    c = x.shape[1]
    vg = x.numel() // c // groups
    stats = torch.empty((groups, 4, c), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x, memory_format=fmt)
    lib.call("mvs_bn_relu_fwd_slots", _p(x), _p(slots), slots.shape[-3], groups, vg, c, _p(gamma), _p(beta), float(eps),
             float(momentum), _p(running_mean), _p(running_var), _p(skip), int(relu), _p(stats), _p(y), _stream(x))
    return y, stats


def bn_finalize_slots(slots, gamma, beta, running_mean, running_var, eps, momentum, count_per_group):
    """stats [groups, 4, C] (mean, invstd, scale, shift) + running-statistics update from the statistic slots [groups, nslots, 2, C]
    of a tensor with `count_per_group` elements per channel and group -- bn_relu_fwd_slots without its elementwise pass (the
    consumer applies the normalisation while it stages its input: conv2d_forward(in_stats=...), conv2d_wgrad_batch(x_stats=...))."""
    lib = _lib_for(gamma)
    groups, nslots, _, c = slots.shape
    stats = torch.empty((groups, 4, c), dtype=torch.float32, device=gamma.device)
    lib.call("mvs_bn_finalize_slots", _p(slots), nslots, groups, int(count_per_group), c, _p(gamma), _p(beta), float(eps), float(momentum),
             _p(running_mean), _p(running_var), _p(stats), _stream(gamma))
    return stats


def bn_relu_bwd_slots(gy, x, stats, slots, have_stats, relu=True, groups=1):
    """BatchNorm(+ReLU) backward: (dx, dgamma, dbeta).  have_stats: the slots already hold (sum dyh, sum dyh*xhat) -- an
    input-gradient epilogue wrote them (conv3d_dgrad(bn=...)); otherwise one reduction pass over (gy, x) fills them first."""
    lib = _lib_for(x)
    fmt = CL2 if x.dim() == 4 else CL3
    x, gy = x.contiguous(memory_format=fmt), gy.contiguous(memory_format=fmt)
    c = x.shape[1]
    vg = x.numel() // c // groups
    st = _stream(x)
    if not have_stats:
        lib.call("mvs_bn_bwd_reduce_slots", _p(gy), _p(x), _p(stats), int(relu), groups, vg, c, _p(slots), slots.shape[-3], st)
    dx = torch.empty_like(x, memory_format=fmt)
    dgb = torch.empty((2, c), dtype=torch.float32, device=x.device)
    lib.call("mvs_bn_relu_bwd_slots", _p(gy), _p(x), _p(stats), _p(slots), slots.shape[-3], int(relu), groups, vg, c, _p(dx),
             _p(dgb[0]), _p(dgb[1]), st)
    return dx, dgb[0], dgb[1]


class ConvBnReLU3dFn(torch.autograd.Function):
    """conv3d | conv_transpose3d (bias-free, k3 p1) -> BatchNorm3d -> ReLU (-> + skip, after the ReLU).

    Train: batch statistics (slots filled by the conv epilogue, finished in the apply kernel's prologue), running stats
    updated in place.
    Eval : BatchNorm folded into the conv epilogue (no gradient support -- the reference only
    evaluates under no_grad, jdacs/eval.py:143)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, skip, stride, transposed, training, eps,
                momentum):
        lib = _lib_for(x)
        st = _stream(x)
        x = as_cl3(x)
        cout = weight.shape[1] if transposed else weight.shape[0]
        dev = x.device
        if not training:
            scale = torch.empty(cout, dtype=torch.float32, device=dev)
            shift = torch.empty_like(scale)
            lib.call("mvs_bn_eval_affine", _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), cout,
                     _p(scale), _p(shift), st)
            y, _ = conv3d_forward(x, weight, stride, transposed, scale=scale, shift=shift, skip=skip, relu=True)
            ctx.eval_mode = True
            return y
        if ctx.needs_input_grad[1]:
            _note_weight_use(weight)
        slots_f, slots_b = stat_slots(x, 1, bn_nslots(lib, cout), cout, 2)
        raw, _ = conv3d_forward(x, weight, stride, transposed, slots=slots_f)
        skip_c = None if skip is None else as_cl3(skip)
        y, stats = bn_relu_fwd_slots(raw, slots_f, gamma, beta, running_mean, running_var, eps, momentum, skip_c)
        ctx.save_for_backward(x, weight, raw, stats, slots_b)
        ctx.cfg = (stride, transposed, skip is not None, cout)
        ctx.eval_mode = False
        ctx.slots_used = False
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.eval_mode:
            raise NotImplementedError("mvs_amd: backward through eval-mode (folded) BatchNorm is not supported; "
                                      "call .train() for training or use torch.no_grad() for inference")
        x, weight, raw, stats, slots_b = ctx.saved_tensors
        stride, transposed, has_skip, cout = ctx.cfg
        gy = as_cl3(gy)
        if ctx.slots_used:                       # a second backward through the same graph: fresh accumulators
            slots_b = torch.zeros_like(slots_b)
        ctx.slots_used = True
        draw, dgamma, dbeta = bn_relu_bwd_slots(gy, raw, stats[0], slots_b, False)
        gx = conv3d_dgrad(draw, weight, tuple(x.shape), stride, transposed) if ctx.needs_input_grad[0] else None
        gw = _wgrad_maybe_async(x, draw, weight, stride, transposed) if ctx.needs_input_grad[1] else None
        gskip = gy if has_skip else None
        return gx, gw, dgamma, dbeta, None, None, gskip, None, None, None, None, None


# ---- the whole regulariser as ONE autograd node ----------------------------------------------------------------------------
# jdacs/models/mvsnet.py:37-74 and jdacs-ms/models/network.py:44-74 are short straight-line programs of ConvBnReLU3D /
# Deconv+BN+ReLU blocks with skips added after the ReLU, closed by the bias-only `prob` convolution.  Running one program
# through ONE torch.autograd.Function buys what the per-layer graph cannot give:
#  * the second gradient contribution of a skip source is summed in the EPILOGUE of the input-gradient kernel of its other
#    consumer (conv3d_dgrad(add=...)) instead of autograd's out-of-place add (three passes over the 126 MB level-0 tensor);
#  * round 4: the BatchNorm BACKWARD statistics of a block are summed in the epilogue of the input-gradient kernel that writes the
#    block's complete output gradient (conv3d_dgrad(bn=...)): no reduction pass over (dy, raw) -- and all BatchNorm statistics,
#    forward and backward, are finished in the prologue of the kernel that applies them (csrc/bn.hip): a block costs conv + apply
#    forward and apply + dgrad backward, where rounds 1-3 ran conv, finalize, apply / reduce, finalize, apply, dgrad;
#  * round 4: the MFMA weight images of all forward and input-gradient convolutions are packed by ONE launch per step;
#  * weight gradients run on a side HIP stream BY DEFAULT: they are handed to autograd only after the side stream has been joined,
#    at the end of this node's backward, so DataParallel / DDP hooks (which fire when a gradient is RETURNED) can never observe an
#    unfinished one -- the per-layer form had to leave that off (see _wgrad_maybe_async);
#  * one node instead of ~25 on the autograd tape.
# MVS_REG_FUSED=0 (or ops.FUSED_REGULARISER = False) restores the per-layer graph; both are tested against the same goldens.
FUSED_REGULARISER = os.environ.get("MVS_REG_FUSED", "1") != "0"
_ASYNC_WGRAD_FUSED = os.environ.get("MVS_ASYNC_WGRAD", "1") != "0"


# ---- one C call per pass (csrc/unet_pass.cpp: mvs_unet_fwd / mvs_unet_bwd) -------------------------------------------------
# The node below issued ~45 C-ABI calls forward and ~55 backward from Python (~1.7 ms of launch-thread time per training step);
# with C_ENTRY the same launch sequence is ONE call per pass over pointer tables into three arenas the node allocates (activations;
# backward work space; weight-gradient partial images).  MVS_REG_C_ENTRY=0 / ops.C_ENTRY = False keeps the per-layer calls (they
# also run whenever a KernelTimer is attached: bench.py's per-kernel HIP-event brackets live in _lib.MvsLib.call).
C_ENTRY = os.environ.get("MVS_REG_C_ENTRY", "1") != "0"
_UNET_PLANS = {}


class _UnetPlan:
    """Everything about a (program, input shape) that is the same every step: the block table, tensor sizes and arena offsets."""

    def __init__(self, lib, prog, xshape, wshapes, prob_cout):
        n = len(prog)
        b, c, d, h, w = xshape
        self.n, self.B = n, b
        self.blocks = (_lib.MvsUnetBlock * n)()
        dims = {-1: (c, d, h, w)}
        self.shapes, self.out, self.sizes = [], [], []
        for i, (transposed, stride, src, skip, eps, momentum) in enumerate(prog):
            cin, di, hi, wi = dims[src]
            cout = wshapes[i][1] if transposed else wshapes[i][0]
            od, oh, ow = _out_dims(di, hi, wi, stride, transposed)
            dims[i] = (cout, od, oh, ow)
            blk = self.blocks[i]
            blk.transposed, blk.stride, blk.src, blk.skip = int(bool(transposed)), int(stride), int(src), int(skip)
            blk.eps, blk.momentum = float(eps), float(momentum)
            blk.cin, blk.cout, blk.d, blk.h, blk.w = cin, cout, di, hi, wi
            self.shapes.append((b, di, hi, wi, cin, cout, stride))
            self.out.append((b, cout, od, oh, ow))
            self.sizes.append(b * cout * od * oh * ow)
        cq, dq, hq, wq = dims[n - 1]
        self.pshape = (b, dq, hq, wq, cq, prob_cout, 1)
        self.logits_shape = (b, prob_cout, dq, hq, wq)
        # the C entry needs no explicit gradient add: every block feeds at most one block through its INPUT, and a skip contribution
        # (consumer k) reaches its source before the source's input-side consumer (i < k: blocks run last to first) does
        src_users, skip_users = {}, {}
        for i, p_ in enumerate(prog):
            if p_[2] >= 0:
                src_users.setdefault(p_[2], []).append(i)
            if p_[3] >= 0:
                skip_users.setdefault(p_[3], []).append(i)
        # mirrors mvs_unet_bwd's preconditions exactly, so that an unsupported program takes the per-layer path at FORWARD
        # time instead of raising in backward(): a block is the input of at most one block and the skip operand of at most one, and the
        # skip consumer runs first in the backward order (a larger index than the input consumer).  Several readers of the volume x
        # are fine: mvs_unet_bwd accumulates gx.
        self.ok = n <= 32 and all(len(v) == 1 for v in src_users.values()) and all(len(v) == 1 for v in skip_users.values()) and all(
            not src_users.get(j) or src_users[j][0] < v[0] for j, v in skip_users.items())
        # forward arena (floats): raw_i, y_i, stats_i ... ; 64-float alignment
        al = lambda v: (v + 63) // 64 * 64
        off = 0
        self.raw_off, self.y_off, self.stats_off = [], [], []
        for i in range(n):
            self.raw_off.append(off); off += al(self.sizes[i])
            self.y_off.append(off); off += al(self.sizes[i])
            self.stats_off.append(off); off += al(4 * self.out[i][1])
        self.ws_prob_off = off
        off += al(_ws_floats(lib, OP_CONV_FWD, *self.pshape))
        self.fwd_floats = off
        # statistic slots (doubles): forward and backward rows of every block
        self.nslots = (C.c_int * n)(*[bn_nslots(lib, self.out[i][1]) for i in range(n)])
        soff = 0
        self.sf_off, self.sb_off = [], []
        for i in range(n):
            cnt = self.nslots[i] * 2 * self.out[i][1]
            self.sf_off.append(soff); soff += cnt
            self.sb_off.append(soff); soff += cnt
        self.slot_doubles = soff
        # backward work arena: gbuf_i, draw_i; dgamma / dbeta; weight-gradient workspaces
        off = 0
        self.g_off, self.draw_off = [], []
        for i in range(n):
            self.g_off.append(off); off += al(self.sizes[i])
            self.draw_off.append(off); off += al(self.sizes[i])
        self.bwd_floats = off
        off = 0
        self.wws_off = []
        for i in range(n):
            op = OP_CONVT_WGRAD if prog[i][0] else OP_CONV_WGRAD
            self.wws_off.append(off); off += al(_ws_floats(lib, op, *self.shapes[i]))
        self.wws_off.append(off); off += al(_ws_floats(lib, OP_CONV_WGRAD, *self.pshape))
        self.wws_floats = off
        self.dgb_off = []
        off = 0
        for i in range(n):
            self.dgb_off.append(off); off += 2 * self.out[i][1]
        self.dgb_floats = off


def _unet_plan(lib, prog, xshape, wshapes, prob_cout):
    key = (prog, tuple(xshape), tuple(wshapes), prob_cout)
    plan = _UNET_PLANS.get(key)
    if plan is None:
        if len(_UNET_PLANS) > 16:
            _UNET_PLANS.clear()
        plan = _UNET_PLANS[key] = _UnetPlan(lib, prog, xshape, wshapes, prob_cout)
    return plan


def _c_entry_allowed(lib, plan, prog, wp) -> bool:
    """The C entry issues no per-layer Python call, so a KernelTimer cannot bracket its kernels from outside: it runs when no timer is
    attached, when the timer looks at none of the regulariser's entry points, or when the timer wants exactly one weight gradient (then
    mvs_unet_bwd brackets that launch itself: _lib.KernelTimer.allows_c_entry)."""
    if _N_SIDE > 1 or _WGRAD_FORK_EARLY != 0:       # knobs only the per-layer path implements (round-robin side streams, early fork)
        return False
    prof = lib.profiler
    if prof is None:
        owner = getattr(lib, "_unet_timer", None)
        if owner is not None:                       # a timer switched the C-side bracket on and has been detached since
            owner.release_c_bracket(lib)
        return True
    allows = getattr(prof, "allows_c_entry", None)
    if allows is None:
        return False
    blocks = [(bool(prog[i][0]), sh[4], sh[5], sh[6], sh[:4]) for i, sh in enumerate(plan.shapes)]
    blocks.append((False, plan.pshape[4], plan.pshape[5], 1, plan.pshape[:4]))
    return bool(allows(lib, blocks))


def _ptrs(base, offs, itemsize=4):
    arr = (C.c_void_p * len(offs))()
    for i, o in enumerate(offs):
        arr[i] = base + o * itemsize
    return arr


class UNetRegulariserFn(torch.autograd.Function):
    """x [B,C,D,H,W] -> logits [B,1,D,H,W].  ``prog``: tuple of (transposed, stride, src, skip, eps, momentum) per Conv/Deconv+BN+ReLU
    block (src / skip = index of the block whose output is this block's input / is added after the ReLU; -1 = the volume x / no skip).
    ``params``: per block weight, gamma, beta, running_mean, running_var; then the prob layer's weight and bias.  Train mode only."""

    @staticmethod
    def forward(ctx, x, prog, *params):
        lib = _lib_for(x)
        x = as_cl3(x)
        n = len(prog)
        need = ctx.needs_input_grad          # [x, prog, *params]
        wp, bp = params[5 * n], params[5 * n + 1]
        # ---- shapes of every block's input, then ONE launch packs all weight images (forward + input gradient) ----
        shapes, in_shape = [], {-1: tuple(x.shape)}
        for i, (transposed, stride, src, skip, eps, momentum) in enumerate(prog):
            w = params[5 * i]
            b, cin, d, h, wd = in_shape[src]
            cout = w.shape[1] if transposed else w.shape[0]
            shapes.append((b, d, h, wd, cin, cout, stride))
            in_shape[i] = (b, cout) + _out_dims(d, h, wd, stride, transposed)
        bq, cq, dq, hq, wq = in_shape[n - 1]
        pshape = (bq, dq, hq, wq, cq, wp.shape[0], 1)
        any_grad = any(need)
        # every convolution weight that gets a gradient came through the tail node (DeferredJoinFn): the join may be left to it
        ctx.tail_join = all(getattr(params[k], "_mvs_tail_join", False) or not need[2 + k] for k in list(range(0, 5 * n, 5)) + [5 * n])
        items = [(OP_CONVT_FWD if prog[i][0] else OP_CONV_FWD, params[5 * i], shapes[i]) for i in range(n)]
        dg_index = {}
        if any_grad:
            for i in range(n):
                if prog[i][2] >= 0 or need[0]:
                    dg_index[i] = len(items)
                    items.append((OP_CONVT_DGRAD if prog[i][0] else OP_CONV_DGRAD, params[5 * i], shapes[i]))
            dg_index[n] = len(items)
            items.append((OP_CONV_DGRAD, wp, pshape))
        packed = pack_conv3d_weights(items, x)
        plan = _unet_plan(lib, prog, tuple(x.shape), tuple(tuple(params[5 * i].shape) for i in range(n)), wp.shape[0]) if C_ENTRY else None
        if plan is not None and plan.ok and _c_entry_allowed(lib, plan, prog, wp):
            dev = x.device
            arena = torch.empty(plan.fwd_floats, dtype=torch.float32, device=dev)
            (slots,) = stat_slots(x, 1, 1, plan.slot_doubles // 2 + 1, 1)      # one zero-filled run of doubles for every block's rows
            logits = empty_cl3(*plan.logits_shape, x)
            ab, sb = arena.data_ptr(), slots.data_ptr()
            ws_c = [params[5 * i] if params[5 * i].is_contiguous() else params[5 * i].contiguous() for i in range(n)]
            bpc = bp.contiguous()
            lib.call("mvs_unet_fwd", n, plan.blocks, plan.B, _p(x), _ptr_array(ws_c), _ptr_array([params[5 * i + 1] for i in range(n)]),
                     _ptr_array([params[5 * i + 2] for i in range(n)]), _ptr_array([params[5 * i + 3] for i in range(n)]),
                     _ptr_array([params[5 * i + 4] for i in range(n)]), _ptr_array(packed[:n]), _ptrs(ab, plan.raw_off), _ptrs(ab, plan.y_off),
                     _ptrs(ab, plan.stats_off), _ptrs(sb, plan.sf_off, 8), plan.nslots, _p(wp), _p(bpc), wp.shape[0],
                     ab + 4 * plan.ws_prob_off, _p(logits), _stream(x))
            ctx.prog, ctx.plan, ctx.dg_index, ctx.c_entry, ctx.slots_used = prog, plan, dg_index, True, False
            # the deferred join is decided on the PARAMETERS (a .contiguous() copy always looks like a hook-free leaf: ADVICE r5)
            ctx.params_w = [params[5 * i] for i in range(n)] + [wp]
            ctx.save_for_backward(x, wp, *ws_c, arena, slots, *packed)
            return logits
        ctx.c_entry = False
        ys, raws, statss, slots_b = [], [], [], []
        for i, (transposed, stride, src, skip, eps, momentum) in enumerate(prog):
            w, gamma, beta, rmean, rvar = params[5 * i:5 * i + 5]
            xin = x if src < 0 else ys[src]
            cout = shapes[i][5]
            sf, sb = stat_slots(x, 1, bn_nslots(lib, cout), cout, 2)
            raw, _ = conv3d_forward(xin, w, stride, transposed, slots=sf, packed_ws=packed[i])
            y, stats = bn_relu_fwd_slots(raw, sf, gamma, beta, rmean, rvar, eps, momentum, ys[skip] if skip >= 0 else None)
            ys.append(y)
            raws.append(raw)
            statss.append(stats[0])
            slots_b.append(sb)
        logits, _ = conv3d_forward(ys[-1], wp, 1, False, shift=bp.contiguous())
        ctx.prog = prog
        ctx.shapes = shapes
        ctx.dg_index = dg_index
        ctx.npacked = len(packed)
        ctx.slots_used = False
        ctx.save_for_backward(x, wp, *[params[5 * i] for i in range(n)], *ys, *raws, *statss, *slots_b, *packed)
        return logits

    @staticmethod
    def _backward_c(ctx, glogits):
        """the whole backward pass as ONE C call (mvs_unet_bwd): same kernels, same order, same fork points as backward() below"""
        prog, plan, dg_index = ctx.prog, ctx.plan, ctx.dg_index
        n = plan.n
        sv = ctx.saved_tensors
        x, wp, ws_, arena, slots = sv[0], sv[1], sv[2:2 + n], sv[2 + n], sv[3 + n]
        packed = sv[4 + n:]
        lib = _lib_for(x)
        dev = x.device
        need = ctx.needs_input_grad          # [x, prog, *params]
        if ctx.slots_used:                   # a second backward through the same graph: fresh backward accumulators
            slots = torch.zeros_like(slots)
        ctx.slots_used = True
        gy = as_cl3(glogits)
        work = torch.empty(plan.bwd_floats, dtype=torch.float32, device=dev)
        wws = torch.empty(plan.wws_floats, dtype=torch.float32, device=dev)
        dgb = torch.empty(plan.dgb_floats, dtype=torch.float32, device=dev)
        gx = empty_cl3(*x.shape, x) if need[0] else None
        gws = [torch.empty(tuple(wt.shape), dtype=torch.float32, device=dev) if need[2 + 5 * i] else None for i, wt in enumerate(ws_)]
        gws.append(torch.empty(tuple(wp.shape), dtype=torch.float32, device=dev) if need[2 + 5 * n] else None)
        main = torch.cuda.current_stream(dev) if x.is_cuda else None
        use_side = _ASYNC_WGRAD_FUSED and x.is_cuda
        side = _side_stream(dev) if use_side else None
        deferred = bool(use_side and (ctx.tail_join or (_DEFER_JOIN and all(gw is None or _async_safe(wt) for gw, wt in zip(gws, ctx.params_w)))))
        ab, sb, wb = arena.data_ptr(), slots.data_ptr(), work.data_ptr()
        pd = (C.c_void_p * (n + 1))()
        for i in range(n + 1):
            if i in dg_index:
                pd[i] = packed[dg_index[i]].data_ptr()
        gwp = (C.c_void_p * (n + 1))()
        for i, t in enumerate(gws):
            if t is not None:
                gwp[i] = t.data_ptr()
        used = C.c_int(0)
        lib.call("mvs_unet_bwd", n, plan.blocks, plan.B, _p(x), _ptr_array(ws_), _p(wp), wp.shape[0], _ptrs(ab, plan.y_off),
                 _ptrs(ab, plan.raw_off), _ptrs(ab, plan.stats_off), _ptrs(sb, plan.sb_off, 8), plan.nslots, pd, _p(gy),
                 _ptrs(wb, plan.g_off), _ptrs(wb, plan.draw_off), _p(gx), gwp, _ptrs(wws.data_ptr(), plan.wws_off),
                 _ptrs(dgb.data_ptr(), plan.dgb_off), _ptrs(dgb.data_ptr(), [o + plan.out[i][1] for i, o in enumerate(plan.dgb_off)]),
                 main.cuda_stream if main is not None else None, side.cuda_stream if side is not None else None,
                 0 if deferred else 1, C.byref(used))
        grads = [None] * (5 * n + 2)
        for i in range(n):
            c = plan.out[i][1]
            grads[5 * i] = gws[i]
            grads[5 * i + 1] = dgb[plan.dgb_off[i]:plan.dgb_off[i] + c]
            grads[5 * i + 2] = dgb[plan.dgb_off[i] + c:plan.dgb_off[i] + 2 * c]
        grads[5 * n] = gws[n]
        if need[2 + 5 * n + 1]:
            grads[5 * n + 1] = gy.sum().reshape(1) if gy.shape[1] == 1 else gy.sum(dim=(0, 2, 3, 4))
        if used.value and side is not None:
            # (not deferred: mvs_unet_bwd has made the main stream wait for the side stream before it returned -- nothing to keep)
            if deferred:                       # ONE join at the end of the whole backward pass (tail node / autograd engine callback)
                idx = dev.index
                _hold(idx, main, [x, arena, work, wws, gy])       # (the weight gradients themselves go to autograd and outlive the join)
                ent = _BWD_OPEN.get(idx)
                if ent is None:
                    ent = _BWD_OPEN[idx] = [main, False]
                    torch.autograd.Variable._execution_engine.queue_callback(lambda: _end_of_backward(idx))
                ent[1] = True
        return (gx, None) + tuple(grads)

    @staticmethod
    def backward(ctx, glogits):
        if ctx.c_entry:
            return UNetRegulariserFn._backward_c(ctx, glogits)
        prog = ctx.prog
        n = len(prog)
        sv = ctx.saved_tensors
        x, wp = sv[0], sv[1]
        ws_, ys, raws, statss, slots_b = (sv[2 + k * n:2 + (k + 1) * n] for k in range(5))
        packed = sv[2 + 5 * n:]
        dg_index = ctx.dg_index
        lib = _lib_for(x)
        dev = x.device
        need = ctx.needs_input_grad          # [x, prog, *params]
        main = torch.cuda.current_stream(dev) if x.is_cuda else None
        side_used = [False]
        if ctx.slots_used:                   # a second backward through the same graph: fresh accumulators
            slots_b = [torch.zeros_like(t) for t in slots_b]
        ctx.slots_used = True

        def wgrad(xin, gout, weight, stride, transposed, wanted):
            """weight gradient, on the side stream when allowed (joined before this node returns)"""
            if not wanted:
                return None
            # (a call the KernelTimer brackets is bracketed on the stream it runs on -- _lib.MvsLib.call records its events on the
            #  current stream --, so a profiled weight gradient stays on the side stream and its duration includes whatever shares
            #  the chip with it; rounds 1-3 moved profiled calls to the main stream, which took the overlap out of the timed step)
            use_side = _ASYNC_WGRAD_FUSED and x.is_cuda
            if not use_side:
                return conv3d_wgrad(xin, gout, tuple(weight.shape), stride, transposed)
            side = _side_stream(dev)
            side.wait_stream(main)                       # gout was produced on the main stream
            gw = conv3d_wgrad(xin, gout, tuple(weight.shape), stride, transposed, on_stream=side)
            side_used[0] = True
            return gw

        # the consumer that contributes LAST to a block's output gradient (blocks run last to first; within a block the skip
        # contribution precedes the input gradient): if it does so through its input gradient, that kernel's epilogue also sums the
        # block's BatchNorm backward statistics
        last = [n] * n                                   # n = the prob layer (only block n-1 feeds it)
        for i in range(n):
            for j in (prog[i][2], prog[i][3]):
                if j >= 0:
                    last[j] = min(last[j], i)

        def bn_of(j, i):
            """BatchNorm operands for the input-gradient kernel of consumer i writing block j's gradient, if it completes it"""
            return (raws[j], statss[j], slots_b[j]) if (last[j] == i and (i == n or prog[i][2] == j)) else None

        grads = [None] * (5 * n + 2)
        g = [None] * n                                   # gradient w.r.t. block outputs, summed over their consumers
        have = [False] * n                               # backward statistics of block j already in slots_b[j]
        gx = None
        # ---- prob layer ----
        gy = as_cl3(glogits)
        bn = bn_of(n - 1, n) if last[n - 1] == n else None
        g[n - 1] = conv3d_dgrad(gy, wp, tuple(ys[-1].shape), 1, False, bn=bn, packed_ws=packed[dg_index[n]])
        have[n - 1] = bn is not None
        grads[5 * n] = wgrad(ys[-1], gy, wp, 1, False, need[2 + 5 * n])
        if need[2 + 5 * n + 1]:
            grads[5 * n + 1] = gy.sum().reshape(1) if gy.shape[1] == 1 else gy.sum(dim=(0, 2, 3, 4))
        # ---- blocks, last to first: every consumer of a block's output has run when the block is reached ----
        for i in range(n - 1, -1, -1):
            transposed, stride, src, skip, eps, momentum = prog[i]
            gy = g[i]
            g[i] = None
            raw, w = raws[i], ws_[i]
            draw, grads[5 * i + 1], grads[5 * i + 2] = bn_relu_bwd_slots(gy, raw, statss[i], slots_b[i], have[i])
            if skip >= 0:                                # y = relu(bn(raw)) + y_skip: the skip source receives gy as it is
                g[skip] = gy if g[skip] is None else g[skip] + gy
            xin = x if src < 0 else ys[src]
            # the side stream forks where the weight gradient is ENQUEUED: after the block's input gradient (it then starts when
            # that kernel has finished) or, knob _WGRAD_FORK_EARLY, before it (it starts as soon as `draw` exists)
            early = _WGRAD_FORK_EARLY == 2 or (_WGRAD_FORK_EARLY == 1 and src >= 0)
            if early:
                grads[5 * i] = wgrad(xin, draw, w, stride, transposed, need[2 + 5 * i])
            if src >= 0:
                bn = bn_of(src, i)
                g[src] = conv3d_dgrad(draw, w, tuple(xin.shape), stride, transposed, add=g[src], bn=bn, packed_ws=packed[dg_index[i]])
                have[src] = bn is not None
            elif need[0]:
                gx = conv3d_dgrad(draw, w, tuple(xin.shape), stride, transposed, add=gx, packed_ws=packed[dg_index[i]])
            if not early:
                grads[5 * i] = wgrad(xin, draw, w, stride, transposed, need[2 + 5 * i])
        if side_used[0]:
            deferred = ctx.tail_join or (_DEFER_JOIN and all(gw is None or _async_safe(params_w) for gw, params_w in zip(grads[0:5 * n:5] + [grads[5 * n]], list(ws_) + [wp])))
            if deferred:
                # ONE join at the end of the whole backward pass (autograd engine callback): see set_async_wgrad
                idx = dev.index
                ent = _BWD_OPEN.get(idx)
                if ent is None:
                    ent = _BWD_OPEN[idx] = [main, False]
                    torch.autograd.Variable._execution_engine.queue_callback(lambda: _end_of_backward(idx))
                ent[1] = True
            else:
                _join_side(main, dev.index)                  # every weight gradient is complete before autograd sees it
        return (gx, None) + tuple(grads)


def unet_regulariser(x, blocks, prob, tail=None):
    """blocks: list of (module-with-.conv/.bn or Sequential(deconv, bn), transposed, stride, src, skip); prob: the bias-only conv.
    Train-mode fp32 forward of a whole regulariser through UNetRegulariserFn (the modules are parameter containers).
    tail: {id(weight): view} of tail_join_views() -- the convolution weights as outputs of the tail node (the node then leaves the join
    of its side-stream weight gradients to it)."""
    from . import nn3d
    tail = tail or {}
    prog, params = [], []
    for conv, bn, transposed, stride, src, skip in blocks:
        momentum = nn3d._bn_step(bn, True)
        prog.append((bool(transposed), int(stride), int(src), int(skip), float(bn.eps), float(momentum)))
        params += [tail.get(id(conv.weight), conv.weight), bn.weight, bn.bias, bn.running_mean, bn.running_var]
    params += [tail.get(id(prob.weight), prob.weight), prob.bias]
    return UNetRegulariserFn.apply(x, tuple(prog), *params)


class Conv2dSplitBwdFn(torch.autograd.Function):
    """A bias-free nn.Conv2d of the feature extractor through the LIBRARY's convolution (MIOpen) whose backward is issued as two
    calls -- input gradient on the main stream, weight gradient on the side stream (when _wgrad side-stream rules allow) -- instead
    of ATen's single node that runs both one after the other.  Same kernels, same results; only the schedule differs."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, hip_forward=False, want_stats=False, groups=1, side_stream=True, packed_ws=None,
                hip_dgrad=False, hip_wgrad=False):
        """hip_forward: the forward pass through csrc/conv2d.hip (3x3 s1 p1 / 5x5 s2 p2 on channels-last input), the backward stays
        the library's two calls.  want_stats (with hip_forward): -> (y, BatchNorm statistic slots of y for `groups` equal batch
        chunks), the slots not differentiable."""
        ctx.side_stream = bool(side_stream)   # False: the weight gradient stays on the main stream whatever set_async_wgrad says
        # the use is counted only where backward() will settle it (_maybe_on_side_stream -> _weight_use_done): with the weight
        # gradient pinned to the main stream or taken by csrc/conv2d.hip the count was never decremented and grew every step, and a
        # data_ptr left in _WEIGHT_MULTI pinned whatever weight the allocator put there next to the synchronous path
        ctx.counted = bool(ctx.needs_input_grad[1] and ctx.side_stream and not hip_wgrad and _ASYNC_WGRAD and weight.is_cuda)
        if ctx.counted:
            _note_weight_use(weight)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (list(stride), list(padding))
        # per-layer choice of the backward kernels (measured per layer: profiles/r04_run9_conv2d_layers.log)
        ctx.hip_dgrad, ctx.hip_wgrad = bool(hip_dgrad), bool(hip_wgrad)
        # (the statistic slots are a second, non-differentiable output: without this autograd would hand backward() a zero-filled
        #  float64 tensor of their shape -- one fill launch per layer and step, seen in profiles/r04_run3_trace_tail.csv)
        ctx.set_materialize_grads(False)
        if hip_forward and want_stats:
            y, slots = conv2d_forward(x, weight, None, stride[0], want_stats=True, groups=groups, packed_ws=packed_ws)
            ctx.mark_non_differentiable(slots)
            return y, slots
        if hip_forward:
            return conv2d_forward(x, weight, None, stride[0])
        return torch.ops.aten.convolution(x, weight, None, list(stride), list(padding), [1, 1], False, [0, 0], 1)

    @staticmethod
    def backward(ctx, gy, *_unused_grad_of_the_slots):
        if gy is None:
            if ctx.counted:
                x, weight = ctx.saved_tensors
                _weight_use_done(weight.device.index, weight.data_ptr())
            return (None,) * 11
        x, weight = ctx.saved_tensors
        stride, padding = ctx.cfg
        bwd = torch.ops.aten.convolution_backward
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if ctx.hip_dgrad:
                gx = conv2d_dgrad(gy, weight, tuple(x.shape), stride[0])
            else:
                gx = bwd(gy, x, weight, None, stride, padding, [1, 1], False, [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            if ctx.hip_wgrad:
                gw = conv2d_wgrad(x, gy, tuple(weight.shape), stride[0])
                if gw.stride() != weight.stride():       # the parameter's layout (channels-last extractor weights)
                    gw = torch.empty_strided(weight.shape, weight.stride(), dtype=gw.dtype, device=gw.device).copy_(gw)
            else:
                fn = lambda: bwd(gy, x, weight, None, stride, padding, [1, 1], False, [0, 0], 1, [False, True, False])[1]
                gw = _maybe_on_side_stream(fn, weight, (x, gy)) if ctx.side_stream else fn()
                if ctx.counted and not (_ASYNC_WGRAD and x.is_cuda):     # the flag was switched off between forward and backward
                    _weight_use_done(weight.device.index, weight.data_ptr())
        return gx, gw, None, None, None, None, None, None, None, None, None


def _maybe_on_side_stream(fn, weight, inputs):
    """Run `fn` (which produces a weight gradient from `inputs`) on the side stream when the per-layer rules of _wgrad_maybe_async
    allow it (opt-in flag, leaf weight without hooks / existing .grad, single forward use), else inline."""
    ref = inputs[0]
    if not (_ASYNC_WGRAD and ref.is_cuda):
        return fn()
    idx = ref.device.index
    main = torch.cuda.current_stream(ref.device)
    ent = _BWD_OPEN.get(idx)
    if ent is None:
        ent = _BWD_OPEN[idx] = [main, False]
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _end_of_backward(idx))
    # (any dense layout: the library convolution takes channels-last weights as they are)
    ok = (weight.is_leaf and weight.grad is None and not getattr(weight, "_backward_hooks", None)
          and not getattr(weight, "_post_accumulate_grad_hooks", None) and weight.data_ptr() not in _WEIGHT_MULTI.get(idx, ()))
    _weight_use_done(idx, weight.data_ptr())
    if not ok:
        return fn()
    side = _side_stream(ref.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = fn()
        if out.stride() != weight.stride():
            # Root cause of round 3's "unexplained" mismatch (60 % relative L1 on feature.conv0.conv.weight): for channels-last
            # activations the library returns the weight gradient in channels-last strides although the parameter is contiguous
            # (or vice versa).  AccumulateGrad only takes a gradient over as .grad when its layout matches the parameter's;
            # otherwise it CLONES it -- a kernel on the MAIN stream, before the end-of-backward join, i.e. a read of a gradient
            # the side stream has not finished.  The layout is fixed here, on the side stream, so AccumulateGrad
            # never launches anything on it.
            out = torch.empty_strided(weight.shape, weight.stride(), dtype=out.dtype, device=out.device).copy_(out)
    for ten in inputs:
        ten.record_stream(side)
    out.record_stream(main)
    ent[1] = True
    return out


# training extractor (FeatureExtractorFn): all layers' weight gradients through csrc/conv2d.hip's one-launch kernel (MVS_FEATURE_WGRAD_BATCH=0:
# the library's per-layer weight gradients)
FEATURE_WGRAD_BATCH = os.environ.get("MVS_FEATURE_WGRAD_BATCH", "1") != "0"
# consumer-side BatchNorm + ReLU in the training extractor -- 6 of its 7 apply passes and their outputs go away.  Built at the end of
# round 4, parity-tested on the emulated kernels and on the GPU (incl. the full-size config-2 / config-3 steps against the oracle);
# measured 5.2168 -> 5.1895 and 5.2086 -> 5.1804 ms per config-2 step, every one of 6 + 5 interleaved pairs (profiles/r04_run33_*,
# r04_run34_*).  ON by default since the round's very last commit -- AFTER the closing line of profiles/r04_final_* (5.238 ms), which
# was measured without it.  MVS_FEATURE_FUSED_APPLY=0 restores the apply passes.
FEATURE_FUSED_APPLY = os.environ.get("MVS_FEATURE_FUSED_APPLY", "1") != "0"
# opt-in: the BatchNorm backward statistics of the block BELOW a csrc/conv2d.hip input gradient in that kernel's epilogue (3 of the
# extractor's 7 reduce launches go); built with the item above; measured NEUTRAL (5.2086 -> 5.2103 ms, profiles/r04_run34_*: the
# epilogue's reads of the raw tensor cost what the three reduce launches did)
FEATURE_DGRAD_BNSTATS = os.environ.get("MVS_FEATURE_DGRAD_BNSTATS", "0") == "1"


FEATURE_WGRAD_EARLY = os.environ.get("MVS_FEATURE_WGRAD_EARLY", "1") != "0"
# Round 6: the training extractor ENTIRELY through csrc/conv2d.hip (VERDICT r5 missing #1: five of its eight input gradients and the
# closing convolution's forward were library calls -- igemm_bwd_gtcx35_* / SubTensorOpWithScalar1d / a CK grouped-conv kernel in the
# step's rocprofv3 table, plus MIOpen's naive_conv_* search at start-up).  MVS_FEATURE_ALL_OWN=0 restores the per-layer choice of round 4.
FEATURE_ALL_OWN = os.environ.get("MVS_FEATURE_ALL_OWN", "1") != "0"
FEATURE_BIAS_SIDE = os.environ.get("MVS_FEATURE_BIAS_SIDE", "1") != "0"   # with FEATURE_WGRAD_EARLY: the closing convolution's bias gradient on the side stream


# Round 6: the node's two passes as ONE C call each (csrc/feature_pass.cpp: mvs_feature_fwd / mvs_feature_bwd), in the node's default
# configuration (every convolution through csrc/conv2d.hip, consumer-side BatchNorm, one-launch weight gradients, the wide layers' forked
# to the side stream).  MVS_FEATURE_C_ENTRY=0 keeps the per-layer calls from Python (same kernels, same order).
FEATURE_C_ENTRY = os.environ.get("MVS_FEATURE_C_ENTRY", "1") != "0"
_FEATURE_PLANS = {}


def _w_layout(w):
    """0: contiguous [Cout][Cin][k][k]; 1: channels-last in memory; None: neither (the C entry does not serve it)"""
    if w.is_contiguous():
        return 0
    return 1 if w.is_contiguous(memory_format=CL2) else None


class _FeaturePlan:
    """Everything about (block table, input shape, weight layouts) that is the same every step: the C block table, tensor sizes, arena offsets."""

    def __init__(self, lib, cfg, xshape, groups, wshapes, wcls, fshape, fwcl):
        n = len(cfg)
        N, c0, h, w = xshape
        self.n, self.N, self.G = n, N, groups
        self.blocks = (_lib.MvsFeatBlock * n)()
        self.out = []                      # (N, cout, Ho, Wo) of every block
        al = lambda v: (v + 63) // 64 * 64
        cin = c0
        self.dgrad_floats = 0
        for i, ((stride, padding, eps, momentum, _), ws_) in enumerate(zip(cfg, wshapes)):
            cout, cin_w, ks, _ = ws_
            b = self.blocks[i]
            b.cin, b.cout, b.ks, b.stride, b.eps, b.momentum, b.w_channels_last, b.h, b.w = cin, cout, ks, stride, float(eps), float(momentum), wcls[i], h, w
            self.dgrad_floats = max(self.dgrad_floats, int(lib.raw("mvs_conv2d_workspace_floats", 1, N, h, w, cin, cout, ks, stride)))
            h, w = (h + 2 * padding - ks) // stride + 1, (w + 2 * padding - ks) // stride + 1
            self.out.append((N, cout, h, w))
            cin = cout
        self.close_cout, self.fwcl = fshape[0], fwcl
        self.dgrad_floats = max(self.dgrad_floats, int(lib.raw("mvs_conv2d_workspace_floats", 1, N, h, w, cin, fshape[0], 3, 1)))
        self.out_shape = (N, fshape[0], h, w)
        sizes = [o[0] * o[1] * o[2] * o[3] for o in self.out]
        # forward arena (floats): raw_i ..., y_last, stats_i ..., forward weight images, the closing convolution's weight image
        off = 0
        self.raw_off, self.stats_off, self.packed_off = [], [], []
        for i in range(n):
            self.raw_off.append(off); off += al(sizes[i])
        self.ylast_off = off; off += al(sizes[n - 1])
        for i in range(n):
            self.stats_off.append(off); off += al(groups * 4 * self.out[i][1])
        for i in range(n):
            b = self.blocks[i]
            self.packed_off.append(off); off += al(int(lib.raw("mvs_conv2d_workspace_floats", 0, 1, 8, 8, b.cin, b.cout, b.ks, b.stride)))
        self.wsclose_off = off; off += al(int(lib.raw("mvs_conv2d_workspace_floats", 0, 1, 8, 8, cin, fshape[0], 3, 1)))
        self.fwd_floats = off
        # statistic slots (doubles): forward and backward rows of every block
        self.nslots = (C.c_int * n)(*[bn_nslots(lib, o[1]) for o in self.out])
        soff = 0
        self.sf_off, self.sb_off = [], []
        for i in range(n):
            cnt = groups * self.nslots[i] * 2 * self.out[i][1]
            self.sf_off.append(soff); soff += cnt
            self.sb_off.append(soff); soff += cnt
        self.slot_doubles = soff
        # backward work arena: gbuf_i, draw_i, the input gradients' weight image
        off = 0
        self.g_off, self.draw_off = [], []
        for i in range(n):
            self.g_off.append(off); off += al(sizes[i])
            self.draw_off.append(off); off += al(sizes[i])
        self.dgws_off = off; off += al(self.dgrad_floats)
        self.bwd_floats = off
        self.dgb_off, off = [], 0
        for i in range(n):
            self.dgb_off.append(off); off += 2 * self.out[i][1]
        self.dgb_floats = off
        # weight-gradient batches: layer j's shape row (N, H, W, Cin, Cout, ks, stride, channels-last) as mvs_feature_bwd builds them
        rows = [[N, self.blocks[j].h, self.blocks[j].w, self.blocks[j].cin, self.blocks[j].cout, self.blocks[j].ks, self.blocks[j].stride, wcls[j]]
                for j in range(n)]
        rows.append([N, self.out[n - 1][2], self.out[n - 1][3], self.out[n - 1][1], fshape[0], 3, 1, fwcl])
        self.wrows = rows
        self.ok = n <= 7 and all(v is not None for v in wcls) and fwcl is not None and self.dgrad_floats > 0 and \
            int(lib.raw("mvs_conv2d_wgrad_batch_workspace_floats", n + 1, (C.c_int * (8 * (n + 1)))(*sum(rows, [])))) >= 0

    def wgrad_floats(self, lib, lo, hi):
        if hi <= lo:
            return 0
        flat = sum(self.wrows[lo:hi], [])
        return int(lib.raw("mvs_conv2d_wgrad_batch_workspace_floats", hi - lo, (C.c_int * len(flat))(*flat)))


def _feature_plan(lib, cfg, xshape, groups, ws_, fw):
    wcls = tuple(_w_layout(w) for w in ws_)
    key = (cfg, tuple(xshape), groups, tuple(tuple(w.shape) for w in ws_), wcls, tuple(fw.shape), _w_layout(fw))
    plan = _FEATURE_PLANS.get(key)
    if plan is None:
        if len(_FEATURE_PLANS) > 16:
            _FEATURE_PLANS.clear()
        plan = _FEATURE_PLANS[key] = _FeaturePlan(lib, cfg, tuple(xshape), groups, key[3], wcls, key[5], key[6])
    return plan


def _feature_early_from(ws_, n, on_gpu):
    """first block with >= 32 output channels: its and the later layers' weight gradients go to the side stream (FEATURE_WGRAD_EARLY)"""
    if not (FEATURE_WGRAD_EARLY and _ASYNC_WGRAD_FUSED and on_gpu and n >= 4):
        return None
    e = next((i for i in range(n) if ws_[i].shape[0] >= 32), None)
    return e if (e is not None and 0 < e < n) else None


class FeatureExtractorFn(torch.autograd.Function):
    """A chain of 2-D ConvBnReLU blocks closed by a plain convolution with bias -- FeatureNet (jdacs/models/mvsnet.py:17-34) -- in
    TRAINING as ONE autograd node: the same kernels in the same order as the per-block graph (conv2d.hip forward with BatchNorm's
    statistics in its epilogue, apply pass; backward: reduce + apply, csrc/conv2d.hip or the library's input gradient per
    ``cfg``, the library's weight gradient), without 14 Function.apply round trips forward and 15 autograd nodes backward on the
    launch thread (the host enqueues a config-2 step in ~4 ms against ~5.4 ms of GPU time: the headroom is what keeps the step
    GPU-bound on a slower host).

    ``cfg``: per block (stride, padding, eps, momentum, hip_dgrad).  ``params``: per block conv weight, gamma, beta, running_mean,
    running_var; then the closing convolution's weight and bias.  ``groups`` equal chunks of the batch keep their own BatchNorm
    statistics (the views of a sample: mvsnet.py:115 calls the extractor once per view)."""

    @staticmethod
    def forward(ctx, x, groups, cfg, *params):
        lib = _lib_for(x)
        n = len(cfg)
        x = as_cl2(x)
        ws_, gammas, betas = [params[5 * i] for i in range(n)], [params[5 * i + 1] for i in range(n)], [params[5 * i + 2] for i in range(n)]
        fw, fb = params[5 * n], params[5 * n + 1]
        packed = pack_conv2d_weights(ws_, [c[0] for c in cfg], x)
        # FEATURE_FUSED_APPLY: block i's BatchNorm + ReLU is applied by block i+1's convolution (forward AND weight gradient) while it
        # stages its input -- no apply pass and no normalised copy for blocks 0 .. n-2; the last block's output is materialised for the
        # closing (library) convolution.  Needs the one-launch weight gradients in the backward pass.
        # (the normalising weight-gradient kernel keeps every group's scale / shift in 512 floats of LDS: groups x channels <= 256)
        fused = bool(FEATURE_FUSED_APPLY and FEATURE_WGRAD_BATCH and n >= 2 and all(c[1] == w.shape[2] // 2 for c, w in zip(cfg, ws_))
                     and groups * max(w.shape[1] for w in ws_) <= 256 and all(need for need in ctx.needs_input_grad[3::5][:n + 1]))
        if fused:
            # the backward pass of the fused form HAS to take the one-launch weight gradient (the normalised activations are not
            # kept): ask now, from the shapes alone, whether every layer has an instantiation (weight layout, < 2^31 elements, <= 8
            # layers) -- a "no" found in backward() would be an error with the forward already done
            shp, (bn_, _, hh, ww_) = [], x.shape
            for (stride, padding, *_), w in zip(cfg, ws_):
                shp.append(torch.empty((bn_, w.shape[1], hh, ww_), device="meta"))
                hh, ww_ = (hh + 2 * padding - w.shape[2]) // stride + 1, (ww_ + 2 * padding - w.shape[3]) // stride + 1
            shp.append(torch.empty((bn_, fw.shape[1], hh, ww_), device="meta"))
            fused = _wgrad_batch_serves_shapes(lib, shp, list(ws_) + [fw], [c[0] for c in cfg] + [1])
        own = bool(FEATURE_ALL_OWN)          # this node only runs where csrc/conv2d.hip serves every block (FeatureNet.forward checks)
        ctx.c_entry = False
        if (FEATURE_C_ENTRY and fused and own and lib.profiler is None and _N_SIDE == 1 and not FEATURE_DGRAD_BNSTATS and fb is not None
                and tuple(fw.shape[2:]) == (3, 3) and all(c[1] == w.shape[2] // 2 for c, w in zip(cfg, ws_))):
            plan = _feature_plan(lib, cfg, x.shape, groups, ws_, fw)
            if plan.ok:
                dev = x.device
                arena = torch.empty(plan.fwd_floats, dtype=torch.float32, device=dev)
                (slots,) = stat_slots(x, 1, 1, plan.slot_doubles // 2 + 1, 1)      # one zero-filled run of doubles for every block's rows
                out = torch.empty(plan.out_shape, dtype=torch.float32, device=dev, memory_format=CL2)
                ab, sb = arena.data_ptr(), slots.data_ptr()
                fbc = fb.contiguous()                 # (a local: the pointer must stay valid until the call has been made)
                lib.call("mvs_feature_fwd", n, plan.blocks, plan.N, groups, _p(x), _ptr_array(ws_), _ptr_array(gammas), _ptr_array(betas),
                         _ptr_array([params[5 * i + 3] for i in range(n)]), _ptr_array([params[5 * i + 4] for i in range(n)]),
                         _ptrs(ab, plan.packed_off), _ptrs(ab, plan.raw_off), ab + 4 * plan.ylast_off, _ptrs(ab, plan.stats_off),
                         _ptrs(sb, plan.sf_off, 8), plan.nslots, _p(fw), _p(fbc), plan.close_cout, plan.fwcl,
                         ab + 4 * plan.wsclose_off, _p(out), _stream(x))
                ctx.cfg, ctx.groups, ctx.slots_used, ctx.fused, ctx.own, ctx.c_entry, ctx.plan = cfg, groups, False, True, True, True, plan
                ctx.save_for_backward(x, fw, *ws_, arena, slots)
                return out
        acts, raws, statss, slots_b = [x], [], [], []
        for i, (stride, padding, eps, momentum, hip_dgrad) in enumerate(cfg):
            if fused and i > 0:
                raw, slots = conv2d_forward(raws[-1], ws_[i], None, stride, want_stats=True, groups=groups, packed_ws=packed[i], in_stats=statss[-1])
            else:
                raw, slots = conv2d_forward(acts[-1], ws_[i], None, stride, want_stats=True, groups=groups, packed_ws=packed[i])
            c = raw.shape[1]
            (sb,) = stat_slots(x, groups, bn_nslots(lib, c), c, 1)
            if fused and i < n - 1:
                stats = bn_finalize_slots(slots, gammas[i], betas[i], params[5 * i + 3], params[5 * i + 4], eps, momentum,
                                          raw.numel() // c // groups)
            else:
                y, stats = bn_relu_fwd_slots(raw, slots, gammas[i], betas[i], params[5 * i + 3], params[5 * i + 4], eps, momentum, None, True, groups)
                acts.append(y)
            raws.append(raw)
            statss.append(stats)
            slots_b.append(sb)
        if own:
            out = conv2d_forward(acts[-1], fw, fb, 1)
        else:
            out = torch.ops.aten.convolution(acts[-1], fw, fb, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
        ctx.cfg, ctx.groups, ctx.slots_used, ctx.fused, ctx.own = cfg, groups, False, fused, own
        if fused:       # acts = [x, y_last]
            ctx.save_for_backward(fw, *ws_, acts[0], acts[-1], *raws, *statss, *slots_b)
        else:
            ctx.save_for_backward(fw, *ws_, *acts, *raws, *statss, *slots_b)
        return out

    @staticmethod
    def _backward_c(ctx, gout):
        """the whole backward pass as ONE C call (mvs_feature_bwd): same kernels, same order, same fork point as backward() below"""
        plan, groups = ctx.plan, ctx.groups
        n = plan.n
        sv = ctx.saved_tensors
        x, fw, ws_, arena, slots = sv[0], sv[1], sv[2:2 + n], sv[2 + n], sv[3 + n]
        lib = _lib_for(x)
        dev = x.device
        need = ctx.needs_input_grad              # [x, groups, cfg, *params]
        if ctx.slots_used:                       # a second backward through the same graph: fresh backward accumulators
            slots = torch.zeros_like(slots)
        ctx.slots_used = True
        gout = as_cl2(gout)
        main = torch.cuda.current_stream(dev) if x.is_cuda else None
        early = _feature_early_from(ws_, n, x.is_cuda)
        side = _side_stream(dev) if early is not None else None
        grads = [None] * (5 * n + 2)
        if need[3 + 5 * n + 1]:
            if side is not None and FEATURE_BIAS_SIDE:
                gb = torch.empty(gout.shape[1], dtype=gout.dtype, device=dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    torch.sum(gout, (0, 2, 3), out=gb)
                grads[5 * n + 1] = gb
            else:
                grads[5 * n + 1] = gout.sum((0, 2, 3))
        work = torch.empty(plan.bwd_floats, dtype=torch.float32, device=dev)
        dgb = torch.empty(plan.dgb_floats, dtype=torch.float32, device=dev)
        gws = [torch.empty_like(w) for w in list(ws_) + [fw]]       # preserve_format: the parameter's strides
        lo = early if early is not None else n + 1
        ws_main = torch.empty(max(1, plan.wgrad_floats(lib, 0, lo)), dtype=torch.float32, device=dev)
        ws_side = torch.empty(max(1, plan.wgrad_floats(lib, lo, n + 1)), dtype=torch.float32, device=dev) if early is not None else None
        gx = torch.empty(tuple(x.shape), dtype=torch.float32, device=dev, memory_format=CL2) if need[0] else None
        ab, sb, wb = arena.data_ptr(), slots.data_ptr(), work.data_ptr()
        used = C.c_int(0)
        lib.call("mvs_feature_bwd", n, plan.blocks, plan.N, groups, _p(x), _ptr_array(ws_), _p(fw), plan.close_cout, plan.fwcl,
                 _ptrs(ab, plan.raw_off), ab + 4 * plan.ylast_off, _ptrs(ab, plan.stats_off), _ptrs(sb, plan.sb_off, 8), plan.nslots, _p(gout),
                 _ptrs(wb, plan.g_off), _ptrs(wb, plan.draw_off), _p(gx), wb + 4 * plan.dgws_off, _ptr_array(gws), _p(ws_main), _p(ws_side),
                 _ptrs(dgb.data_ptr(), plan.dgb_off), _ptrs(dgb.data_ptr(), [o + plan.out[i][1] for i, o in enumerate(plan.dgb_off)]),
                 early if early is not None else 0, main.cuda_stream if main is not None else None,
                 side.cuda_stream if side is not None else None, C.byref(used))
        # (no record_stream on the tensors the side stream touched: mvs_feature_bwd has made the main stream wait for the side stream
        #  before it returned, so everything enqueued on the main stream from here on -- which is where the caching allocator hands these
        #  blocks out again -- is ordered behind the side stream's work.  With record_stream the allocator could not reuse the 1 GB work
        #  arena of a batch-4 step until an event had been polled complete and fell back to hipMalloc / hipFree every step: 48 ms per
        #  step instead of 17, profiles/r06_run12_*.)
        for i in range(n):
            c = plan.out[i][1]
            grads[5 * i] = gws[i]
            grads[5 * i + 1] = dgb[plan.dgb_off[i]:plan.dgb_off[i] + c]
            grads[5 * i + 2] = dgb[plan.dgb_off[i] + c:plan.dgb_off[i] + 2 * c]
        grads[5 * n] = gws[n]
        return (gx, None, None) + tuple(grads)

    @staticmethod
    def backward(ctx, gout):
        if ctx.c_entry:
            return FeatureExtractorFn._backward_c(ctx, gout)
        cfg, groups = ctx.cfg, ctx.groups
        n = len(cfg)
        sv = ctx.saved_tensors
        fw, ws_ = sv[0], sv[1:1 + n]
        if ctx.fused:
            # the input of block i > 0 is relu(bn(raws[i-1])): never materialised; the library's input gradient only looks at its shape
            x0, y_last = sv[1 + n], sv[2 + n]
            raws, statss, slots_b = sv[3 + n:3 + 2 * n], sv[3 + 2 * n:3 + 3 * n], sv[3 + 3 * n:3 + 4 * n]
            acts = [x0] + list(raws[:n - 1]) + [y_last]
            x_stats = [None] + list(statss[:n - 1]) + [None]
        else:
            acts = sv[1 + n:2 + 2 * n]
            raws, statss, slots_b = sv[2 + 2 * n:2 + 3 * n], sv[2 + 3 * n:2 + 4 * n], sv[2 + 4 * n:2 + 5 * n]
            x_stats = None
        if ctx.slots_used:
            slots_b = [torch.zeros_like(t) for t in slots_b]
        ctx.slots_used = True
        need = ctx.needs_input_grad              # [x, groups, cfg, *params]
        bwd = torch.ops.aten.convolution_backward
        grads = [None] * (5 * n + 2)
        gout = as_cl2(gout)
        # weight gradients: all eight layers in ONE launch at the end (csrc/conv2d.hip conv2d_wgrad_batch_kernel; every layer's
        # input and output gradient is alive until then), unless a layer has no instantiation / FEATURE_WGRAD_BATCH is off ->
        # the library's, layer by layer
        batch = ((FEATURE_WGRAD_BATCH or ctx.fused) and all(need[3 + 5 * i] for i in range(n + 1)) and all(c[1] == w.shape[2] // 2 for c, w in zip(cfg, ws_))
                 and conv2d_wgrad_batch_serves(list(acts), list(ws_) + [fw], [c[0] for c in cfg] + [1]))
        if ctx.fused and not batch:
            raise RuntimeError("mvs_amd: FEATURE_FUSED_APPLY needs every convolution weight to require a gradient and a one-launch "
                               "weight-gradient instantiation for every layer (the normalised activations were not kept)")
        draws = [None] * n
        # FEATURE_WGRAD_EARLY: the weight gradients of the LAST layers (the 32-channel ones: two of the batch kernel's three launches'
        # worth of MFMA work) are enqueued on the side stream as soon as their output gradients exist and run next to the rest of
        # this backward pass -- the main stream is the step's critical path (bench.py --step-events: the side stream ends 0.1 ms
        # before it), so what leaves it shortens the step.  Joined before this node returns.
        early_from = _feature_early_from(ws_, n, gout.is_cuda) if batch else None
        if batch:
            if need[3 + 5 * n + 1]:
                if early_from is not None and FEATURE_BIAS_SIDE:
                    # the closing convolution's bias gradient (two reduction launches over the output gradient) is nobody's input
                    # before the optimiser: on the side stream too, into a buffer of the main stream's allocator pool
                    main = torch.cuda.current_stream(gout.device)
                    side = _side_stream(gout.device)
                    gb = torch.empty(gout.shape[1], dtype=gout.dtype, device=gout.device)
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        torch.sum(gout, (0, 2, 3), out=gb)
                    grads[5 * n + 1] = gb
                else:
                    grads[5 * n + 1] = gout.sum((0, 2, 3))
            if ctx.own:
                g = conv2d_dgrad(gout, fw, tuple(acts[n].shape), 1)
            else:
                g = bwd(gout, acts[n], fw, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        else:
            g, gfw, gfb = bwd(gout, acts[n], fw, [fw.shape[0]], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                              [True, bool(need[3 + 5 * n]), bool(need[3 + 5 * n + 1])])
            grads[5 * n], grads[5 * n + 1] = gfw, gfb
        have = False     # the block's backward statistics are already in its slots (the input gradient above it put them there)
        early_gws = None
        for i in range(n - 1, -1, -1):
            stride, padding, eps, momentum, hip_dgrad = cfg[i]
            draw, grads[5 * i + 1], grads[5 * i + 2] = bn_relu_bwd_slots(g, raws[i], statss[i], slots_b[i], have, True, groups)
            have = False
            draws[i] = draw
            if early_from is not None and i == early_from:
                main = torch.cuda.current_stream(gout.device)
                side = _side_stream(gout.device)
                side.wait_stream(main)
                lo = early_from
                early_gws = conv2d_wgrad_batch(list(acts[lo:]), draws[lo:] + [gout], list(ws_[lo:]) + [fw], [c[0] for c in cfg[lo:]] + [1],
                                               None if x_stats is None else x_stats[lo:], groups, on_stream=side)
            w = ws_[i]
            want_x = i > 0 or need[0]
            if want_x and (hip_dgrad or ctx.own):
                if FEATURE_DGRAD_BNSTATS and i > 0 and stride == 1 and w.shape[2] == 3:
                    # gx is the complete output gradient of block i-1: its BatchNorm backward statistics ride in this epilogue
                    g = conv2d_dgrad(draw, w, tuple(acts[i].shape), stride, bn=(raws[i - 1], statss[i - 1], slots_b[i - 1]), groups=groups)
                    have = True
                else:
                    g = conv2d_dgrad(draw, w, tuple(acts[i].shape), stride)
                want_x = False
            want_w = bool(need[3 + 5 * i]) and not batch
            if want_x or want_w:
                gx, gw, _ = bwd(draw, acts[i], w, None, [stride, stride], [padding, padding], [1, 1], False, [0, 0], 1,
                                [bool(want_x), want_w, False])
                if want_x:
                    g = gx
                grads[5 * i] = gw
        if batch and early_gws is not None:
            lo = early_from
            gws = conv2d_wgrad_batch(list(acts[:lo]), draws[:lo], list(ws_[:lo]), [c[0] for c in cfg[:lo]], None if x_stats is None else x_stats[:lo], groups)
            for i in range(lo):
                grads[5 * i] = gws[i]
            for k, i in enumerate(range(lo, n + 1)):
                grads[5 * i] = early_gws[k]
            _join_side(torch.cuda.current_stream(gout.device), gout.device.index)     # complete before autograd sees them
        elif batch:
            gws = conv2d_wgrad_batch(list(acts), draws + [gout], list(ws_) + [fw], [c[0] for c in cfg] + [1], x_stats, groups)
            for i in range(n + 1):
                grads[5 * i] = gws[i]
        return (g if need[0] else None, None, None) + tuple(grads)


class BnReLUFn(torch.autograd.Function):
    """BatchNorm (+ReLU) over the channel dim of a channels-last tensor [B,C,H,W] / [B,C,D,H,W] with the same
    HIP kernels as the 3-D regulariser (statistic slots -> ONE apply pass that finishes them in its prologue; backward: one
    reduction pass + one apply pass).  Used by the 2-D ConvBnReLU blocks of the feature extractors
    (jdacs/models/module.py:15-22), where MIOpen's BatchNorm kernels take ~40 us per call at B=1 with 8-32 channels.

    ``groups`` > 1: the batch holds `groups` equal chunks (the N views of a sample pushed through the
    shared-weight extractor as one batch); statistics are taken per chunk and the running statistics are
    updated chunk after chunk, i.e. exactly what `groups` successive BatchNorm calls do."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, eps, momentum, groups, slots=None):
        """slots [groups, nslots, 2, C] fp64 (train mode): statistics of x already summed by the convolution that produced it
        (conv2d_forward(want_stats=True)) -- no statistics pass over x."""
        lib = _lib_for(x)
        st = _stream(x)
        c = x.shape[1]
        if x.shape[0] % groups:
            raise ValueError("batch %d not divisible into %d BatchNorm groups" % (x.shape[0], groups))
        fmt = CL2 if x.dim() == 4 else CL3
        x = x.contiguous(memory_format=fmt)
        vg = x.numel() // c // groups
        ctx.cfg = (training, c, fmt, groups)
        if not training:
            scale = torch.empty(c, dtype=torch.float32, device=x.device)
            shift = torch.empty_like(scale)
            lib.call("mvs_bn_eval_affine", _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), c, _p(scale), _p(shift), st)
            y = torch.empty_like(x, memory_format=fmt)
            lib.call("mvs_bn_relu_fwd", _p(x), _p(scale), _p(shift), None, 1, x.numel() // c, c, _p(y), st)
            return y
        nslots = bn_nslots(lib, c)
        if slots is not None:
            if slots.dtype != torch.float64 or tuple(slots.shape[0:1] + slots.shape[2:]) != (groups, 2, c) or not slots.is_contiguous():
                raise ValueError("BatchNorm statistic slots %s %s do not match %d groups of [nslots,2,%d] float64"
                                 % (slots.dtype, tuple(slots.shape), groups, c))
            (slots_b,) = stat_slots(x, groups, nslots, c, 1)
        else:
            slots, slots_b = stat_slots(x, groups, nslots, c, 2)
            lib.call("mvs_bn_stats_slots", _p(x), groups, vg, c, _p(slots), slots.shape[1], st)
        y, stats = bn_relu_fwd_slots(x, slots, gamma, beta, running_mean, running_var, eps, momentum, None, True, groups)
        ctx.save_for_backward(x, stats, slots_b)
        ctx.slots_used = False
        return y

    @staticmethod
    def backward(ctx, gy):
        training, c, fmt, groups = ctx.cfg
        if not training:
            raise NotImplementedError("mvs_amd: backward through eval-mode BatchNorm is not supported")
        x, stats, slots_b = ctx.saved_tensors
        gy = gy.contiguous(memory_format=fmt)
        if ctx.slots_used:
            slots_b = torch.zeros_like(slots_b)
        ctx.slots_used = True
        dx, dgamma, dbeta = bn_relu_bwd_slots(gy, x, stats, slots_b, False, True, groups)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


class ConvBias3dFn(torch.autograd.Function):
    """Conv3d k3 p1 s1 with bias, no BN / activation: the `prob` layer (jdacs/models/mvsnet.py:63,73;
    jdacs-ms/models/network.py:65,73)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = as_cl3(x)
        if ctx.needs_input_grad[1]:
            _note_weight_use(weight)
        y, _ = conv3d_forward(x, weight, 1, False, shift=bias.contiguous())
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = as_cl3(gy)
        gx = conv3d_dgrad(gy, weight, tuple(x.shape), 1, False) if ctx.needs_input_grad[0] else None
        gw = _wgrad_maybe_async(x, gy, weight, 1, False) if ctx.needs_input_grad[1] else None
        gb = None
        if ctx.needs_input_grad[2]:
            # Cout = 1 (the prob layer): a full reduction; ATen's reduce over dims (0, 2, 3, 4) of a [B,1,D,H,W] tensor takes 53 us
            # for 15.7 MB at config 2 (torch profile, run 16)
            gb = gy.sum().reshape(1) if gy.shape[1] == 1 else gy.sum(dim=(0, 2, 3, 4))
        return gx, gw, gb


# ------------------------------------------------------------------------------------------------
# soft-argmin + confidence (K9/K10)
# ------------------------------------------------------------------------------------------------
class SoftArgminConf(torch.autograd.Function):
    """logits [B,D,H,W], depth hypotheses [B,D] | [B,D,H,W] -> (depth [B,H,W], confidence [B,H,W]).
    Confidence is computed under no_grad in the reference (mvsnet.py:145) -> non differentiable."""

    @staticmethod
    def forward(ctx, logits, depth_values):
        lib = _lib_for(logits)
        logits = logits.contiguous()
        b, nd, h, w = logits.shape
        dv, per_pixel = _depth_arg(depth_values, b, h, w)
        if dv.shape[1] != nd:
            raise ValueError("depth hypotheses D=%d != logits D=%d" % (dv.shape[1], nd))
        depth, conf, smax, ssum = (torch.empty((b, h, w), dtype=torch.float32, device=logits.device)
                                   for _ in range(4))
        lib.call("mvs_softargmin_conf_fwd", _p(logits), _p(dv), per_pixel, b, nd, h, w, _p(depth), _p(conf),
                 _p(smax), _p(ssum), _stream(logits))
        ctx.save_for_backward(logits, dv, depth, smax, ssum)
        ctx.per_pixel = per_pixel
        ctx.mark_non_differentiable(conf)
        ctx.set_materialize_grads(False)     # no zero-filled gradient tensor for the confidence output
        return depth, conf

    @staticmethod
    def backward(ctx, gdepth, _gconf):
        if gdepth is None:
            return None, None
        logits, dv, depth, smax, ssum = ctx.saved_tensors
        lib = _lib_for(logits)
        b, nd, h, w = logits.shape
        gl = torch.empty_like(logits)
        gdepth = gdepth.contiguous()
        lib.call("mvs_softargmin_conf_bwd", _p(gdepth), _p(logits), _p(dv), ctx.per_pixel, _p(depth),
                 _p(smax), _p(ssum), b, nd, h, w, _p(gl), _stream(logits))
        return gl, None


def softargmin_conf(logits, depth_values):
    return SoftArgminConf.apply(logits, depth_values)


class MaskedSmoothL1(torch.autograd.Function):
    """mvsnet_loss (jdacs/models/mvsnet.py:164-166): mean smooth-L1 of est - gt over mask > 0.5, as one launch forward and one
    backward; differentiable w.r.t. the estimate."""

    @staticmethod
    def forward(ctx, est, gt, mask):
        lib = _lib_for(est)
        est, gt = est.contiguous(), gt.contiguous().to(torch.float32)
        mask = mask.contiguous().to(torch.float32)
        if est.shape != gt.shape or est.shape != mask.shape:
            raise ValueError("mvsnet_loss: shapes differ: %s %s %s" % (tuple(est.shape), tuple(gt.shape), tuple(mask.shape)))
        out = torch.empty(2, dtype=torch.float32, device=est.device)
        lib.call("mvs_masked_smooth_l1_fwd", _p(est), _p(gt), _p(mask), est.numel(), _p(out), _stream(est))
        ctx.save_for_backward(est, gt, mask, out)
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        est, gt, mask, out = ctx.saved_tensors
        lib = _lib_for(est)
        gest = torch.empty_like(est)
        g = gloss.contiguous().reshape(1).to(torch.float32)
        lib.call("mvs_masked_smooth_l1_bwd", _p(est), _p(gt), _p(mask), _p(out), _p(g), est.numel(), _p(gest), _stream(est))
        return gest, None, None


# ------------------------------------------------------------------------------------------------
# self-supervised loss on the path's output
# ------------------------------------------------------------------------------------------------
def unsup_view_transforms(cams: torch.Tensor):
    """cams [B,N,2,4,4] (extrinsic in [:, :, 0], K in [:, :, 1, :3, :3]) -> kinv [B,9], proj [B,N-1,12]: host torch code
    for the few 3x3 products of jdacs/losses/homography.py:188-236 (K_ref^-1; K_ref.[R_v R_ref^T | t_v - R_rel t_ref])."""
    R_ref, t_ref = cams[:, 0, 0, :3, :3], cams[:, 0, 0, :3, 3:4]
    K_ref = cams[:, 0, 1, :3, :3]
    R_v, t_v = cams[:, 1:, 0, :3, :3], cams[:, 1:, 0, :3, 3:4]
    R_rel = torch.matmul(R_v, R_ref.transpose(1, 2).unsqueeze(1))
    t_rel = t_v - torch.matmul(R_rel, t_ref.unsqueeze(1))
    proj = torch.matmul(K_ref.unsqueeze(1), torch.cat([R_rel, t_rel], 3))
    kinv = torch.linalg.inv_ex(K_ref).inverse
    b = cams.shape[0]
    return kinv.reshape(b, 9).contiguous(), proj.reshape(b, -1, 12).contiguous()


class UnsupLossFn(torch.autograd.Function):
    """depth [B,h,w] -> (total, reconstr, ssim, smooth) scalars; differentiable w.r.t. the depth map only (the images and
    cameras are inputs of the training step, jdacs/train.py:199-210)."""

    @staticmethod
    def forward(ctx, depth, ref_q, views_q, kinv, proj, smooth_lambda):
        lib = _lib_for(depth)
        depth = depth.contiguous()
        b, h, w = depth.shape
        nv = len(views_q)
        ref_q = ref_q.contiguous()
        views_q = [v.contiguous() for v in views_q]
        for t in [ref_q] + views_q:
            if tuple(t.shape) != (b, h, w, 3):
                raise ValueError("quarter-resolution images must be [B,h,w,3] = %s, got %s" % ((b, h, w, 3), tuple(t.shape)))
        if tuple(kinv.shape) != (b, 9) or tuple(proj.shape) != (b, nv, 12):
            raise ValueError("kinv must be [B,9] and proj [B,V,12], got %s / %s" % (tuple(kinv.shape), tuple(proj.shape)))
        nws = lib.raw("mvs_unsup_loss_workspace_floats", b, nv, h, w)
        if nws < 0:
            raise ValueError("unsup_loss: bad shape B=%d V=%d h=%d w=%d" % (b, nv, h, w))
        ws = torch.empty(nws, dtype=torch.float32, device=depth.device)
        out = torch.empty(4, dtype=torch.float32, device=depth.device)
        lib.call("mvs_unsup_loss_fwd", _p(ref_q), _ptr_array(views_q), _p(kinv), _p(proj), _p(depth), b, nv, h, w,
                 float(smooth_lambda), _p(ws), _p(out), _stream(depth))
        ctx.save_for_backward(depth, ref_q, kinv, proj, ws, *views_q)
        ctx.lam = float(smooth_lambda)
        total, reconstr, ssim, smooth = out[0], out[1], out[2], out[3]
        ctx.mark_non_differentiable(reconstr, ssim, smooth)   # reported for logging, like the reference's attributes
        return total, reconstr, ssim, smooth

    @staticmethod
    def backward(ctx, g_total, g_reconstr, g_ssim, g_smooth):
        depth, ref_q, kinv, proj, ws, *views_q = ctx.saved_tensors
        lib = _lib_for(depth)
        b, h, w = depth.shape
        # only the total is differentiable here (its three terms are reported for logging, like the reference's attributes)
        g = g_total.contiguous().reshape(1).to(torch.float32)
        gd = torch.empty_like(depth)
        lib.call("mvs_unsup_loss_bwd", _p(ref_q), _ptr_array(views_q), _p(kinv), _p(proj), _p(depth), b, len(views_q), h, w,
                 ctx.lam, _p(ws), _p(g), _p(gd), _stream(depth))
        return gd, None, None, None, None, None


def unsup_loss(depth, ref_q, views_q, kinv, proj, smooth_lambda=1.0):
    return UnsupLossFn.apply(depth, ref_q, list(views_q), kinv, proj, smooth_lambda)


# ------------------------------------------------------------------------------------------------
# per-level depth hypotheses of CVP-MVSNet
# ------------------------------------------------------------------------------------------------
def depth_hypotheses(ref_depths: torch.Tensor, mats: torch.Tensor) -> torch.Tensor:
    """ref_depths [B,H,W] fp32, mats [B,30] fp64 (K_ref^-1 | K_src (E_src E_ref^-1)[:3,:] | (K_ref R_ref)(K_src R_src)^-1)
    -> hypotheses [B,8,H,W] fp32 (jdacs-ms/models/modules.py:107-206).  No gradient, like the reference (no_grad)."""
    lib = _lib_for(ref_depths)
    ref_depths = ref_depths.detach().contiguous()
    b, h, w = ref_depths.shape
    if mats.dtype != torch.float64 or tuple(mats.shape) != (b, 30):
        raise ValueError("mats must be float64 [B,30], got %s %s" % (mats.dtype, tuple(mats.shape)))
    mats = mats.contiguous()
    ws = torch.empty(lib.raw("mvs_depth_hypo_workspace_doubles", b, h, w), dtype=torch.float64, device=ref_depths.device)
    hypos = torch.empty((b, 8, h, w), dtype=torch.float32, device=ref_depths.device)
    lib.call("mvs_depth_hypo", _p(ref_depths), _p(mats), b, h, w, _p(ws), _p(hypos), _stream(ref_depths))
    return hypos


# ------------------------------------------------------------------------------------------------
# SURVEY 8(f)-3, first cut (not used by default): the feature extractors' 2-D convolutions
# ------------------------------------------------------------------------------------------------
def _c2_ws(lib, op, n, h, w, cin, cout, ks, stride, like):
    nfl = lib.raw("mvs_conv2d_workspace_floats", op, n, h, w, cin, cout, ks, stride)
    if nfl < 0:
        raise ValueError("conv2d: unsupported shape (3x3 stride 1 or 5x5 stride 2, 1..32 or 64 channels): Cin=%d Cout=%d k=%d s=%d"
                         % (cin, cout, ks, stride))
    return torch.empty(nfl, dtype=torch.float32, device=like.device)


def pack_conv2d_weights(weights, strides, like):
    """Forward weight images of a list of Conv2d layers (weights [Cout,Cin,k,k], contiguous OR channels-last in memory) in ONE
    launch; returns the workspaces to hand to conv2d_forward(want_stats=True, packed_ws=...)."""
    lib = _lib_for(like)
    n = len(weights)
    sizes, shapes, wcl, ptrs = [], [], [], []
    for wt, st in zip(weights, strides):
        cout, cin, ks, _ = wt.shape
        nfl = lib.raw("mvs_conv2d_workspace_floats", 0, 1, 8, 8, cin, cout, ks, st)
        if nfl < 0:
            raise ValueError("conv2d: unsupported shape Cin=%d Cout=%d k=%d s=%d" % (cin, cout, ks, st))
        sizes.append((nfl + 3) // 4 * 4)
        shapes += [cin, cout, ks, st]
        if wt.is_contiguous():
            wcl.append(0)
        elif wt.is_contiguous(memory_format=CL2):
            wcl.append(1)
        else:
            wt = wt.contiguous()
            wcl.append(0)
        ptrs.append(wt)
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=like.device)
    views, off = [], 0
    for sz in sizes:
        views.append(buf[off:off + sz])
        off += sz
    lib.call("mvs_conv2d_pack_weights_batch", n, _ptr_array(ptrs), _ptr_array(views), (C.c_int * (4 * n))(*shapes),
             (C.c_int * n)(*wcl), _stream(like))
    return views


def _wgrad_batch_shapes(xs, weights, strides):
    shapes = []
    for x, wt, st in zip(xs, weights, strides):
        n, cin, h, w = x.shape
        cout, cin_w, ks, ks2 = wt.shape
        if cin_w != cin or ks != ks2:
            raise ValueError("weight shape %s does not match %d input channels" % (tuple(wt.shape), cin))
        if wt.is_contiguous():
            wcl = 0
        elif wt.is_contiguous(memory_format=CL2):
            wcl = 1
        else:
            return None
        shapes += [n, h, w, cin, cout, ks, st, wcl]
    return shapes


_WGRAD_BATCH_PLANS = {}


def conv2d_wgrad_batch_serves(xs, weights, strides) -> bool:
    """csrc/conv2d.hip's one-launch weight gradient has an instantiation for every one of these layers (<= 8 of them)"""
    return _wgrad_batch_serves_shapes(_lib_for(xs[0]), xs, weights, strides)


def _wgrad_batch_serves_shapes(lib, xs, weights, strides) -> bool:
    """conv2d_wgrad_batch_serves from the SHAPES of xs (meta tensors will do)"""
    if not (0 < len(xs) <= 8):
        return False
    try:
        shapes = _wgrad_batch_shapes(xs, weights, strides)
    except ValueError:
        return False
    if shapes is None:
        return False
    return _wgrad_batch_plan(lib, tuple(shapes))[0] >= 0


def _wgrad_batch_plan(lib, key):
    """(workspace floats, the shapes as a C array).  Only the array is cached: the size depends on the library's "wgrad2d_batch" knob
    and is asked for on every call (a stale, smaller size after a knob change would let the kernel write past the workspace)."""
    arr = _WGRAD_BATCH_PLANS.get(key)
    if arr is None:
        arr = _WGRAD_BATCH_PLANS[key] = (C.c_int * len(key))(*key)
    return int(lib.raw("mvs_conv2d_wgrad_batch_workspace_floats", len(key) // 8, arr)), arr


def conv2d_wgrad_batch(xs, gys, weights, strides, x_stats=None, groups=1, on_stream=None):
    """Weight gradients of several Conv2d layers (x_i channels-last [N,Cin,H,W], gy_i channels-last [N,Cout,Ho,Wo], pad k//2) in
    ONE launch + one reduction launch; -> gradients with the shape AND memory layout of `weights` (contiguous or channels-last
    parameters alike, so autograd's AccumulateGrad takes them over without a copy).  x_stats: per layer None or the [groups,4,Cin]
    statistics of the BatchNorm + ReLU block whose RAW output x_i is (the layer's input is normalised while it is staged).
    on_stream: a torch stream other than the current one to enqueue the two launches on (the caller has made it wait for the
    producers of xs / gys); outputs and workspace come from the current stream's pool and are handed over with record_stream."""
    lib = _lib_for(xs[0])
    xs, gys = [as_cl2(t) for t in xs], [as_cl2(t) for t in gys]
    st_handle = _stream(xs[0]) if on_stream is None else on_stream.cuda_stream
    shapes = _wgrad_batch_shapes(xs, weights, strides)
    nfl, arr = _wgrad_batch_plan(lib, tuple(shapes)) if shapes is not None else (-1, None)
    if nfl < 0:
        raise ValueError("conv2d_wgrad_batch: a layer of %s is not served" % ([tuple(w.shape) for w in weights],))
    ws = torch.empty(nfl, dtype=torch.float32, device=xs[0].device)
    gws = [torch.empty_like(w) for w in weights]       # preserve_format: the parameter's strides
    if x_stats is not None and any(t is not None for t in x_stats):
        n_img = xs[0].shape[0]
        if n_img % groups or any(x.shape[0] != n_img for x in xs):
            raise ValueError("conv2d_wgrad_batch: %d images do not split into %d groups" % (n_img, groups))
        st = (C.c_void_p * len(xs))()
        for i, (t, x) in enumerate(zip(x_stats, xs)):
            if t is not None:
                if tuple(t.shape) != (groups, 4, x.shape[1]) or t.dtype != torch.float32 or not t.is_contiguous():
                    raise ValueError("conv2d_wgrad_batch: x_stats[%d] %s does not match %d groups of [4,%d]" % (i, tuple(t.shape), groups, x.shape[1]))
                st[i] = t.data_ptr()
        lib.call("mvs_conv2d_wgrad_batch_xf", len(xs), _ptr_array(xs), st, n_img // groups, _ptr_array(gys), _ptr_array(gws), _p(ws), arr,
                 st_handle, tstream=on_stream)
    else:
        lib.call("mvs_conv2d_wgrad_batch", len(xs), _ptr_array(xs), _ptr_array(gys), _ptr_array(gws), _p(ws), arr, st_handle, tstream=on_stream)
    if on_stream is not None:
        for ten in list(xs) + list(gys) + gws + [ws] + [t for t in (x_stats or ()) if t is not None]:
            ten.record_stream(on_stream)
    return gws


def conv2d_forward(x, weight, bias=None, stride=1, negative_slope=None, want_stats=False, groups=1, packed_ws=None, in_stats=None):
    """x [N,Cin,H,W] (channels_last), weight [Cout,Cin,k,k], pad k//2 -> y [N,Cout,Ho,Wo] (channels_last);
    negative_slope: LeakyReLU fused after the bias.  want_stats (no bias / activation): -> (y, slots [groups,nslots,2,Cout] fp64),
    the BatchNorm statistics of y summed by the convolution's epilogue, the N images being `groups` equal chunks (BnReLUFn's
    ``slots``).  in_stats [groups,4,Cin] (with want_stats): x is the RAW output of the BatchNorm + ReLU block in front and
    relu(x * scale + shift) is applied while the kernel stages it (bn_finalize_slots made the statistics)."""
    lib = _lib_for(x)
    x = as_cl2(x)
    n, cin, h, w = x.shape
    cout, cin_w, ks, ks2 = weight.shape
    if cin_w != cin or ks != ks2:
        raise ValueError("weight shape %s does not match %d input channels" % (tuple(weight.shape), cin))
    ho, wo = (h, w) if stride == 1 else ((h - 1) // 2 + 1, (w - 1) // 2 + 1)
    ws = _c2_ws(lib, 0, n, h, w, cin, cout, ks, stride, x) if packed_ws is None else packed_ws
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=CL2)
    if (packed_ws is not None or in_stats is not None) and not want_stats:
        raise ValueError("conv2d_forward: packed_ws / in_stats serve the want_stats (training) form")
    if want_stats:
        wc = weight if packed_ws is not None else weight.contiguous()     # (a local: the pointer stays valid until the call is made)
        if bias is not None or negative_slope is not None:
            raise ValueError("conv2d_forward: statistics are those of the plain convolution (no bias / activation)")
        if n % groups:
            raise ValueError("conv2d_forward: %d images do not split into %d statistics groups" % (n, groups))
        (slots,) = stat_slots(x, groups, bn_nslots(lib, cout), cout, 1)
        if in_stats is not None:
            if tuple(in_stats.shape) != (groups, 4, cin) or in_stats.dtype != torch.float32 or not in_stats.is_contiguous():
                raise ValueError("conv2d_forward: in_stats %s does not match %d groups of [4,%d]" % (tuple(in_stats.shape), groups, cin))
            lib.call("mvs_conv2d_fwd_stats_xf", _p(x), _p(in_stats), _p(wc), _p(y), _p(ws),
                     _p(slots), slots.shape[1], groups, n, h, w, cin, cout, ks, stride, int(packed_ws is not None), _stream(x),
                     tag="fwd2d_stats_xf:%d>%d:k%d:s%d" % (cin, cout, ks, stride))
            return y, slots
        lib.call("mvs_conv2d_fwd_stats", _p(x), _p(wc), _p(y), _p(ws), _p(slots),
                 slots.shape[1], groups, n, h, w, cin, cout, ks, stride, int(packed_ws is not None), _stream(x),
                 tag="fwd2d_stats:%d>%d:k%d:s%d" % (cin, cout, ks, stride))
        return y, slots
    if negative_slope is not None:
        wc, bc = weight.contiguous(), (None if bias is None else bias.contiguous())   # (locals: the pointers stay valid until the call is made)
        lib.call("mvs_conv2d_lrelu_fwd", _p(x), _p(wc), _p(bc), _p(y), _p(ws),
                 n, h, w, cin, cout, ks, stride, float(negative_slope), _stream(x), tag="fwd2d_lrelu:%d>%d:k%d:s%d" % (cin, cout, ks, stride))
        return y
    wl = _w_layout(weight)
    if wl is None:
        weight, wl = weight.contiguous(), 0        # (kept in a local: the pointer must stay valid until the call has been made)
    bias_c = None if bias is None else bias.contiguous()
    lib.call("mvs_conv2d_fwd_wl", _p(x), _p(weight), _p(bias_c), _p(y), _p(ws), n, h, w, cin, cout, ks, stride, wl, _stream(x),
             tag="fwd2d:%d>%d:k%d:s%d" % (cin, cout, ks, stride))
    return y


def conv2d_dgrad(gy, weight, in_shape, stride=1, bn=None, groups=1):
    """bn = (raw, stats, slots) of the BatchNorm + ReLU block whose COMPLETE output gradient this input gradient is (3x3 stride 1):
    the epilogue adds that block's backward statistics into `slots` [groups,nslots,2,Cin] (bn_relu_bwd_slots(have_stats=True))."""
    lib = _lib_for(gy)
    gy = as_cl2(gy)
    n, cin, h, w = in_shape
    cout, _, ks, _ = weight.shape
    ws = _c2_ws(lib, 1, n, h, w, cin, cout, ks, stride, gy)
    gx = torch.empty((n, cin, h, w), dtype=torch.float32, device=gy.device, memory_format=CL2)
    if bn is not None:
        raw, stats, slots = bn
        if stride != 1 or ks != 3 or tuple(raw.shape) != (n, cin, h, w) or tuple(stats.shape) != (groups, 4, cin) or n % groups:
            raise ValueError("conv2d_dgrad(bn=...): a 3x3 stride-1 layer, raw %s like the input %s, stats [%d,4,%d]"
                             % (tuple(raw.shape), (n, cin, h, w), groups, cin))
        wc, rawc = weight.contiguous(), as_cl2(raw)
        lib.call("mvs_conv2d_dgrad_bnstats", _p(gy), _p(wc), _p(gx), _p(ws), n, h, w, cin, cout, ks, _p(rawc),
                 _p(stats), _p(slots), slots.shape[-3], groups, _stream(gy), tag="dgrad2d_bn:%d>%d:k%d" % (cin, cout, ks))
        return gx
    wl = _w_layout(weight)
    if wl is None:
        weight, wl = weight.contiguous(), 0
    lib.call("mvs_conv2d_dgrad_wl", _p(gy), _p(weight), _p(gx), _p(ws), n, h, w, cin, cout, ks, stride, wl, _stream(gy),
             tag="dgrad2d:%d>%d:k%d:s%d" % (cin, cout, ks, stride))
    return gx


def conv2d_wgrad(x, gy, weight_shape, stride=1):
    lib = _lib_for(x)
    x, gy = as_cl2(x), as_cl2(gy)
    n, cin, h, w = x.shape
    cout, _, ks, _ = weight_shape
    ws = _c2_ws(lib, 2, n, h, w, cin, cout, ks, stride, x)
    gw = torch.empty(tuple(weight_shape), dtype=torch.float32, device=x.device)
    lib.call("mvs_conv2d_wgrad", _p(x), _p(gy), _p(gw), _p(ws), n, h, w, cin, cout, ks, stride, _stream(x),
             tag="wgrad2d:%d>%d:k%d:s%d" % (cin, cout, ks, stride))
    return gw


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d (k3 s1 p1 | k5 s2 p2, optional bias) on channels-last activations through csrc/conv2d.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        x = as_cl2(x)
        y = conv2d_forward(x, weight, bias, stride)
        ctx.save_for_backward(x, weight)
        ctx.stride = stride
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = as_cl2(gy)
        gx = conv2d_dgrad(gy, weight, tuple(x.shape), ctx.stride) if ctx.needs_input_grad[0] else None
        gw = conv2d_wgrad(x, gy, tuple(weight.shape), ctx.stride) if ctx.needs_input_grad[1] else None
        gb = gy.sum(dim=(0, 2, 3)) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None


class Conv2dLReLUFn(torch.autograd.Function):
    """nn.Sequential(nn.Conv2d(k3 s1 p1, bias), nn.LeakyReLU(slope)) -- the `conv` block of the CVP feature pyramid
    (jdacs-ms/models/modules.py:15-19) -- forward in ONE csrc/conv2d.hip pass; backward: the activation's mask (one
    elementwise launch), then the input- and weight-gradient kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, negative_slope):
        x = as_cl2(x)
        y = conv2d_forward(x, weight, bias, 1, negative_slope=negative_slope)
        ctx.save_for_backward(x, weight, y)
        ctx.slope = float(negative_slope)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        # y > 0 <=> the pre-activation was > 0 (slope > 0), so the output is enough to mask the gradient
        g = torch.ops.aten.leaky_relu_backward(as_cl2(gy), y, ctx.slope, True).contiguous(memory_format=CL2)
        gx = conv2d_dgrad(g, weight, tuple(x.shape), 1) if ctx.needs_input_grad[0] else None
        gw = conv2d_wgrad(x, g, tuple(weight.shape), 1) if ctx.needs_input_grad[1] else None
        gb = g.sum(dim=(0, 2, 3)) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None
