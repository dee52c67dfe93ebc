"""ctypes binding of libmvs_hip.so (C ABI declared in include/mvs_hip.h).

The product has NO fallback: if the HIP library is missing or fails to load, ``get()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmvs_hip.so")

_f = C.c_void_p      # float* (device pointer)
_i = C.c_int
_ll = C.c_longlong
_s = C.c_void_p      # hipStream_t
_fl = C.c_float

class MvsUnetBlock(C.Structure):
    """include/mvs_hip.h: one Conv/Deconv + BatchNorm + ReLU (+ skip) block of a regulariser program"""
    _fields_ = [("transposed", _i), ("stride", _i), ("src", _i), ("skip", _i), ("eps", _fl), ("momentum", _fl),
                ("cin", _i), ("cout", _i), ("d", _i), ("h", _i), ("w", _i)]


class MvsFeatBlock(C.Structure):
    """include/mvs_hip.h: one ConvBnReLU block of the training 2-D extractor (mvs_feature_fwd / mvs_feature_bwd)"""
    _fields_ = [("cin", _i), ("cout", _i), ("ks", _i), ("stride", _i), ("eps", _fl), ("momentum", _fl), ("w_channels_last", _i),
                ("h", _i), ("w", _i)]


_pp = C.POINTER(C.c_void_p)   # array of device pointers

# name -> (restype, argtypes); every symbol include/mvs_hip.h declares
SIGNATURES = {
    "mvs_unet_time_wgrad": (_i, [_i]),
    "mvs_unet_time_read": (_i, [C.POINTER(_fl), _i]),
    "mvs_unet_fwd": (_i, [_i, C.POINTER(MvsUnetBlock), _i, _f, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _pp, C.POINTER(_i), _f, _f, _i,
                          _f, _f, _s]),
    "mvs_unet_bwd": (_i, [_i, C.POINTER(MvsUnetBlock), _i, _f, _pp, _f, _i, _pp, _pp, _pp, _pp, C.POINTER(_i), _pp, _f, _pp, _pp, _f, _pp, _pp,
                          _pp, _pp, _s, _s, _i, C.POINTER(_i)]),
    "mvs_feature_fwd": (_i, [_i, C.POINTER(MvsFeatBlock), _i, _i, _f, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _f, _pp, _pp, C.POINTER(_i), _f, _f, _i,
                             _i, _f, _f, _s]),
    "mvs_feature_bwd": (_i, [_i, C.POINTER(MvsFeatBlock), _i, _i, _f, _pp, _f, _i, _i, _pp, _f, _pp, _pp, C.POINTER(_i), _f, _pp, _pp, _f, _f,
                             _pp, _f, _f, _pp, _pp, _i, _s, _s, C.POINTER(_i)]),
    "mvs_conv2d_dgrad_wl": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_conv2d_fwd_wl": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_version": (_i, []),
    "mvs_last_error": (C.c_char_p, []),
    "mvs_is_emulation": (_i, []),
    "mvs_set_tuning": (_i, [C.c_char_p, _i]),
    "mvs_get_tuning": (_i, [C.c_char_p, C.POINTER(_i)]),
    "mvs_plane_sweep_variance_fwd": (_i, [_f, C.POINTER(C.c_void_p), _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _s]),
    "mvs_plane_sweep_variance_fwd_bf16": (_i, [_f, C.POINTER(C.c_void_p), _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _s]),
    "mvs_conv3d_bf16_workspace_bytes": (_ll, [_i] * 4),
    "mvs_conv3d_bf16_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _s]),
    "mvs_cast_f32_bf16": (_i, [_f, _f, _ll, _s]),
    "mvs_plane_sweep_variance_bwd": (_i, [_f, _f, C.POINTER(C.c_void_p), _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                          _f, C.POINTER(C.c_void_p), _s]),
    "mvs_relative_projection": (_i, [_f, _f, _i, _i, _f, _f, _s]),
    "mvs_homo_warp_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _s]),
    "mvs_homo_warp_bwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _s]),
    "mvs_conv3d_workspace_bytes": (_ll, [_i] * 8),
    "mvs_conv3d_pack_weights": (_i, [_i, _f, _f, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_conv3d_pack_weights_batch": (_i, [_i, C.POINTER(_i), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(_i), _s]),
    "mvs_conv3d_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _f, _i, _i, _s]),
    "mvs_conv3d_dgrad": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _s]),
    "mvs_conv3d_wgrad": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_convT3d_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _f, _i, _i, _s]),
    "mvs_convT3d_dgrad": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _s]),
    "mvs_convT3d_wgrad": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_bn_slots": (_i, [_i]),
    "mvs_bn_stats_slots": (_i, [_f, _i, _ll, _i, _f, _i, _s]),
    "mvs_bn_relu_fwd_slots": (_i, [_f, _f, _i, _i, _ll, _i, _f, _f, _fl, _fl, _f, _f, _f, _i, _f, _f, _s]),
    "mvs_bn_finalize_slots": (_i, [_f, _i, _i, _ll, _i, _f, _f, _fl, _fl, _f, _f, _f, _s]),
    "mvs_bn_bwd_reduce_slots": (_i, [_f, _f, _f, _i, _i, _ll, _i, _f, _i, _s]),
    "mvs_bn_relu_bwd_slots": (_i, [_f, _f, _f, _f, _i, _i, _i, _ll, _i, _f, _f, _f, _s]),
    "mvs_bn_eval_affine": (_i, [_f, _f, _f, _f, _fl, _i, _f, _f, _s]),
    "mvs_bn_relu_fwd": (_i, [_f, _f, _f, _f, _i, _ll, _i, _f, _s]),
    "mvs_masked_smooth_l1_fwd": (_i, [_f, _f, _f, _ll, _f, _s]),
    "mvs_masked_smooth_l1_bwd": (_i, [_f, _f, _f, _f, _f, _ll, _f, _s]),
    "mvs_softargmin_conf_fwd": (_i, [_f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _s]),
    "mvs_softargmin_conf_bwd": (_i, [_f, _f, _f, _i, _f, _f, _f, _i, _i, _i, _i, _f, _s]),
    "mvs_conv2d_workspace_floats": (_ll, [_i] * 8),
    "mvs_conv2d_fwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_conv2d_lrelu_fwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _fl, _s]),
    "mvs_conv2d_fwd_stats": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_conv2d_pack_weights_batch": (_i, [_i, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(_i), C.POINTER(_i), _s]),
    "mvs_conv2d_dgrad": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_conv2d_dgrad_bnstats": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _s]),
    "mvs_conv2d_wgrad": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_conv2d_wgrad_batch_workspace_floats": (_ll, [_i, C.POINTER(_i)]),
    "mvs_conv2d_wgrad_batch": (_i, [_i, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _f, C.POINTER(_i), _s]),
    "mvs_conv2d_wgrad_batch_xf": (_i, [_i, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _f,
                                       C.POINTER(_i), _s]),
    "mvs_conv2d_fwd_stats_xf": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _s]),
    "mvs_depth_hypo_workspace_doubles": (_ll, [_i, _i, _i]),
    "mvs_depth_hypo": (_i, [_f, _f, _i, _i, _i, _f, _f, _s]),
    "mvs_geo_consistency": (_i, [_f, C.POINTER(C.c_void_p), _f, _i, _i, _i, _fl, _fl, _f, _f, _f, _f, _f, _s]),
    "mvs_fusibile_fuse": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _fl, _fl, _i, _i, _f, _s]),
    "mvs_unsup_loss_workspace_floats": (_ll, [_i, _i, _i, _i]),
    "mvs_unsup_loss_fwd": (_i, [_f, C.POINTER(C.c_void_p), _f, _f, _f, _i, _i, _i, _i, _fl, _f, _f, _s]),
    "mvs_unsup_loss_bwd": (_i, [_f, C.POINTER(C.c_void_p), _f, _f, _f, _i, _i, _i, _i, _fl, _f, _f, _f, _s]),
}

OP_CONV_FWD, OP_CONV_DGRAD, OP_CONV_WGRAD, OP_CONVT_FWD, OP_CONVT_DGRAD, OP_CONVT_WGRAD = range(6)


class MvsLib:
    """Loaded library + checked calls.  ``device_type`` is the torch device type whose pointers the
    library accepts ("cuda" for the product)."""

    def __init__(self, path: str = LIB_PATH, device_type: str = "cuda"):
        if not os.path.exists(path):
            raise RuntimeError(
                "libmvs_hip.so not found at %s -- build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950); there is no CPU / PyTorch fallback for the hot path" % path)
        self.path = path
        self.device_type = device_type
        self.cdll = C.CDLL(path)
        self.profiler = None  # optional KernelTimer (HIP-event timing per C-ABI call, used by bench.py)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_" + name, fn)

    def call(self, name: str, *args, tag="", tstream=None):
        """tag: a string, or (format, *values) -- formatted only when a KernelTimer is attached (the hot path builds ~100 tags per
        training step that nobody reads).  tstream: the torch stream the kernel is enqueued on when that is not the current one
        (the HIP-event brackets of a KernelTimer go on that stream)."""
        prof = self.profiler
        if prof is not None and (prof.names is None or name in prof.names):
            if not isinstance(tag, str):
                tag = tag[0] % tuple(tag[1:])
            if prof.wants(name, tag):
                import torch
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(tstream)  # default: the current stream == the stream the kernel is enqueued on (ops._stream)
                rc = getattr(self, "_" + name)(*args)
                ev1.record(tstream)
                prof.add(name, tag, ev0, ev1)
                if rc != 0:
                    self._raise(name, rc)
                return
        rc = getattr(self, "_" + name)(*args)
        if rc != 0:
            self._raise(name, rc)

    def _raise(self, name, rc):
        msg = self._mvs_last_error().decode("utf-8", "replace")
        if rc in (-1, -2, -4):
            raise ValueError("%s failed (%d): %s" % (name, rc, msg))
        raise RuntimeError("%s failed (%d): %s" % (name, rc, msg))

    def raw(self, name: str, *args):
        return getattr(self, "_" + name)(*args)


_INSTANCE = None

# library defaults of the measurement knobs (csrc: g_conv_c8, g_conv_xcd); MVS_TUNING="k8=2,xcd=0" overrides them
# for A/B runs of bench.py / tools without touching code
DEFAULT_TUNING = {"k8": 7, "xcd": 1, "side_pre": 1, "conv_pers": 1, "conv_pers_min": 1024, "conv_pers_nw": 8, "wgrad_pers": 1, "conv_small": 1, "tr2pw": 1, "sweep_bwd": 0, "wgrad_small": 0, "wgrad_groups": 768, "wgrad8_groups": 192, "wgrad8_gs": 2, "wgrad8_nch": 2, "cout1_h4": 1, "wgrad2d_batch": 2048}


def get() -> MvsLib:
    global _INSTANCE
    if _INSTANCE is None:
        _INSTANCE = MvsLib()
        for item in filter(None, os.environ.get("MVS_TUNING", "").split(",")):
            key, _, val = item.partition("=")
            DEFAULT_TUNING[key.strip()] = int(val)
            _INSTANCE.call("mvs_set_tuning", key.strip().encode(), int(val))
    return _INSTANCE


class KernelTimer:
    """HIP-event timing of individual C-ABI calls on the stream they are launched on.
    ``only``: None = every call, or a set of call names / tags."""

    def __init__(self, only=None, names=None):
        self.only = only
        self.names = names      # entry-point names worth looking at (None: all): calls of other names skip even the tag formatting
        self.events = {}

    def wants(self, name, tag):
        return self.only is None or name in self.only or tag in self.only

    def add(self, name, tag, ev0, ev1):
        self.events.setdefault((name, tag), []).append((ev0, ev1))

    # ---- the regulariser as ONE C call per pass (ops.C_ENTRY): no per-layer Python call to bracket.  A timer that looks at exactly one
    # weight gradient of the regulariser (bench.py's roofline kernel) has it bracketed INSIDE mvs_unet_bwd instead (mvs_unet_time_wgrad)
    _REG_NAMES = frozenset(("mvs_conv3d_fwd", "mvs_convT3d_fwd", "mvs_conv3d_dgrad", "mvs_convT3d_dgrad", "mvs_conv3d_wgrad", "mvs_convT3d_wgrad",
                            "mvs_bn_relu_fwd_slots", "mvs_bn_relu_bwd_slots", "mvs_bn_bwd_reduce_slots", "mvs_conv3d_pack_weights_batch"))

    def allows_c_entry(self, lib, blocks):
        """blocks: [(transposed, cin, cout, stride, (b, d, h, w))] of the regulariser about to run (+ the prob layer last).  True: the C
        entry may run (nothing of it is timed, or the one weight gradient this timer wants is bracketed in C)."""
        if self.names is None:
            return False
        hit = self.names & self._REG_NAMES
        if not hit:
            return True
        if not hit <= {"mvs_conv3d_wgrad", "mvs_convT3d_wgrad"} or self.only is None:
            return False
        want = [t for t in self.only if isinstance(t, str) and t.startswith("wgrad")]
        found = []
        for i, (transposed, cin, cout, stride, dims) in enumerate(blocks):
            tag = "%s:%d>%d:s%d:%dx%dx%dx%d" % (("wgradT" if transposed else "wgrad", cin, cout, stride) + tuple(dims))
            if tag in want:
                found.append((i, "mvs_convT3d_wgrad" if transposed else "mvs_conv3d_wgrad", tag))
        if len(found) != 1 or len(want) != 1:
            return False
        owner = getattr(lib, "_unet_timer", None)
        if owner is not self:
            if owner is not None:
                owner._harvest()
                owner._c = None
            lib.raw("mvs_unet_time_read", (C.c_float * 1024)(), 1024)      # forget brackets nobody owns
            lib.raw("mvs_unet_time_wgrad", found[0][0])
            lib._unet_timer = self
            self._c = (lib, found[0])
        return True

    def _harvest(self):
        """collect the brackets mvs_unet_bwd has recorded for this timer so far"""
        c = getattr(self, "_c", None)
        if c is None:
            return
        lib, (idx, name, tag) = c
        buf = (C.c_float * 1024)()
        cnt = lib.raw("mvs_unet_time_read", buf, 1024)
        if cnt > 0:
            acc = self.__dict__.setdefault("_c_acc", {})
            n0, s0 = acc.get((name, tag), (0, 0.0))
            acc[(name, tag)] = (n0 + cnt, s0 + sum(buf[:cnt]))

    def release_c_bracket(self, lib):
        """the timer is no longer attached: collect what is there and switch the C-side bracket off"""
        self._harvest()
        self._c = None
        lib.raw("mvs_unet_time_wgrad", -2)
        lib._unet_timer = None

    def summary(self):
        """{(name, tag): (calls, mean_ms)} -- call after torch.cuda.synchronize()."""
        out = {}
        for k, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[k] = (len(ms), sum(ms) / len(ms))
        c = getattr(self, "_c", None)
        if c is not None:
            self.release_c_bracket(c[0])
        for (name, tag), (cnt, tot) in getattr(self, "_c_acc", {}).items():
            if (name, tag) in out:
                n0, m0 = out[(name, tag)]
                out[(name, tag)] = (n0 + cnt, (n0 * m0 + tot) / (n0 + cnt))
            else:
                out[(name, tag)] = (cnt, tot / cnt)
        return out
