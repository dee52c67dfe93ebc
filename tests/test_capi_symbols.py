"""CPU: the C-ABI library loads and exports every symbol include/mvs_hip.h declares (no compute)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "mvs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mvs_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    import mvs_amd
    from mvs_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.MvsLib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib.cdll, n), "not exported: " + n
        assert n in _lib.SIGNATURES, "declared in mvs_hip.h but not bound in _lib.py: " + n
    for n in _lib.SIGNATURES:
        assert n in names, "bound but not declared in mvs_hip.h: " + n
    assert lib.raw("mvs_version") == 100
    assert lib.raw("mvs_is_emulation") == 0


def test_product_fails_loudly_without_library(tmp_path):
    from mvs_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        _lib.MvsLib(str(tmp_path / "missing.so"))


def test_cpu_tensors_are_rejected():
    import torch
    from mvs_amd import _lib, ops
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    _lib._INSTANCE = None
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.softargmin_conf(torch.zeros(1, 4, 2, 2), torch.zeros(1, 4))


def test_error_convention():
    """Bad arguments come back as negative codes + message, raised as ValueError by the wrapper."""
    from mvs_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    lib = _lib.MvsLib()
    with pytest.raises(ValueError, match="null pointer"):
        lib.call("mvs_softargmin_conf_fwd", None, None, 0, 1, 4, 2, 2, None, None, None, None, None)
    assert lib.raw("mvs_conv3d_workspace_bytes", 99, 1, 8, 8, 8, 8, 8, 1) == -1


def test_error_convention_of_every_entry_point():
    """Each compute entry point validates its arguments on the host BEFORE launching anything: null pointers and shapes
    the kernels do not support come back as a negative code with a message (ValueError in the wrapper), never as a crash.
    Runs against the product library on the CPU (validation happens before any device call)."""
    from mvs_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    lib = _lib.MvsLib()
    null_calls = [
        ("mvs_plane_sweep_variance_fwd", (None, None, None, None, None, 0, 1, 3, 32, 8, 8, 8, 0, 0, None, None)),
        ("mvs_homo_warp_fwd", (None, None, None, None, 0, 1, 8, 4, 8, 8, 0, None, None)),
        ("mvs_conv3d_fwd", (None, None, None, None, 1, 8, 8, 16, 8, 8, 1, None, None, None, 0, None, 0, 0, None)),
        ("mvs_conv3d_dgrad", (None, None, None, None, None, 1, 8, 8, 16, 8, 8, 1, None, None, None, 0, 0, None)),
        ("mvs_convT3d_dgrad", (None, None, None, None, None, 1, 4, 4, 8, 16, 8, 2, None, None, None, 0, 0, None)),
        ("mvs_conv3d_wgrad", (None, None, None, None, 1, 8, 8, 16, 8, 8, 1, None)),
        ("mvs_convT3d_fwd", (None, None, None, None, 1, 4, 4, 8, 16, 8, 2, None, None, None, 0, None, 0, 0, None)),
        ("mvs_conv3d_pack_weights", (0, None, None, 1, 8, 8, 16, 8, 8, 1, None)),
        ("mvs_bn_stats_slots", (None, 1, 64, 8, None, 16, None)),
        ("mvs_bn_relu_fwd_slots", (None, None, 16, 1, 64, 8, None, None, 1e-5, 0.1, None, None, None, 1, None, None, None)),
        ("mvs_bn_bwd_reduce_slots", (None, None, None, 1, 1, 64, 8, None, 16, None)),
        ("mvs_bn_relu_bwd_slots", (None, None, None, None, 16, 1, 1, 64, 8, None, None, None, None)),
        ("mvs_bn_relu_fwd", (None, None, None, None, 1, 64, 8, None, None)),
        ("mvs_softargmin_conf_bwd", (None, None, None, 0, None, None, None, 1, 4, 2, 2, None, None)),
        ("mvs_unsup_loss_fwd", (None, None, None, None, None, 1, 4, 8, 8, 1.0, None, None, None)),
        ("mvs_unsup_loss_bwd", (None, None, None, None, None, 1, 4, 8, 8, 1.0, None, None, None, None)),
        ("mvs_depth_hypo", (None, None, 1, 8, 8, None, None, None)),
        ("mvs_relative_projection", (None, None, 1, 2, None, None, None)),
    ]
    for name, args in null_calls:
        with pytest.raises(ValueError):
            lib.call(name, *args)
        assert lib.raw("mvs_last_error"), name
    # shape / size queries answer -1 for impossible shapes instead of a size
    assert lib.raw("mvs_unsup_loss_workspace_floats", 1, 4, 2, 8) == -1
    assert lib.raw("mvs_depth_hypo_workspace_doubles", 0, 8, 8) == -1
    assert lib.raw("mvs_unsup_loss_workspace_floats", 2, 4, 16, 20) == 4 * 2 * 320 * 4 + (4 * 4 + 2) * 3 + 64 + 2 * 2 * 14 * 18 * 9
    assert [lib.raw("mvs_bn_slots", c) for c in (4, 8, 16, 32, 64, 12)] == [128, 128, 64, 32, 16, -1]
    # unknown tuning key
    with pytest.raises(ValueError, match="unknown key"):
        lib.call("mvs_set_tuning", b"zz_no_such_knob", 1)


def test_default_tuning_table_matches_the_library():
    """_lib.DEFAULT_TUNING (what `bench.py --ab key=value` restores after a toggled run, and what MVS_TUNING overrides) against the
    values a freshly loaded library reports through mvs_get_tuning: a knob whose compiled default moved without the table makes
    every later A/B "default" run something else."""
    import ctypes as C
    from mvs_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    lib = _lib.MvsLib()       # a fresh handle: the same process-wide globals, which no test leaves changed
    bad = {}
    for key, want in _lib.DEFAULT_TUNING.items():
        got = C.c_int(-12345)
        lib.call("mvs_get_tuning", key.encode(), C.byref(got))
        if got.value != want:
            bad[key] = (got.value, want)
    assert not bad, "library default != _lib.DEFAULT_TUNING: %r" % bad
    with pytest.raises(ValueError, match="unknown key"):
        lib.call("mvs_get_tuning", b"no_such_knob", C.byref(C.c_int()))
