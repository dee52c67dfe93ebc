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
