"""TEST INFRASTRUCTURE: run the product's Python op wrappers against the CPU-emulated build of the
kernel sources (tests/cpu_emul).  Only tests use this; the product never loads the emulation."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, "tests", "cpu_emul")
EMUL_LIB = os.path.join(EMUL_DIR, "libmvs_emul.so")


def build_emul():
    r = subprocess.run(["make", "-C", EMUL_DIR], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulation build failed:\n" + r.stdout[-2000:] + r.stderr[-4000:])
    return EMUL_LIB


@pytest.fixture(scope="module")
def emul_lib():
    import mvs_amd
    from mvs_amd import _lib
    path = build_emul()
    lib = _lib.MvsLib(path, device_type="cpu")
    assert lib.raw("mvs_is_emulation") == 1
    old = _lib._INSTANCE
    _lib._INSTANCE = lib
    yield lib
    _lib._INSTANCE = old
