"""mvs_amd -- MI355X-native hot path of Self-Supervised-MVS (JDACS / JDACS-MS).

Plane-sweep cost-volume build (homography warp + variance), 3-D CNN regularisation and soft-argmin
depth regression as hand-written HIP kernels for gfx950 behind the reference's own
``nn.Module.forward`` surface:

    from mvs_amd.jdacs.models.mvsnet import MVSNet            # == jdacs/models/mvsnet.py::MVSNet
    from mvs_amd.jdacs_ms.models.network import CVPMVSNet     # == jdacs-ms/models/network.py::CVPMVSNet

The kernels live in ``libmvs_hip.so`` (C ABI: include/mvs_hip.h); there is no PyTorch/CPU fallback.
"""
from . import _lib, ops  # noqa: F401

__version__ = "0.1.0"


def library_path() -> str:
    return _lib.LIB_PATH
