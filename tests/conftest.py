import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        t = torch.from_numpy(a)
        if a.dtype == np.float16:
            t = t.float()
        out[k] = t
    return out


def state_dict_from(gold, prefix="sd."):
    return {k[len(prefix):]: v for k, v in gold.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_l1(a, b):
    return float((a - b).abs().mean() / b.abs().mean().clamp_min(1e-12))
