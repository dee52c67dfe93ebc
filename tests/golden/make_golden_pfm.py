#!/usr/bin/env python3
"""tests/golden/g9_pfm.npz: the bytes the reference's save_pfm (jdacs/datasets/data_io.py:53-80) writes for a seeded depth
map and a 3-channel image (build container only; pure numpy code of the reference, imported, nothing copied)."""
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/jdacs")
from datasets.data_io import read_pfm, save_pfm  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(9)
depth = (425.0 + 500.0 * rng.rand(32, 40)).astype(np.float32)
color = rng.randn(5, 7, 3).astype(np.float32)
out = {"depth": depth, "color": color}
with tempfile.TemporaryDirectory() as d:
    for k, v, sc in (("depth", depth, 1), ("color", color, 2.5)):
        p = os.path.join(d, k + ".pfm")
        save_pfm(p, v, sc)
        out[k + "_bytes"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        back, scale = read_pfm(p)
        assert np.array_equal(back, v) and scale == sc
np.savez_compressed(os.path.join(HERE, "g9_pfm.npz"), **out)
print("g9_pfm.npz", os.path.getsize(os.path.join(HERE, "g9_pfm.npz")), "bytes")
