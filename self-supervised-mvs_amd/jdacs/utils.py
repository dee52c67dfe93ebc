"""Caller-side helpers of the path (jdacs/utils.py:36-76, jdacs/eval.py:125-165): recursive
tensor -> numpy conversion, checkpoint loading with the ``module.`` prefix DataParallel leaves, and writing the depth /
confidence maps as PFM files.  Device-agnostic: nothing here hard-codes ``.cuda()``."""
import os

import numpy as np
import torch

from .datasets.data_io import save_pfm


def _recursive(fn):
    def wrapper(v):
        if isinstance(v, list):
            return [wrapper(x) for x in v]
        if isinstance(v, tuple):
            return tuple(wrapper(x) for x in v)
        if isinstance(v, dict):
            return {k: wrapper(x) for k, x in v.items()}
        return fn(v)
    return wrapper


@_recursive
def tensor2numpy(v):
    if isinstance(v, np.ndarray):
        return v
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy().copy()
    raise NotImplementedError("invalid input type {} for tensor2numpy".format(type(v)))


@_recursive
def tensor2float(v):
    if isinstance(v, float):
        return v
    if isinstance(v, torch.Tensor):
        return v.data.item()
    raise NotImplementedError("invalid input type {} for tensor2float".format(type(v)))


def load_checkpoint(model, ckpt, strict=True):
    """``ckpt``: a path, the reference's ``{'model': state_dict, ...}`` dict (jdacs/train.py:169) or a bare state_dict; keys
    may carry the ``module.`` prefix of nn.DataParallel (jdacs/eval.py:136-137 loads them into a wrapped model)."""
    if isinstance(ckpt, (str, bytes, os.PathLike)):
        ckpt = torch.load(ckpt, map_location="cpu")
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict) else ckpt
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)


def write_depth_img(filename, depth):
    """jdacs/eval.py:110-123: the depth map as an 8-bit PNG, grey = (depth - 500) / 2 through PIL's float -> "L" conversion
    (the reference's own two calls; PIL is the reference's dependency for this file too).  Returns 1 like the reference."""
    from PIL import Image
    d = os.path.dirname(filename)
    if d and not os.path.exists(d):
        os.makedirs(d, exist_ok=True)
    Image.fromarray((np.asarray(depth) - 500) / 2).convert("L").save(filename)
    return 1


def save_depth_outputs(outputs, filenames, outdir, depth_png=False):
    """What jdacs/eval.py:150-164 does with a batch of outputs: ``{}/depth_est/{:0>8}.pfm``-style names (``filename`` is
    the dataset's format string with two slots) -> depth_est and confidence PFM files (+ `<depth>.pfm.png` through
    write_depth_img when depth_png, eval.py:165).  Returns the written paths."""
    outputs = tensor2numpy(outputs)
    written = []
    for name, depth, conf in zip(filenames, outputs["depth"], outputs["photometric_confidence"]):
        for kind, arr in (("depth_est", depth), ("confidence", conf)):
            path = os.path.join(outdir, name.format(kind, ".pfm"))
            os.makedirs(os.path.dirname(path), exist_ok=True)
            save_pfm(path, np.ascontiguousarray(arr, dtype=np.float32))
            written.append(path)
            if depth_png and kind == "depth_est":
                write_depth_img(path + ".png", np.ascontiguousarray(arr, dtype=np.float32))
                written.append(path + ".png")
    return written
