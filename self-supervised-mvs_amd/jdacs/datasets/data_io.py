"""PFM reader / writer with the reference's byte format and call signatures (jdacs/datasets/data_io.py:15-80), the
on-disk format of the path's outputs (depth and confidence maps, jdacs/eval.py:155-164).

File layout: ``Pf\\n`` (1 channel) or ``PF\\n`` (3 channels), ``"{W} {H}\\n"``, ``"%f\\n" % scale`` with a NEGATIVE scale for
little-endian data, then H*W*(1|3) fp32 values, rows stored bottom-up."""
import re
import sys

import numpy as np


def save_pfm(filename, image, scale=1):
    """image: float32 [H,W], [H,W,1] or [H,W,3].  Returns None, raises on other dtypes / shapes like the reference."""
    image = np.asarray(image)
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        magic = b"PF\n"
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        magic = b"Pf\n"
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    header = magic + ("%d %d\n" % (image.shape[1], image.shape[0])).encode("utf-8") \
        + ("%f\n" % (-scale if little else scale)).encode("utf-8")
    with open(filename, "wb") as fh:
        fh.write(header)
        fh.write(np.ascontiguousarray(image[::-1]).tobytes())      # bottom row first


def read_pfm(filename):
    """-> (array [H,W] or [H,W,3] float32 in file byte order, top row first; scale > 0)."""
    with open(filename, "rb") as fh:
        magic = fh.readline().decode("utf-8").rstrip()
        if magic not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        dims = re.match(r"^(\d+)\s(\d+)\s$", fh.readline().decode("utf-8"))
        if not dims:
            raise Exception("Malformed PFM header.")
        width, height = int(dims.group(1)), int(dims.group(2))
        scale = float(fh.readline().rstrip())
        order = "<" if scale < 0 else ">"
        data = np.frombuffer(fh.read(), dtype=order + "f4")
    shape = (height, width, 3) if magic == "PF" else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)
