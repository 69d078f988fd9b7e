"""PIL ``Image.resize(size, Image.BILINEAR)`` on 8-bit images, restated (oracle).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Reference call site: ``dataset/dataloader.py:50`` (``img.resize(self.resize, Image.BILINEAR)``).  The arithmetic is a third-party
dependency of the reference (Pillow, un-pinned in ``requirements.txt``); Pillow 12.2.0 is importable in the authoring container
and this restatement of its published algorithm (``src/libImaging/Resample.c``: ``precompute_coeffs``, ``normalize_coeffs_8bpc``,
``ImagingResampleHorizontal_8bpc`` / ``Vertical_8bpc``) is pinned bit-exactly against arrays produced by Pillow itself
(``tools/make_goldens.py`` g9 -> ``tests/golden/g9_pil_bilinear.npz``).

Algorithm: separable triangle filter whose support is stretched by the downscale factor (antialiasing), horizontal pass first,
each pass rounding to uint8: coefficients in float64 normalised to sum 1, converted to fixed point with 22 fractional bits
(round half away from zero), accumulated in int32 from 2^21, shifted, clipped to 0..255.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size, out_size):
    """-> (bounds int32 [out, 2] = (first tap, tap count), kk int32 [out, ksize]) of one axis (box = whole image)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                      # bilinear: filter support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            v = 1.0 - t if t < 1.0 else 0.0
            w.append(v)
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(0.5 + v * (1 << PRECISION_BITS)) if v >= 0 else int(-0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """img uint8 [..]: resample along `axis`."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_bilinear_resize_u8(img, size):
    """img uint8 [H, W, C]; size = (W_out, H_out) as PIL takes it.  -> uint8 [H_out, W_out, C]."""
    h, w = img.shape[:2]
    wo, ho = size
    out = img
    if wo != w:
        bx, kx = coeffs(w, wo)
        out = _pass(out, bx, kx, 1)
    if ho != h:
        by, ky = coeffs(h, ho)
        out = _pass(out, by, ky, 0)
    return out
