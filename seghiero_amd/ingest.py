"""On-device half of the reference's input pipeline (``dataset/dataloader.py:36-63`` ``JointTransform`` and the
``ToTensor`` / ``Normalize`` of ``:176-180``), batched: SURVEY 8f row 3.

PIL decoding stays on the host; the image enters as decoded interleaved RGB ``uint8 [N,H,W,3]``.  Everything after that runs
on the device: PIL's antialiasing ``img.resize(size, Image.BILINEAR)`` (``:50``; two integer passes, bit-identical to Pillow,
``resize_bilinear_u8``), per-sample horizontal flip, ``/255``, ``(v - mean) / std`` written directly in the stem's NHWC4 fp32
layout, and for the label map ``F.interpolate(mode="nearest")`` + the same flip + int64 -> uint8.  The arithmetic is the
reference's (same operation order), so results are bit-identical to ``ToTensor`` + ``Normalize`` on the same pixels.
"""
import ctypes

import torch

from . import ops
from ._lib import LIB, SegHieroHipError

_TABLES = {}


def _resize_tables(in_size, out_size, device):
    """(bounds, kk, ksize) device tables of one axis: Pillow's coefficients (host arithmetic in the C library), cached."""
    key = (in_size, out_size, str(device))
    hit = _TABLES.get(key)
    if hit is None:
        fn = LIB.raw("sh_resize_bilinear_coeffs")
        ksize = fn(in_size, out_size, None, None, 0)
        if ksize <= 0:
            raise SegHieroHipError("sh_resize_bilinear_coeffs rejected the sizes")
        bounds = (ctypes.c_int * (2 * out_size))()
        kk = (ctypes.c_int * (ksize * out_size))()
        if fn(in_size, out_size, bounds, kk, ksize * out_size) != ksize:
            raise SegHieroHipError("sh_resize_bilinear_coeffs failed (downscale factor beyond 31x?)")
        hit = (torch.tensor(list(bounds), dtype=torch.int32, device=device), torch.tensor(list(kk), dtype=torch.int32, device=device), ksize)
        _TABLES[key] = hit
    return hit


def resize_bilinear_u8(rgb_u8, size):
    """``PIL.Image.resize(size, Image.BILINEAR)`` (``dataset/dataloader.py:50``) of a batch of interleaved RGB uint8 images
    ``[N,H,W,3]`` on the device, bit-identical to Pillow.  size = (W_out, H_out), as PIL takes it."""
    ops._require_gpu(rgb_u8)
    if rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 4 or rgb_u8.shape[3] != 3 or not rgb_u8.is_contiguous():
        raise SegHieroHipError("rgb_u8 must be a contiguous uint8 [N,H,W,3] tensor")
    n, h, w, _ = rgb_u8.shape
    wo, ho = int(size[0]), int(size[1])
    dev = rgb_u8.device
    out = torch.empty((n, ho, wo, 3), dtype=torch.uint8, device=dev)
    bx = kx = by = ky = None
    ksx = ksy = 0
    if wo != w:
        bx, kx, ksx = _resize_tables(w, wo, dev)
    if ho != h:
        by, ky, ksy = _resize_tables(h, ho, dev)
    tmp = torch.empty((n, h, wo, 3), dtype=torch.uint8, device=dev) if (wo != w and ho != h) else None
    p = lambda t: None if t is None else t.data_ptr()
    ops._call("sh_resize_bilinear_u8", rgb_u8.data_ptr(), p(tmp), out.data_ptr(), n, h, w, ho, wo, p(bx), p(kx), ksx, p(by), p(ky), ksy,
              ops._st())
    return out


class JointTransformDevice:
    """Batched, on-device ``JointTransform(resize, hflip_prob, normalize_mean, normalize_std)``.

    ``__call__(rgb_u8, mask, generator=None)``: ``rgb_u8`` uint8 ``[N,H,W,3]`` (device), ``mask`` int64 or uint8
    ``[N,Hs,Ws]`` (device).  Returns ``(img, labels)``: ``img`` is a logical ``[N,4,H,W]`` channels_last fp32 tensor (RGB +
    a zero channel) that ``ResNetBackbone`` consumes without any further layout pass, ``labels`` uint8 ``[N,H,W]``.
    One flip decision per sample, ``torch.rand(N, generator) < hflip_prob`` (the reference draws ``torch.rand(1)`` per
    sample, dataloader.py:57)."""

    def __init__(self, resize=None, hflip_prob=0.5, normalize_mean=(0.485, 0.456, 0.406), normalize_std=(0.229, 0.224, 0.225)):
        self.resize = resize
        self.hflip_prob = hflip_prob
        self.normalize_mean = tuple(float(v) for v in normalize_mean)
        self.normalize_std = tuple(float(v) for v in normalize_std)

    def __call__(self, rgb_u8, mask, generator=None):
        ops._require_gpu(rgb_u8)
        if rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 4 or rgb_u8.shape[3] != 3 or not rgb_u8.is_contiguous():
            raise SegHieroHipError("rgb_u8 must be a contiguous uint8 [N,H,W,3] tensor")
        if self.resize is not None and (rgb_u8.shape[2], rgb_u8.shape[1]) != tuple(self.resize):
            rgb_u8 = resize_bilinear_u8(rgb_u8, self.resize)          # img.resize(self.resize, Image.BILINEAR), dataloader.py:50
        n, h, w, _ = rgb_u8.shape
        if mask.dtype not in (torch.int64, torch.uint8) or mask.dim() != 3 or mask.shape[0] != n:
            raise SegHieroHipError("mask must be int64 or uint8 [N,Hs,Ws]")
        mask = mask.contiguous()
        flip = None
        if self.hflip_prob > 0:
            flip = (torch.rand(n, generator=generator) < self.hflip_prob).to(torch.uint8).to(rgb_u8.device)
        img = ops.new_act(n, 4, h, w, rgb_u8.device)
        mean = (ctypes.c_float * 3)(*self.normalize_mean)
        std = (ctypes.c_float * 3)(*self.normalize_std)
        fp = None if flip is None else flip.data_ptr()
        ops._call("sh_ingest_image_u8", rgb_u8.data_ptr(), img.data_ptr(), fp, n, h, w, mean, std, ops._st())
        labels = torch.empty((n, h, w), device=rgb_u8.device, dtype=torch.uint8)
        ops._call("sh_ingest_mask", mask.data_ptr(), int(mask.dtype == torch.int64), labels.data_ptr(), fp, n,
                  mask.shape[1], mask.shape[2], h, w, ops._st())
        return img, labels
