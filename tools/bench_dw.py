"""Depthwise 3x3 kernels on the decoder's shapes (16x128x128, C = 560 / 512): time and effective HBM rate.
   python tools/bench_dw.py            (SEGHIERO_DW_WALK=0 -> the per-tile kernels)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seghiero_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = "cuda:0"
for c in (560, 512):
    n, h, w = 16, 128, 128
    x = ops.new_act(n, c, h, w, dev); x.normal_()
    dy = ops.new_act(n, c, h, w, dev); dy.normal_()
    y = ops.new_act(n, c, h, w, dev)
    wt = torch.randn(c, 1, 3, 3, device=dev) / 3
    coefs = torch.stack([torch.zeros(c), torch.ones(c), torch.ones(c), torch.zeros(c)]).to(dev).contiguous()
    part = torch.empty((ops.dw_partials_rows(n, h, w), 2, c), device=dev)
    dw = torch.empty_like(wt)
    el = n * h * w * c * 4 / 1e6    # MB per tensor
    res = []
    for name, fn, nt in (("fprop", lambda: ops.dwconv_fprop(x, wt, y, part, 1), 2), ("fprop+aff", lambda: ops.dwconv_fprop(x, wt, y, part, 1, aff=coefs), 2),
                         ("dgrad", lambda: ops.dwconv_dgrad(dy, wt, y, 1), 2), ("dgrad+bnb", lambda: ops.dwconv_dgrad_bnb(dy, wt, y, x, coefs, part, 1), 3),
                         ("wgrad", lambda: ops.dwconv_wgrad(x, dy, dw, 1), 2), ("wgrad+aff", lambda: ops.dwconv_wgrad(x, dy, dw, 1, aff=coefs), 2)):
        us = timeit(fn)
        res.append(f"{name} {us:6.1f}us {nt * el / us:5.2f}TB/s")
    print(f"C={c}: " + " | ".join(res), flush=True)
