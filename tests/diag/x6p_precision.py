#!/usr/bin/env python3
"""Precision diagnostic (GPU box): relative error of the conv kernels and their fused BatchNorm hooks against fp64 references,
next to the error of torch's own fp32 CPU conv.  Usage: SEGHIERO_X6P=0|1 python tests/diag/x6p_precision.py"""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from seghiero_amd import ops

DEV = "cuda:0"


def nhwc(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


CASES = [(4, 16, 16, 512, 128, 1, 1, 0), (4, 16, 16, 128, 128, 3, 1, 1), (4, 16, 16, 128, 512, 1, 1, 0),
         (4, 32, 32, 256, 64, 1, 1, 0), (4, 32, 32, 64, 64, 3, 1, 1), (2, 64, 64, 256, 256, 3, 2, 1), (2, 128, 128, 512, 512, 1, 1, 0)]
print("X6P =", os.environ.get("SEGHIERO_X6P", "1"))
for n, h, w, cin, cout, k, s, p in CASES:
    g = torch.Generator().manual_seed(n + h + cin + cout + k)
    raw = torch.randn(n, cin, h, w, generator=g)
    scale, shift = 0.5 + torch.rand(cin, generator=g), 0.3 * torch.randn(cin, generator=g)
    mean, invstd = 0.2 * torch.randn(cin, generator=g), 0.5 + torch.rand(cin, generator=g)
    coefs = torch.stack([mean, invstd, scale, shift]).to(DEV).contiguous()
    v = lambda t: t.view(1, -1, 1, 1)
    act = torch.relu(raw * v(scale) + v(shift))
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    a64, w64 = act.double().requires_grad_(True), wt.double().requires_grad_(True)
    y64 = F.conv2d(a64, w64, None, s, p)
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    a32, w32 = act.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    y32 = F.conv2d(a32, w32, None, s, p)
    y32.backward(dy)
    ho, wo = y64.shape[2:]
    rawg, actg, wg = nhwc(raw), nhwc(act), wt.to(DEV).contiguous(memory_format=torch.channels_last)
    y = ops.new_act(n, cout, ho, wo, DEV)
    ops.conv_fprop(actg, wg, None, y, None, s, p, 1)
    ya = ops.new_act(n, cout, ho, wo, DEV)
    ok_a = ops.conv_fprop_aff(rawg, coefs, wg, None, ya, None, s, p, 1)
    dyg = nhwc(dy)
    dx = ops.new_act(n, cin, h, w, DEV)
    ops.conv_dgrad(dyg, wg, dx, s, p, 1)
    dw = torch.empty_like(wg)
    ops.conv_wgrad(actg, dyg, dw, s, p, 1)
    line = (f"{n}x{h}x{w} {cin}->{cout} k{k} s{s}: fprop {rel(y, y64):.1e} (torch32 {rel(y32, y64):.1e}) aff {rel(ya, y64) if ok_a else -1:.1e} | "
            f"dgrad {rel(dx, a64.grad):.1e} (t32 {rel(a32.grad, a64.grad):.1e}) | wgrad {rel(dw, w64.grad):.1e} (t32 {rel(w32.grad, w64.grad):.1e})")
    if ops.wgrad_aff_ok(rawg, wg, s, p, 1):
        dwa = torch.empty_like(wg)
        ops.conv_wgrad(rawg, dyg, dwa, s, p, 1, aff=coefs)
        line += f" aff {rel(dwa, w64.grad):.1e}"
    if s == 1:
        gb = ops.new_act(n, cin, h, w, DEV)
        bp = torch.empty((-(-n * h * w // 64), 2, cin), device=DEV)
        if ops.conv_dgrad_bnb(dyg, wg, gb, rawg, coefs, True, bp, s, p, 1):
            mask = (raw * v(scale) + v(shift) > 0).double()
            g64 = a64.grad * mask
            xh = (raw.double() - v(mean).double()) * v(invstd).double()
            gk = gb.cpu().double()                               # the kernel's own g: isolates the summation from the dgrad error
            s_self, q_self = gk.sum((0, 2, 3)), (gk * xh).sum((0, 2, 3))
            sums = bp.double().sum(0).cpu()
            line += (f" | bnb g {rel(gb, g64):.1e} sum-vs-own-g {rel(sums[0], s_self):.1e} {rel(sums[1], q_self):.1e}"
                     f" sums-vs-f64 {rel(sums[0], g64.sum((0, 2, 3))):.1e} {rel(sums[1], (g64 * xh).sum((0, 2, 3))):.1e}")
    print(line, flush=True)
