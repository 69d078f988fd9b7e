#!/usr/bin/env python3
"""The north-star unit: one ASPP depthwise-separable branch forward at B=16, 512^2 input (c4 = [16, 2048, 16, 16]):
depthwise 3x3 (dilation d) -> BN -> ReLU -> pointwise 2048->512 -> BN -> ReLU, train-mode statistics.
Algorithmic traffic 46.2 MB and 8.74 GF => 5.8 us at 8 TB/s, 21 us at the 416.7 TF x6 peak, 55.6 us at the f32 MFMA peak."""
import sys
import torch
import torch.nn as nn
sys.path.insert(0, ".")
from seghiero_amd import layers as L, ops

DEV = "cuda:0"
torch.manual_seed(0)
x = ops.new_act(16, 2048, 16, 16, DEV); x.normal_()
dw = torch.randn(2048, 1, 3, 3, device=DEV) / 3
pw = (torch.randn(512, 2048, 1, 1, device=DEV) / 45).contiguous(memory_format=torch.channels_last)
bn_dw, bn_pw = nn.BatchNorm2d(2048).to(DEV), nn.BatchNorm2d(512).to(DEV)
out = ops.new_act(16, 512, 16, 16, DEV)


def branch(d):
    t, _ = L.dw_fwd(x, dw, d, bn_dw, True)
    L.cba_fwd(t, pw, (1, 0, 1), bn_pw, True, True, out=out)


for d in (12, 24, 36):
    for _ in range(5):
        branch(d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        branch(d)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    with ops.profile() as prof:
        branch(d)
    parts = ", ".join("%s %.1f" % (k.replace("sh_", ""), v["ms"] * 1e3) for k, v in prof.rows.items())
    print("dilation %2d: %.1f us per branch (%.1f %% of the 8 TB/s bound, %.1f %% of the x6 MFMA bound)   kernels [us]: %s"
          % (d, us, 100 * 5.8 / us, 100 * 21.0 / us, parts))
