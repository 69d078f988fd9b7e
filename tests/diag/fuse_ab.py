#!/usr/bin/env python3
"""Diagnostic (GPU box): gradients of the full model with the fused BatchNorm paths on vs off (same weights, same batch)."""
import os, sys, torch
sys.path.insert(0, ".")
from seghiero_amd import layers as L, ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer
torch.manual_seed(0)
kw = dict(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01)
a = SegHieroTrainer(device="cuda:0", **kw); b = SegHieroTrainer(device="cuda:0", **kw)
b.load_state_dicts({k: m.state_dict() for k, m in a.modules().items()})
a.train(); b.train()
img, lab = make_batch(4, 256, 9, seed=0, device="cuda:0")
def grads(tr, fuse):
    L.FUSE_BN = fuse
    loss, _, _, _ = tr.forward_loss(img, lab, 0); loss.backward(); torch.cuda.synchronize()
    return float(loss), {k + "." + n: p.grad.detach().double().clone() for k, m in tr.modules().items() for n, p in m.named_parameters()}
la, ga = grads(a, True); lb, gb = grads(b, False)
print("loss fused", la, "unfused", lb)
rows = sorted(((float((ga[k] - gb[k]).norm() / gb[k].norm().clamp_min(1e-30)), k) for k in ga), reverse=True)
for e, k in rows[:15]:
    print(f"  {k:55s} {e:.2e}")
