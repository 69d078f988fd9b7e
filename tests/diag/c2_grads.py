#!/usr/bin/env python3
"""Diagnostic (GPU box): per-parameter gradient error of the full C2 model (R50, 512^2, B=2) vs the CPU oracle, worst first."""
import os, sys, torch
sys.path.insert(0, ".")
from oracle.step import OracleTrainer
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer
from seghiero_amd import ops
torch.manual_seed(0)
kw = dict(depth=int(os.environ.get("DEPTH", "50")), n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01)
ref = OracleTrainer(**kw); mine = SegHieroTrainer(device="cuda:0", **kw)
mine.load_state_dicts(ref.state_dicts()); ref.train(); mine.train()
size = int(os.environ.get("SIZE", "512"))
img, lab = make_batch(2, size, 9, seed=0)
lr_, _, _, _ = ref.forward_loss(img, lab, 0); lr_.backward()
ops.prepare_dgrad_weights(mine._dgrad_weights, mine._wt_cache)
lm, _, _, _ = mine.forward_loss(img.cuda(), lab.cuda(), 0); lm.backward(); ops.release_dgrad_weights()
torch.cuda.synchronize()
print({k: os.environ.get(k) for k in ("SEGHIERO_X6P", "SEGHIERO_FUSE_BN", "SEGHIERO_X6P_VEC", "SEGHIERO_WGRAD_STREAM")}, "loss", float(lm), float(lr_))
rows = []
for name, m in ref.modules().items():
    pm = dict(mine.modules()[name].named_parameters())
    for k, p in m.named_parameters():
        a, b = pm[k].grad.detach().cpu().double(), p.grad.double()
        rows.append((float((a - b).norm() / b.norm().clamp_min(1e-30)), float(b.norm()), f"{name}.{k}"))
rows.sort(reverse=True)
for e, nb, k in rows[:12]:
    print(f"  {k:50s} relerr {e:.2e}  |grad| {nb:.3e}")
