"""Sensitivity of the test network's parameter gradients to a 1e-6 relative perturbation of the input (single rank, no SyncBN):
how much of a 2-rank vs 1-rank difference is conditioning of the test problem itself."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import test_ddp_gpu as T
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
BB = int(sys.argv[2]) if len(sys.argv) > 2 else T.B
SS = int(sys.argv[3]) if len(sys.argv) > 3 else T.S
base = T._run(0, BB, sync=False, depth=depth, B=BB, S=SS)
_randn = torch.randn
def noisy(*a, **k):
    t = _randn(*a, **k)
    if len(a) == 4 and a[1] == 3:                     # the input image only
        t = t * (1 + 1e-6 * _randn(t.shape, generator=torch.Generator().manual_seed(99)))
    return t
torch.randn = noisy
pert = T._run(0, BB, sync=False, depth=depth, B=BB, S=SS)
torch.randn = _randn
print("forward logits rel", T._rel(pert["logits"], base["logits"]))
rows = sorted(((T._rel(pert["grads"][k], base["grads"][k]), k) for k in base["grads"]), reverse=True)
for e, k in rows[:8]: print(f"{e:.3e} {k}")
print("median", rows[len(rows)//2])
