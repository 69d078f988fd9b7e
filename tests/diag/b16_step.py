"""bf16 compute mode vs the fp32-accurate HIP path on the same weights / inputs: step-0 losses, per-module gradient cosine and relative
error.  Usage: python tests/diag/b16_step.py [depth] [size] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
kw = dict(depth=depth, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01, device="cuda:0")
torch.manual_seed(0)
a = SegHieroTrainer(**kw)
b = SegHieroTrainer(compute_dtype=torch.bfloat16, **kw)
for k, m in a.modules().items():
    b.modules()[k].load_state_dict(m.state_dict())
a.train(); b.train()
img, lab = make_batch(batch, size, 9, seed=1, device="cuda:0")
la, lb = float(a.train_step(img, lab, 0)), float(b.train_step(img, lab, 0))
print(f"step-0 loss fp32-accurate {la:.6f}  bf16 compute {lb:.6f}  rel {abs(la - lb) / abs(la):.2e}")
for name in a.modules():
    pa, pb = dict(a.modules()[name].named_parameters()), dict(b.modules()[name].named_parameters())
    ga = torch.cat([p.grad.double().flatten() for p in pa.values()])
    gb = torch.cat([pb[k].grad.double().flatten() for k in pa])
    print(f"{name:10s} cosine {float(torch.dot(ga, gb) / (ga.norm() * gb.norm())):.5f}  rel-L2 {float((ga - gb).norm() / ga.norm()):.3e}  finite {bool(torch.isfinite(gb).all())}")
worst = []
for name in a.modules():
    pa, pb = dict(a.modules()[name].named_parameters()), dict(b.modules()[name].named_parameters())
    for k in pa:
        x, y = pa[k].grad.double().flatten(), pb[k].grad.double().flatten()
        worst.append((float(torch.dot(x, y) / (x.norm() * y.norm() + 1e-30)), name + "." + k))
worst.sort()
print("lowest per-tensor cosines:", worst[:6])
