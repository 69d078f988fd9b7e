import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from seghiero_amd.head import DepthwiseSeparableASPPContrastHead
DEV = "cuda:0"
def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
torch.manual_seed(4)
kw = dict(in_channels=256, c1_in_channels=64, c1_channels=48, aspp_channels=128, dilations=(1, 12, 24, 36), num_classes=13, proj_dim=64, proj_type="convmlp")
a = DepthwiseSeparableASPPContrastHead(**kw).to(DEV).train()
b = copy.deepcopy(a); b.act_dtype = torch.bfloat16
g = torch.Generator().manual_seed(9)
c1 = torch.randn(2, 64, 64, 64, generator=g).relu().to(DEV)
c4 = torch.randn(2, 256, 8, 8, generator=g).relu().to(DEV)
gl = torch.randn(2, 13, 64, 64, generator=g).to(DEV)
res = []
for head in (a, b):
    x1, x4 = c1.clone().requires_grad_(True), c4.clone().requires_grad_(True)
    logits, emb = head([x1, None, None, x4])
    (logits * gl).sum().backward()
    res.append({k: p.grad.clone() for k, p in head.named_parameters() if p.grad is not None})
for k in res[0]:
    e = relerr(res[1][k], res[0][k])
    print(f"{e:9.3e} |ref| {float(res[0][k].norm()):9.3e} |bf| {float(res[1][k].norm()):9.3e}  {k}")
