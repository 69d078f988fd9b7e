#!/usr/bin/env python3
"""Bisect diagnostic (GPU box): stem / layer1 weight gradients of the R18 trunk at 256^2 vs the oracle, under the env switches."""
import os, sys, torch
sys.path.insert(0, ".")
from oracle import nets
from seghiero_amd.backbone import ResNetBackbone
torch.manual_seed(0)
ref = nets.ResNetBackbone(18, pretrained=False).train()
mine = ResNetBackbone(18, pretrained=False)
mine.load_state_dict(ref.state_dict()); mine.to("cuda:0").train()
x = torch.randn(2, 3, 256, 256)
outs = ref(x); gs = [torch.randn(o.shape) for o in outs]
sum((o * g).sum() for o, g in zip(outs, gs)).backward()
om = mine(x.cuda()); sum((o * g.cuda()).sum() for o, g in zip(om, gs)).backward()
rel = lambda a, b: float((a.cpu().double() - b.double()).norm() / b.double().norm())
pm, pr = dict(mine.named_parameters()), dict(ref.named_parameters())
print({k: os.environ.get(k) for k in ("SEGHIERO_X6P", "SEGHIERO_FUSE_BN", "SEGHIERO_X6P_VEC")})
for k in ("stem_conv.weight", "stem_bn.weight", "stem_bn.bias", "layer1.0.conv1.weight", "layer2.0.conv1.weight", "layer4.1.conv2.weight"):
    print(f"  {k:28s} {rel(pm[k].grad, pr[k].grad):.2e}")
for i, (a, b) in enumerate(zip(om, outs)):
    print(f"  c{i+1} {rel(a, b):.2e}")
