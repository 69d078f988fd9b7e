"""Distance of the 3-level loss's logits gradient from the reference goldens (G6), per case: max|err| / max|ref|."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from seghiero_amd import loss
g = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", "g6_rmi_hiera_triplet_loss.npz"))
F2M, F2H = [0, 1, 1, 1, 1, 2, 2], [0, 1, 1, 1, 1, 1, 1]
for tag in ("even", "odd"):
    for lam in (0.0, 0.5):
        for step in (0, 30000):
            fn = loss.RMIHieraTripletLoss(7, 3, 2, torch.tensor(F2M), torch.tensor(F2H), loss_weight_lambda=lam).cuda()
            z = torch.tensor(g[f"{tag}_z"]).cuda().requires_grad_(True)
            e = torch.tensor(g[f"{tag}_emb"]).cuda().requires_grad_(True)
            val = fn(torch.tensor([step]), e, None, z, torch.tensor(g[f"{tag}_lab"]).cuda())
            val.backward()
            ref = g[f"{tag}_lam{lam}_s{step}_dz"]
            d = np.abs(z.grad.cpu().numpy() - ref)
            print(tag, lam, step, "max|err|/max|ref| %.3e" % (d.max() / np.abs(ref).max()),
                  "rel-l2 %.3e" % (np.linalg.norm(d) / np.linalg.norm(ref)), flush=True)
