"""Per (C-ABI entry point, layer shape) time of one training step on the bench configuration (side stream off), sorted by time:
   python tools/shape_table.py [batch]"""
import os, sys
os.environ.setdefault("SEGHIERO_WGRAD_STREAM", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
b16 = len(sys.argv) > 2 and sys.argv[2] == "b16"          # bf16 compute mode
tr = SegHieroTrainer(device="cuda:0", depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01,
                     compute_dtype=torch.bfloat16 if b16 else torch.float32)
tr.train()
img, lab = make_batch(batch, 512, 9, seed=0)
img, lab = img.cuda(), ops.labels_u8(lab.cuda())
for _ in range(3):
    tr.train_step(img, lab, 0)
with ops.profile() as prof:
    tr.train_step(img, lab, 0)
tot = sum(v["ms"] for v in prof.rows.values())
print(f"total {tot:.2f} ms")
for (name, key), r in sorted(prof.shapes.items(), key=lambda kv: -kv[1]["ms"]):
    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["flops"] else 0.0
    gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["bytes"] else 0.0
    print(f"{name:26s} {str(key):34s} x{r['calls']:<2d} {r['ms']:7.3f} ms  {tf:6.1f} TF  {gbs:6.0f} GB/s(alg)")
