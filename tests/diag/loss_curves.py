"""Training dynamics of the two modes on the bench workload (diagnosis): the same seeded trainer and the same four synthetic batches cycled for
N steps in fp32-accurate and in bf16 compute mode; prints the loss every 25 steps.   python tests/diag/loss_curves.py [steps]"""
import sys
import torch
sys.path.insert(0, ".")
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = dict(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]])
batches = []
for s in range(4):
    img, lab = make_batch(16, 512, 9, seed=s, device="cuda:0")
    batches.append((img, ops.labels_u8(lab)))
curves = {}
for mode, dt in (("f32", torch.float32), ("b16", torch.bfloat16)):
    torch.manual_seed(0)
    tr = SegHieroTrainer(lr=0.01, device="cuda:0", compute_dtype=dt, **cfg)
    tr.train()
    out = []
    for i in range(steps):
        img, lab8 = batches[i % 4]
        l = tr.train_step(img, lab8, i)
        if i % 25 == 0 or i == steps - 1:
            out.append((i, float(l)))
    curves[mode] = out
for (i, a), (_, b) in zip(curves["f32"], curves["b16"]):
    print(f"step {i:4d}: fp32-accurate {a:9.5f}   bf16 compute {b:9.5f}   diff {b - a:+.5f}")
