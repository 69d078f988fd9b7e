"""Where does trip_class_kernel's time go?  Time sh_triplet_fwd at the step's shape (16 x 256 x 16 x 16 embedding, 512^2 labels, 9 fine / 4
coarse) for max_triplet = 200 / 50 / 8 (the triplet loop shrinks, the compaction scan does not)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from seghiero_amd import ops
from seghiero_amd.loss import TreeTripletLoss
from seghiero_amd.hierarchy import build_fine_to_coarse_map, build_hiera_index
from seghiero_amd.synthetic import make_batch

dev = torch.device("cuda:0")
c2f = [[0, 3], [4, 6], [7], [8]]
f2c = build_fine_to_coarse_map(c2f, 9)
mod = TreeTripletLoss(9, f2c.tolist(), build_hiera_index(c2f))
masks, ok = mod.tables(dev)
_, lab = make_batch(16, 512, 9, seed=0, device=dev)
lab8 = ops.labels_u8(lab)
emb = torch.nn.functional.normalize(torch.randn(16, 256, 16, 16, device=dev), dim=1).contiguous(memory_format=torch.channels_last)
for mt in (200, 50, 8):
    for _ in range(3):
        ops.triplet_fwd(emb, lab8, masks, ok, mt, 0.6)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out, ws = ops.triplet_fwd(emb, lab8, masks, ok, mt, 0.6)
    e1.record(); torch.cuda.synchronize()
    print(f"max_triplet {mt:4d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per sh_triplet_fwd (labels + class + finalize), out {out.tolist()}")
