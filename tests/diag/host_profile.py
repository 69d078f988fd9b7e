"""Where does the host time of one training step go?  (tiny input => GPU idle, cProfile of 20 steps)"""
import cProfile
import pstats
import sys
import torch
sys.path.insert(0, ".")
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer
torch.manual_seed(0)
tr = SegHieroTrainer(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01, device="cuda:0")
tr.train()
img, lab = make_batch(1, 64, 9, seed=0, device="cuda:0")
lab8 = ops.labels_u8(lab)
for _ in range(5):
    tr.train_step(img, lab8, 0)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)      # backward in this thread, so cProfile sees it
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.train_step(img, lab8, 0)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
