import sys, time, torch
sys.path.insert(0, ".")
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer
torch.manual_seed(0)
tr = SegHieroTrainer(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01, device="cuda:0")
tr.train()
img, lab = make_batch(16, 512, 9, seed=0, device="cuda:0")
lab8 = ops.labels_u8(lab)
for _ in range(5):
    tr.train_step(img, lab8, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.train_step(img, lab8, 0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) * 50, (t2 - t0) * 50))
# host-only cost: tiny batch so the GPU is never the bottleneck
img2, lab2 = make_batch(1, 64, 9, seed=0, device="cuda:0")
lab28 = ops.labels_u8(lab2)
for _ in range(5):
    tr.train_step(img2, lab28, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.train_step(img2, lab28, 0)
torch.cuda.synchronize()
print("tiny-input step (host-bound) %.2f ms/step" % ((time.perf_counter() - t0) * 50))
