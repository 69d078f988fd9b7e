"""Informal comparison point (NOT the product, NOT the CPU baseline): the oracle's plain torch ops moved to the GPU,
i.e. what the reference's own code path (ATen + MIOpen, fp32) does on this MI355X for the C2 step."""
import sys, time, torch
sys.path.insert(0, ".")
from oracle.step import OracleTrainer
from seghiero_amd.synthetic import make_batch
torch.backends.cudnn.benchmark = True
dev = "cuda:0"
tr = OracleTrainer(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]])
for m in tr.modules().values():
    m.to(dev)
tr.hiera_loss_fn.to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
img, lab = make_batch(B, 512, 9, seed=0, device=dev)
for cl in (False, True):
    if cl:
        for m in tr.modules().values():
            m.to(memory_format=torch.channels_last)
        img = img.contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        tr.train_step(img, lab, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        tr.train_step(img, lab, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"torch eager fp32 on GPU (channels_last={cl}): {dt*1e3:.1f} ms/step, {B/dt:.1f} img/s", flush=True)
