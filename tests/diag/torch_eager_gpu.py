"""Informal comparison point (NOT the product, NOT the CPU baseline): the oracle's plain torch ops moved to the GPU, i.e. what
the reference's own code path (PyTorch eager: ATen + MIOpen + rocBLAS, fp32) does on this MI355X for the C2 training step.

    python tests/diag/torch_eager_gpu.py [batch] > gpurun_out/eager.log 2>&1      (prints progress lines as it goes: MIOpen's
    first-use kernel search / compilation can take minutes)"""
import os
import sys
import time

os.environ.setdefault("MIOPEN_FIND_MODE", "2")            # fast find: no exhaustive tuning
import torch

sys.path.insert(0, ".")
from oracle.step import OracleTrainer
from seghiero_amd.synthetic import make_batch


def log(msg):
    print("[%7.1f s] %s" % (time.perf_counter() - T0, msg), flush=True)


T0 = time.perf_counter()
torch.backends.cudnn.benchmark = False
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tr = OracleTrainer(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]])
for m in tr.modules().values():
    m.to(dev)
tr.hiera_loss_fn.to(dev)
tr.train()
img, lab = make_batch(B, 512, 9, seed=0, device=dev)
log("model on GPU, batch %d" % B)
hooks = []
for name, mod in list(tr.backbone.named_children()) + list(tr.aspp_head.named_children()):
    hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: log("first forward: %s done" % name)))
loss = tr.train_step(img, lab, 0)
torch.cuda.synchronize()
for h in hooks:
    h.remove()
log("first step done (includes MIOpen find), loss %.5f" % float(loss))
for k in range(3):
    tr.train_step(img, lab, 0)
    torch.cuda.synchronize()
    log("warm-up step %d" % k)
def timed(tag):
    for _ in range(2):
        tr.train_step(img_cur, lab, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        loss = tr.train_step(img_cur, lab, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    log("PyTorch eager fp32 %s: %.1f ms/step = %.1f images/s (loss %.5f)" % (tag, dt * 1e3, B / dt, float(loss)))


img_cur = img
timed("NCHW, MIOpen fast find")
torch.backends.cudnn.benchmark = True                     # MIOpen exhaustive find per new shape: the reference's best case
os.environ["MIOPEN_FIND_MODE"] = "1"
tr.train_step(img_cur, lab, 0)
torch.cuda.synchronize()
log("benchmark=True search step done")
timed("NCHW, cudnn.benchmark=True")
for m in tr.modules().values():
    m.to(memory_format=torch.channels_last)
img_cur = img.contiguous(memory_format=torch.channels_last)
tr.train_step(img_cur, lab, 0)
torch.cuda.synchronize()
log("channels_last search step done")
timed("channels_last, cudnn.benchmark=True")
