"""Can the whole training step (forward, loss, hand-scheduled backward with its weight-gradient side stream, fused SGD) be captured in one
hipGraph and replayed?  Prints eager vs replay time per step and checks that replays train exactly like eager steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer

b16 = len(sys.argv) > 1 and sys.argv[1] == "b16"
kw = dict(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01, device="cuda:0",
          compute_dtype=torch.bfloat16 if b16 else torch.float32)
torch.manual_seed(0)
tr = SegHieroTrainer(**kw)
tr.train()
img, lab = make_batch(16, 512, 9, seed=0, device="cuda:0")
lab8 = ops.labels_u8(lab)
for _ in range(3):
    tr.train_step(img, lab8, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    l = tr.train_step(img, lab8, 0)
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step, loss {float(l):.5f}")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        tr.train_step(img, lab8, 0)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        loss = tr.train_step(img, lab8, 0)
except Exception as e:
    import traceback
    traceback.print_exc()
    print("CAPTURE FAILED:", repr(e)[:400])
    sys.exit(0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms/step, loss {float(loss):.5f}")
