"""Host-side cost of one training step (diagnosis): enqueue time without synchronisation vs the synchronised step time, and a cProfile of the
enqueue path.   python tests/diag/host_profile.py C5 4 b16"""
import cProfile, io, pstats, sys, time
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from time_config import CFGS
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer

name, batch, mode = sys.argv[1], int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "f32")
cfg = dict(CFGS[name]); size = cfg.pop("size"); cfg.pop("batch")
if mode == "b16":
    cfg["compute_dtype"] = torch.bfloat16
torch.manual_seed(0)
tr = SegHieroTrainer(lr=0.01, device="cuda:0", **cfg); tr.train()
img, lab = make_batch(batch, size, cfg["n_fine"], seed=0, device="cuda:0")
lab8 = ops.labels_u8(lab)
for _ in range(5):
    tr.train_step(img, lab8, 0)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.train_step(img, lab8, 0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{name} {mode} batch {batch}: enqueue {(t1 - t0) / n * 1e3:.2f} ms/step, with the final sync {(t2 - t0) / n * 1e3:.2f} ms/step")
# host-only cost: every step synchronised first, so the enqueue never waits on a full queue
hs = []
for _ in range(10):
    torch.cuda.synchronize()
    a = time.perf_counter(); tr.train_step(img, lab8, 0); hs.append(time.perf_counter() - a)
print(f"   enqueue on an idle queue: {sorted(hs)[len(hs) // 2] * 1e3:.2f} ms/step (median of 10)")
pr = cProfile.Profile()
# the backward functions run on autograd's own thread: profile them there, into a second profile
from seghiero_amd import backbone, head, loss, train_step
prb = cProfile.Profile()
for cls in (backbone._BackboneFn, head._HeadFn, head._AuxFn, loss._Hiera2Fn, loss._Hiera3Fn, loss._TripletFn, loss._CEAllPixFn, train_step._AuxCEFn):
    inner = cls.backward
    def wrapped(ctx, *g, _inner=inner):
        prb.enable()
        try:
            return _inner(ctx, *g)
        finally:
            prb.disable()
    cls.backward = staticmethod(wrapped)
torch.cuda.synchronize()
pr.enable()
for _ in range(5):
    tr.train_step(img, lab8, 0)
pr.disable()
torch.cuda.synchronize()
for title, prof in (("forward thread", pr), ("backward functions", prb)):
    s = io.StringIO()
    pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(30)
    print("=====", title); print(s.getvalue()[:7000])
    s = io.StringIO()
    pstats.Stats(prof, stream=s).sort_stats("cumtime").print_stats(25)
    print("===== (cumulative)", title); print(s.getvalue()[:6000])
