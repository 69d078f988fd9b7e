"""Why does bench.py's instrumented step time sh_conv_wgrad_x6_lin 64->256 at 300 us when tools/shape_table.py sees 130 us?
Mirrors bench.py's setup; VARIANT env: seed (torch.manual_seed(0) before the trainer), devbatch (make_batch on the device)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer

variant = os.environ.get("VARIANT", "")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if "seed" in variant:
    torch.manual_seed(0)
tr = SegHieroTrainer(device=dev, depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01)
tr.train()
if "devbatch" in variant:
    img, lab = make_batch(16, 512, 9, seed=0, device=dev)
else:
    img, lab = make_batch(16, 512, 9, seed=0)
    img, lab = img.cuda(), lab.cuda()
lab = ops.labels_u8(lab)

orig = ops._call_fused
ptrs = []


def spy(name, *a, **kw):
    if name == "sh_conv_wgrad_x6_lin" and kw.get("key", "").startswith("16x128x128 64->256"):
        ptrs.append((a[0], a[4], a[6], a[9], a[10]))
    return orig(name, *a, **kw)


ops._call_fused = spy


def show(tag):
    ptrs.clear()
    with ops.profile() as prof:
        tr.train_step(img, lab, 0)
    torch.cuda.synchronize()
    tot = sum(v["ms"] for v in prof.rows.values())
    print(tag, f"total {tot:.2f} ms")
    for (name, key), r in prof.shapes.items():
        if "64->256" in str(key):
            print(f"   {name:26s} {str(key):34s} x{r['calls']:<2d} {r['ms']:7.3f} ms")
    for p in ptrs:
        print("   x %x  g %x  y %x  dw %x  ws %x" % p)
    print("   per call us:", [round(1e3 * e0.elapsed_time(e1)) for (name, cost, e0, e1, key) in prof.rec if name == "sh_conv_wgrad_x6_lin" and str(key).startswith("16x128x128 64->256")])


if "bench" in variant:
    for _ in range(3):
        tr.train_step(img, lab, 0)
    torch.cuda.synchronize()
    for _ in range(20):
        loss = tr.train_step(img, lab, 0)
    torch.cuda.synchronize()
    print(float(loss))
    show(variant + " 3 + 20 steps, first instrumented step")
    show(variant + " second")
    sys.exit(0)
for _ in range(3):
    tr.train_step(img, lab, 0)
torch.cuda.synchronize()
show(variant + " after 3 steps")
for _ in range(20):
    tr.train_step(img, lab, 0)
torch.cuda.synchronize()
show(variant + " after 20 more")
