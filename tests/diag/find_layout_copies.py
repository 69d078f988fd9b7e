"""Diagnosis aid: where do layout conversions (sh_nchw_to_nhwc) and device-to-device torch copies happen in one training step?"""
import collections
import sys
import traceback
import torch
sys.path.insert(0, ".")
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer

sites = collections.Counter()
orig_call = ops._call


def spy(name, *a, **k):
    if name == "sh_nchw_to_nhwc":
        sites["sh_nchw_to_nhwc " + str(a[2:7]) + " @ " + traceback.format_stack(limit=3)[0].strip().splitlines()[0]] += 1
    return orig_call(name, *a, **k)


ops._call = spy
tr = SegHieroTrainer(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01, device="cuda:0")
img, lab = make_batch(2, 128, 9, seed=0, device="cuda:0")
lab8 = ops.labels_u8(lab)
tr.train_step(img, lab8, 0)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    tr.train_step(img, lab8, 0)
    torch.cuda.synchronize()
for ev in prof.key_averages(group_by_stack_n=6):
    if ev.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::mul", "aten::fill_", "aten::zero_", "aten::zeros", "aten::ones"):
        print("%-18s x%-4d %s" % (ev.key, ev.count, " <- ".join(s.split("/")[-1] for s in ev.stack[:4])))
for k, v in sites.items():
    print(v, k)
