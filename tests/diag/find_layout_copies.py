import sys, traceback, torch
sys.path.insert(0, ".")
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer
orig = ops._call
def spy(name, *a, **k):
    if name == "sh_nchw_to_nhwc":
        print("sh_nchw_to_nhwc n,c,h,w,cpad =", a[2:7])
        traceback.print_stack(limit=6)
    return orig(name, *a, **k)
ops._call = spy
tr = SegHieroTrainer(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], lr=0.01, device="cuda:0")
img, lab = make_batch(2, 128, 9, seed=0, device="cuda:0")
tr.train_step(img, ops.labels_u8(lab), 0)
torch.cuda.synchronize()
