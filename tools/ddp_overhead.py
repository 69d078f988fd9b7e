#!/usr/bin/env python3
"""Cost of the data-parallel machinery itself on ONE GPU: the bench step with every collective of the N > 1 step issued for real at world
size 1 (backend "nccl" = RCCL, ddp.FORCE_COLLECTIVES: bucketed gradient all-reduce from the arena views on the side stream, the
class_count MIN-reduce; with --syncbn also the 2 x 69 BatchNorm statistics all-reduces) against the plain single-process step.
    python tools/ddp_overhead.py [--syncbn]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29377", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch
import torch.distributed as dist
import bench
from seghiero_amd import ddp, ops

syncbn = "--syncbn" in sys.argv
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def run(tr, img, lab8, n=40):
    for _ in range(5):
        tr.train_step(img, lab8, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(img, lab8, 0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


img, lab = bench.make_inputs(16, 0, dev)
lab8 = ops.labels_u8(lab)
plain = run(bench.make_trainer(dev), img, lab8)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
ddp.FORCE_COLLECTIVES = True
tr = bench.make_trainer(dev)
ddp.broadcast_module_state(list(tr.modules().values()))
tr.grad_sync = ddp.GradSync(tr.params)
ops.SYNC_BN = syncbn
forced = run(tr, img, lab8)
print(f"plain step {plain:.2f} ms; with every collective of the N > 1 step issued at world 1 (RCCL{', SyncBN' if syncbn else ''}) {forced:.2f} ms: "
      f"+{forced - plain:.2f} ms = {100 * (forced / plain - 1):.1f} %  ({len(tr.grad_sync.buckets)} gradient buckets of <= 32 MB)")
ddp.shutdown()
