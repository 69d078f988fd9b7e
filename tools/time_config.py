#!/usr/bin/env python3
"""Step time + per-kernel breakdown of one of the BASELINE configs on the GPU (not the bench metric; a diagnosis aid).
    python tools/time_config.py C4 [batch]      C2 | C4 (R101 3-level RMI, 7/3/2) | C5 (R101 2-level 20f/5c, 1024^2)"""
import os
import sys
import time
import torch
sys.path.insert(0, ".")
from seghiero_amd import ops
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer

CFGS = {
    "C2": dict(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], size=512, batch=16),
    "C4": dict(depth=101, n_fine=7, coarse_to_fine_map=[[0], [1, 4], [5, 6]], super_coarse_to_coarse_map=[[0], [1, 6]], size=512, batch=16),
    "C5": dict(depth=101, n_fine=20, coarse_to_fine_map=[[0, 3], [4, 7], [8, 11], [12, 15], [16, 19]], size=1024, batch=4),
}


def ab(name, batch):
    """fp32-accurate vs bf16 compute mode of one configuration in ONE process, alternating (the ratio without box-to-box and clock-ramp noise)."""
    cfg = dict(CFGS[name])
    size, b0 = cfg.pop("size"), cfg.pop("batch")
    batch = batch or b0
    img, lab = make_batch(batch, size, cfg["n_fine"], seed=0, device="cuda:0")
    lab8 = ops.labels_u8(lab)
    trs = {}
    for mode, dt in (("f32", torch.float32), ("b16", torch.bfloat16)):
        torch.manual_seed(0)
        trs[mode] = SegHieroTrainer(lr=0.01, device="cuda:0", compute_dtype=dt, **cfg)
        trs[mode].train()
        for _ in range(3):
            trs[mode].train_step(img, lab8, 0)
    res = {"f32": [], "b16": []}
    for rnd in range(4):
        for mode in ("f32", "b16"):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                trs[mode].train_step(img, lab8, 0)
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) / 20 * 1e3)
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    print(f"{name} batch {batch} @ {size}^2, 4 alternating rounds of 20 steps: fp32-accurate {[round(x, 2) for x in res['f32']]} ms, "
          f"bf16 compute {[round(x, 2) for x in res['b16']]} ms; medians {med['f32']:.2f} / {med['b16']:.2f} ms = {med['f32'] / med['b16']:.2f}x "
          f"({batch / med['f32'] * 1e3:.1f} -> {batch / med['b16'] * 1e3:.1f} images/s)")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    if len(sys.argv) > 3 and sys.argv[3] == "ab":
        return ab(name, int(sys.argv[2]))
    cfg = dict(CFGS[name])
    size, batch = cfg.pop("size"), cfg.pop("batch")
    if len(sys.argv) > 2:
        batch = int(sys.argv[2])
    if len(sys.argv) > 3 and sys.argv[3] == "bf16":          # trunk activations stored as bf16 (BASELINE configs[4])
        cfg["act_dtype"] = torch.bfloat16
    if len(sys.argv) > 3 and sys.argv[3] == "b16":           # bf16 COMPUTE mode: bf16 storage + one MFMA product per tile
        cfg["compute_dtype"] = torch.bfloat16
    torch.manual_seed(0)
    tr = SegHieroTrainer(lr=0.01, device="cuda:0", **cfg)
    tr.train()
    img, lab = make_batch(batch, size, cfg["n_fine"], seed=0, device="cuda:0")
    lab8 = ops.labels_u8(lab)
    for _ in range(3):
        tr.train_step(img, lab8, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        loss = tr.train_step(img, lab8, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: batch {batch} @ {size}^2: {dt * 1e3:.2f} ms/step = {batch / dt:.1f} images/s, loss {float(loss):.5f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    with ops.profile() as prof:
        tr.train_step(img, lab8, 0)
    torch.cuda.synchronize()
    tot = sum(v["ms"] for v in prof.rows.values())
    print(f"   one-stream kernel time {tot:.2f} ms in {sum(v['calls'] for v in prof.rows.values())} launches")
    for k, v in sorted(prof.rows.items(), key=lambda kv: -kv[1]["ms"])[:22]:
        rate = f"{v['flops'] / v['ms'] / 1e9:7.1f} TF  {v['bytes'] / v['ms'] / 1e6:7.0f} GB/s(alg)" if v["flops"] else ""
        print(f"   {k:28s} {v['ms']:8.2f} ms  {v['calls']:5d} calls  {rate}")
    if os.environ.get("SHAPES"):          # per (entry point, layer shape) rows whose entry-point name contains $SHAPES
        for (nm, key), r in sorted(prof.shapes.items(), key=lambda kv: -kv[1]["ms"]):
            if os.environ["SHAPES"] in nm or os.environ["SHAPES"] == "all":
                print(f"      {nm:28s} {str(key):40s} x{r['calls']:<2d} {r['ms']:7.3f} ms")


if __name__ == "__main__":
    main()
