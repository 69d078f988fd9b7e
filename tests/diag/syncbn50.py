"""Per-parameter gradient distance of the 2-rank SyncBN run from the single-rank full batch (ResNet-50 + head)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import test_ddp_gpu as T

if __name__ == "__main__":
    import torch.multiprocessing as mp
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    cuts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=T._worker, args=(r, 2, 29877, q, "gloo", cuts, depth)) for r in range(2)]
    for p in procs: p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs: p.join(60)
    for r in res.values():
        assert not isinstance(r, str), r
    full = T._run(0, T.B, sync=False, depth=depth)
    rows = sorted(((T._rel(res[0]["grads"][k] + res[1]["grads"][k], full["grads"][k]), k) for k in full["grads"]), reverse=True)
    for e, k in rows[:25]:
        print(f"{e:.3e} {k}")
    print("...", rows[len(rows) // 2])
    for e, k in rows[-25:]:
        print(f"{e:.3e} {k}")
