"""Diagnostic: per-tensor gradient distance from the fp64 oracle DDP statement for (a) two ranks with SyncBN, (b) the oracle in fp32.
Usage: python tests/diag/ddp_sharded.py [size] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import test_ddp_gpu as T

if __name__ == "__main__":
    T.SH_S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    T.SH_B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    SYNC = (sys.argv[3] if len(sys.argv) > 3 else "sync") == "sync"
    import torch.multiprocessing as mp
    from oracle.step import OracleTrainer
    from seghiero_amd.synthetic import make_batch
    cuts = [0, T.SH_B // 2, T.SH_B]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=T._sharded_worker, args=(r, 2, 29333, q, cuts, T.SH_S, T.SH_B, SYNC)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=300) for _ in procs)
    [p.join(60) for p in procs]
    for r in res.values():
        assert not isinstance(r, str), r
    img, lab = make_batch(T.SH_B, T.SH_S, 4, seed=5)
    init = {k: {n: torch.from_numpy(v) for n, v in sd.items()} for k, sd in res[0]["init"].items()}
    ref, ref64 = OracleTrainer(**T.TR_KW), OracleTrainer(**T.TR_KW)
    for k, m in ref.modules().items():
        m.load_state_dict(init[k]); ref64.modules()[k].load_state_dict(init[k]); ref64.modules()[k].double()
    ref.train(); ref64.train()
    if SYNC:
        m32, t32, _, _ = ref.ddp_forward_loss(img, lab, T.SH_EPOCH, cuts); m32.backward()
        m64, t64, _, _ = ref64.ddp_forward_loss(img.double(), lab, T.SH_EPOCH, cuts); m64.backward()
    else:                                   # per-rank BatchNorm statistics: every shard is its own forward; mean of the losses
        t32 = [ref.forward_loss(img[a:b], lab[a:b], T.SH_EPOCH)[0] for a, b in zip(cuts[:-1], cuts[1:])]; (sum(t32) / 2).backward()
        t64 = [ref64.forward_loss(img[a:b].double(), lab[a:b], T.SH_EPOCH)[0] for a, b in zip(cuts[:-1], cuts[1:])]; (sum(t64) / 2).backward()
    print("losses", [res[r]["loss"] for r in (0, 1)], [float(x) for x in t32], [float(x) for x in t64])
    g64 = {f"{mk}.{k}": p.grad.numpy() for mk, m in ref64.modules().items() for k, p in m.named_parameters()}
    g32 = {f"{mk}.{k}": p.grad.numpy() for mk, m in ref.modules().items() for k, p in m.named_parameters()}
    names = list(g64)
    cat = lambda d: np.concatenate([np.asarray(d[k], np.float64).ravel() for k in names])
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print("whole vector: e_m %.3e  e_r %.3e" % (rel(cat(res[0]["grads"]), cat(g64)), rel(cat(g32), cat(g64))))
    hd = [k for k in names if not k.startswith("backbone.")]
    cat = lambda d: np.concatenate([np.asarray(d[k], np.float64).ravel() for k in hd])
    print("heads only  : e_m %.3e  e_r %.3e" % (rel(cat(res[0]["grads"]), cat(g64)), rel(cat(g32), cat(g64))))
    worst = []
    for k, t in g64.items():
        sc = max(float(np.abs(t).max()), 1e-3)
        pc = lambda a: np.abs(a - t).reshape(t.shape[0], -1).max(1) / sc
        em, er = pc(res[0]["grads"][k]), pc(g32[k])
        bound = 3 * np.median(er) + 1e-4
        worst.append((np.median(em) / bound, k, float(np.median(em)), float(np.median(er)), int((em > 10 * bound).sum()), t.shape[0]))
    worst.sort(reverse=True)
    print("median-per-channel statistic, worst 8 (ratio to 3*median(e_r)+1e-4, tensor, med e_m, med e_r, channels beyond 10x, of):")
    for w in worst[:8]:
        print("   ", w)
    print("tensors with flipped channels:", [(w[1], w[4], w[5]) for w in worst if w[4]])
