#!/usr/bin/env python3
"""bench.py -- images/sec of the SegHiero training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic batch resident in HBM: ResNet-50 trunk -> DS-ASPP contrast head
-> fused resize + 2-level HieraTripletLoss (+ aux head + CE) -> backward -> (RCCL gradient all-reduce) -> fused SGD.
Workload = BASELINE.json configs[1]: 512x512, batch 16 per GPU (weak scaling), 9 fine / 4 coarse classes, fp32.
Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the launch stream over one extra
instrumented step; `cpu_baseline` times the CPU oracle (oracle.step.OracleTrainer, kind "port") on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TF = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TF = 2500.0       # dense bf16 MFMA peak; the x6 conv path spends 6 bf16 MFMA flops per fp32 flop
HBM_PEAK_GBS = 8000.0
CFG = dict(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], size=512, batch=16)


def csrc_sha():
    """hash of the kernel sources (profiles/*_hbm_traffic.json records the one it was measured with)"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "seghiero_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "seghiero_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def make_trainer(dev):
    """The trainer exactly as the timed run builds it (tests/test_model_gpu.py pins its step-0 loss against the CPU oracle)."""
    from seghiero_amd.train_step import SegHieroTrainer
    torch.manual_seed(0)
    return SegHieroTrainer(depth=CFG["depth"], n_fine=CFG["n_fine"], coarse_to_fine_map=CFG["coarse_to_fine_map"], lr=0.01, device=dev)


def make_inputs(batch, rank, dev):
    """The synthetic batch of rank `rank`: seed = rank, resident on `dev`."""
    from seghiero_amd.synthetic import make_batch
    return make_batch(batch, CFG["size"], CFG["n_fine"], seed=rank, device=dev)


def cpu_baseline(seconds_budget=30.0):
    """Reference CPU train loop (the oracle) at the headline shape, bounded: batch 2 at 512x512, ResNet-50."""
    from oracle.step import OracleTrainer
    from seghiero_amd.synthetic import make_batch
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    tr = OracleTrainer(depth=CFG["depth"], n_fine=CFG["n_fine"], coarse_to_fine_map=CFG["coarse_to_fine_map"])
    tr.train()
    b = 2
    img, lab = make_batch(b, CFG["size"], CFG["n_fine"], seed=0)
    t0 = time.perf_counter()
    tr.train_step(img, lab, 0)                       # warm-up (allocator, thread pool)
    warm = time.perf_counter() - t0
    steps = 1 if warm > seconds_budget / 2 else max(1, min(3, int(seconds_budget / max(warm, 1e-3)) - 1))
    t0 = time.perf_counter()
    for s in range(steps):
        tr.train_step(img, lab, 0)
    dt = time.perf_counter() - t0
    return {"value": round(b * steps / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"oracle.step.OracleTrainer (pure torch CPU restatement of train.py:260-320), ResNet-50 2-level, "
                      f"512x512, batch {b}, {steps} timed step(s) after 1 warm-up, {threads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)        # ~5 s timed region: a sustained-clock number
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-units", action="store_true", help="skip the separate north-star unit measurement (profiling runs)")
    ap.add_argument("--batch", type=int, default=CFG["batch"])
    ap.add_argument("--exact", action="store_true",
                    help="N > 1: exact data-parallel mode (ddp.EXACT: all-reduced loss normalisers, summed gradients; with --syncbn a sharded "
                         "step equals the single-process step on the global batch)")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary bf16-compute-mode measurement of the same workload")
    ap.add_argument("--syncbn", action="store_true",
                    help="N > 1: BatchNorm statistics over all ranks (BASELINE configs[2] variant; default = per-rank statistics, "
                         "the weak-scaling setting of SURVEY 8d)")
    args = ap.parse_args()

    from seghiero_amd import ddp, ops

    # SEGHIERO_BENCH_BACKEND=gloo + SEGHIERO_BENCH_ONE_DEVICE=1 rehearse the N>1 code path on a single-GPU box
    rank, local, world = ddp.init_from_env(backend=os.environ.get("SEGHIERO_BENCH_BACKEND"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("SEGHIERO_BENCH_ONE_DEVICE"):
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    tr = make_trainer(dev)
    if world > 1:
        ddp.broadcast_module_state(list(tr.modules().values()))
        tr.grad_sync = ddp.GradSync(tr.params)
        ops.SYNC_BN = bool(args.syncbn)
        ddp.EXACT = bool(args.exact)
    tr.train()
    img, lab = make_inputs(args.batch, rank, dev)
    lab8 = ops.labels_u8(lab)                       # the loader contract is i64 labels; convert once, outside the loop

    trace = (lambda m: print(f"[rank {rank}] {m}", file=sys.stderr, flush=True)) if os.environ.get("SEGHIERO_TRACE") else (lambda m: None)
    trace("setup done")

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    loss0 = None
    for _ in range(args.warmup):
        l = tr.train_step(img, lab8, 0)
        loss0 = l if loss0 is None else loss0       # the very first step from the seeded initial weights (read back after the loop)
        trace("warmup step done")
    barrier()
    trace("barrier passed")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.train_step(img, lab8, 0)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        ddp.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss_val = float(loss)
    # host side of a step: Python + launch time with the device idle at the start (nothing can block on a full queue)
    host_ms = []
    for _ in range(3):
        barrier()
        h0 = time.perf_counter()
        tr.train_step(img, lab8, 0)
        host_ms.append(1e3 * (time.perf_counter() - h0))
    barrier()

    # ---- roofline of the dominant kernel, measured live (HIP events on the launch stream), one instrumented step.
    # Every rank runs it (the step contains collectives); only rank 0 instruments and reports.
    # Two such steps: inside the context the weight gradients run on the launch stream (co-running kernels would inflate each other's
    # duration), and the first single-stream step allocates that stream's split-K workspaces (first launches measured 2-6x long).
    if rank != 0:
        for _ in range(2):
            tr.train_step(img, lab8, 0)
            barrier()
        return
    for _ in range(2):
        with ops.profile() as prof:
            tr.train_step(img, lab8, 0)
        barrier()
    rows, shapes = prof.rows, prof.shapes
    total_ms = sum(v["ms"] for v in rows.values())

    def price(name, r):
        """achieved rate of one kernel (family or single shape) against the roofline that bounds it"""
        x6 = "_x6" in name
        if r["flops"] > 0:
            peak = BF16_MFMA_PEAK_TF / 6.0 if x6 else FP32_MFMA_PEAK_TF
            ach = r["flops"] / (r["ms"] * 1e-3) / 1e12
            return {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
        return {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None}

    dom = max(rows, key=lambda k: rows[k]["ms"])
    r = rows[dom]
    roof = {"kernel": dom, **price(dom, r),
            "peak_note": ("fp32-equivalent peak of the 6x bf16-split MFMA path = 2500 TF dense bf16 / 6 products; "
                          "the plain f32 MFMA peak is 157.3 TF") if "_x6" in dom else "dense f32 MFMA peak",
            "traffic": None, "launches": r["calls"], "avg_launch_us": round(1e3 * r["ms"] / r["calls"], 1),
            "alg_bytes_per_step": r["bytes"], "alg_flops_per_step": r["flops"],
            "share_of_step": round(r["ms"] / total_ms, 3), "scope": "kernel family: all launches of this C-ABI entry point in one step"}
    # the dominant SINGLE kernel: one entry point on one layer shape (several launches only where layers repeat the shape)
    if shapes:
        (sname, skey), sr = max(shapes.items(), key=lambda kv: kv[1]["ms"])
        roof["single"] = {"kernel": sname, "shape": skey, **price(sname, sr), "launches": sr["calls"],
                          "avg_launch_us": round(1e3 * sr["ms"] / sr["calls"], 1), "share_of_step": round(sr["ms"] / total_ms, 3)}
    # HBM traffic of the same kernel family from the committed PMC passes (tools/profile_bench.sh -> tools/summarize_profile.py):
    # rocprofv3 cannot wrap this process from the inside, so the number is read from profiles/ and labelled with its source; the
    # profile records the hash of csrc/ it was taken with, and a number from other kernel sources is flagged stale.
    fam = {"sh_conv_fprop_x6": "conv_fprop_x6", "sh_conv_dgrad_x6": "conv_dgrad_x6", "sh_conv_wgrad_x6": "conv_wgrad_x6",
           "sh_conv_fprop_x6_aff": "conv_fprop_x6", "sh_conv_dgrad_x6_bnb": "conv_dgrad_x6", "sh_conv_wgrad_x6_aff": "conv_wgrad_x6",
           "sh_conv_dgrad_x6_lin": "conv_dgrad_x6", "sh_conv_wgrad_x6_lin": "conv_wgrad_x6",
           "sh_conv_fprop": "conv_fprop_f32", "sh_conv_dgrad": "conv_dgrad_f32", "sh_conv_wgrad": "conv_wgrad_f32"}.get(dom)
    try:
        import glob
        tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))[-1]
        tjs = json.load(open(tf))
        tj = tjs["per_kernel_family"].get(fam)
        if tj:
            roof["traffic"] = round(tj["hbm_bytes_per_launch"])
            roof["traffic_source"] = os.path.relpath(tf, ROOT) + " (PMC FETCH_SIZE x2 + WRITE_SIZE, per launch of the family)"
            roof["scope"] += "; traffic read from the committed rocprofv3 PMC passes of the same command, not measured in this run"
            roof["alg_bytes_per_launch"] = round(r["bytes"] / r["calls"])
            roof["traffic_stale"] = tjs.get("csrc_sha") != csrc_sha()
    except Exception:
        pass
    # the north-star unit (BASELINE.json): ASPP depthwise-separable branch forward, timed on its own
    roof_units = None
    ops.SYNC_BN = False          # from here on only rank 0 is running: no collective may be issued (the other ranks have returned)
    ddp.EXACT = False
    if not args.no_units:
        try:
            from seghiero_amd import units
            roof_units = {"aspp_ds_branch": units.measure(device=dev, batch=args.batch),
                          "aspp_ds_branch_bf16": units.measure(device=dev, batch=args.batch, bf16=True)}
        except Exception as e:                  # never lose the headline line to the side measurement
            roof_units = {"aspp_ds_branch": {"error": repr(e)}}
    # the same workload in bf16 COMPUTE mode (BASELINE configs[4]'s arithmetic: operands rounded once to bf16, one MFMA product, fp32
    # accumulate / statistics / weights) -- a secondary figure; the headline `value` above is the fp32-accurate path
    bf16_mode = None
    if world == 1 and not args.no_bf16:
        try:
            from seghiero_amd.train_step import SegHieroTrainer
            torch.manual_seed(0)
            tb = SegHieroTrainer(depth=CFG["depth"], n_fine=CFG["n_fine"], coarse_to_fine_map=CFG["coarse_to_fine_map"], lr=0.01, device=dev,
                                 compute_dtype=torch.bfloat16)
            tb.train()
            l0 = float(tb.train_step(img, lab8, 0))
            for _ in range(4):
                tb.train_step(img, lab8, 0)
            barrier()
            b0 = time.perf_counter()
            for _ in range(40):
                tb.train_step(img, lab8, 0)
            barrier()
            bdt = (time.perf_counter() - b0) / 40
            bf16_mode = {"images_per_s": round(args.batch / bdt, 1), "ms_per_step": round(1e3 * bdt, 2), "loss_step0": round(l0, 5),
                         "dtype": "bf16 operands (rounded once in the loaders), one MFMA product per tile, f32 accumulate / BatchNorm statistics / "
                                  "weights / SGD; bf16-stored activations and activation gradients", "steps": 40}
            del tb
        except Exception as e:
            bf16_mode = {"error": repr(e)}
    breakdown = {k: round(v["ms"], 2) for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:16]}
    out = {
        "metric": "images/sec at 512x512 (ResNet-50 2-level), full train step", "value": round(args.batch * world * args.steps / dt, 2),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (convolutions: exact 3-way bf16 split of every fp32 operand, 6 bf16 MFMA products, f32 accumulate; rest plain f32)",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: ResNet-50 + DepthwiseSeparableASPPContrastHead + 2-level HieraTripletLoss "
                               "+ aux head, 9 fine / 4 coarse, 512x512 synthetic, fwd+loss+bwd+SGD",
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                   "parallelism": f"dp{world}" + ("" if world == 1 else (" + SyncBN" if args.syncbn else " (per-rank BatchNorm statistics)")) +
                                  (" + exact normalisers" if args.exact and world > 1 else "")},
        "loss": round(loss_val, 5), "loss_step0": None if loss0 is None else round(float(loss0), 6),
        "host_ms_per_step": round(sorted(host_ms)[1], 2), "bf16_compute_mode": bf16_mode, "roofline": roof, "roofline_units": roof_units, "kernel_ms_per_step": breakdown,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
