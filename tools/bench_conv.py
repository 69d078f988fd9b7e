#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernels on the layer shapes of the ResNet-50 / 512x512 / B=16 step.
Times fprop / dgrad / wgrad with HIP events (torch current stream) and prints TFLOP/s (fp32 MFMA peak 157.3)."""
import os
import sys
import torch
sys.path.insert(0, ".")
from seghiero_amd import ops

DEV = "cuda:0"
SHAPES = [  # name, N, H, W, Cin, Cout, k, stride, pad
    ("stem7x7", 16, 512, 512, 4, 64, 7, 2, 3),
    ("l1.conv1 256>64", 16, 128, 128, 256, 64, 1, 1, 0),
    ("l1.conv2 3x3 64", 16, 128, 128, 64, 64, 3, 1, 1),
    ("l1.conv3 64>256", 16, 128, 128, 64, 256, 1, 1, 0),
    ("l2.conv2 3x3 128", 16, 64, 64, 128, 128, 3, 1, 1),
    ("l2.conv3 128>512", 16, 64, 64, 128, 512, 1, 1, 0),
    ("l2.0.conv2 3x3/2", 16, 128, 128, 128, 128, 3, 2, 1),
    ("l3.conv2 3x3 256", 16, 32, 32, 256, 256, 3, 1, 1),
    ("l3.conv3 256>1024", 16, 32, 32, 256, 1024, 1, 1, 0),
    ("l3.conv1 1024>256", 16, 32, 32, 1024, 256, 1, 1, 0),
    ("l4.conv2 3x3 512", 16, 16, 16, 512, 512, 3, 1, 1),
    ("l4.conv3 512>2048", 16, 16, 16, 512, 2048, 1, 1, 0),
    ("aspp.pw 2048>512", 16, 16, 16, 2048, 512, 1, 1, 0),
    ("proj 2048>2048", 16, 16, 16, 2048, 2048, 1, 1, 0),
    ("sep0.pw 560>512", 16, 128, 128, 560, 512, 1, 1, 0),
    ("sep1.pw 512>512", 16, 128, 128, 512, 512, 1, 1, 0),
]


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def phase_profile(x, wt, y, part, n, h, w, cin, cout, k, s, p):
    """SEGHIERO_X6_VARIANT=7 SEGHIERO_X6_PROF=1: run the instrumented fprop once and print the per-wave phase cycle totals
    (s_memtime ticks, averaged over the waves of the first 64 blocks)."""
    import numpy as np
    dbg = torch.zeros(1 << 20, dtype=torch.uint8, device=DEV)
    xp, ldx = ops.pm(x)
    yp, ldy = ops.pm(y)
    ops._call("sh_conv_fprop_x6", xp, ldx, ops.w_ohwi(wt).data_ptr(), None, yp, ldy, part.data_ptr(), n, h, w, cin, cout, k, k, s, p, 1,
              dbg.data_ptr(), dbg.numel(), 0, ops._st())
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy().view(np.uint64)
    total = raw[65536:65536 + 64 * 16].astype(np.float64)
    total = total[total > 0]
    d = raw[:65536].reshape(-1, 8)
    d = d[d[:, 5] > 0].astype(np.float64)
    if len(d) == 0:
        print("   (no instrumented instantiation for this shape)")
        return
    tot = d[:, 5].mean()
    names = ["lds-read+mfma", "barrier1", "vmcnt wait", "split+lds-write", "barrier2"]
    print(f"   waves sampled {len(d)}, k tiles {int(d[0, 6])}, loop cycles/wave {tot:.0f} = {tot / d[0, 6]:.0f} per k tile;  " +
          "  ".join(f"{nm} {d[:, i].mean() / tot * 100:.1f}%" for i, nm in enumerate(names)))
    if len(total):
        print(f"   whole block {total.mean():.0f} cycles: prologue {d[:, 7].mean():.0f}, loop {tot:.0f}, epilogue {total.mean() - tot - d[:, 7].mean():.0f}")


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    tot = {"fprop": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
    for name, n, h, w, cin, cout, k, s, p in SHAPES:
        if only and only not in name:
            continue
        ho, wo = ops.conv_out_hw(h, w, k, k, s, p, 1)
        x = ops.new_act(n, cin, h, w, DEV); x.normal_()
        wt = torch.randn(cout, cin, k, k, device=DEV).contiguous(memory_format=torch.channels_last) * 0.05
        y = ops.new_act(n, cout, ho, wo, DEV)
        part = ops.conv_partials(n * ho * wo, cout, DEV)
        dy = ops.new_act(n, cout, ho, wo, DEV); dy.normal_()
        dx = ops.new_act(n, cin, h, w, DEV)
        dw = torch.empty_like(wt)
        fl = 2.0 * n * ho * wo * cout * cin * k * k
        tf = timeit(lambda: ops.conv_fprop(x, wt, None, y, part, s, p, 1))
        if os.environ.get("SEGHIERO_X6_PROF") == "1":
            phase_profile(x, wt, y, part, n, h, w, cin, cout, k, s, p)
        td = timeit(lambda: ops.conv_dgrad(dy, wt, dx, s, p, 1))
        tw = timeit(lambda: ops.conv_wgrad(x, dy, dw, s, p, 1))
        extra = ""
        if os.environ.get("SEGHIERO_BENCH_FUSED") == "1" and cin % 4 == 0:
            # the fused BatchNorm hooks on the same shape: input read through BN+ReLU (fprop / wgrad), BN-backward front half in dgrad
            coefs = torch.rand(4, cin, device=DEV) + 0.5
            bp = torch.empty((-(-n * h * w // 64), 2, cin), device=DEV)
            tfa = timeit(lambda: ops.conv_fprop_aff(x, coefs, wt, None, y, part, s, p, 1)) if ops.conv_fprop_aff(x, coefs, wt, None, y, part, s, p, 1) else float("nan")
            twa = timeit(lambda: ops.conv_wgrad(x, dy, dw, s, p, 1, aff=coefs)) if ops.wgrad_aff_ok(x, dw, s, p, 1) else float("nan")
            tdb = timeit(lambda: ops.conv_dgrad_bnb(dy, wt, dx, x, coefs, True, bp, s, p, 1)) if ops.conv_dgrad_bnb(dy, wt, dx, x, coefs, True, bp, s, p, 1) else float("nan")
            extra = f" || fused: fprop_aff {tfa*1e3:7.1f}us dgrad_bnb {tdb*1e3:7.1f}us wgrad_aff {twa*1e3:7.1f}us"
        for key, t in (("fprop", tf), ("dgrad", td), ("wgrad", tw)):
            tot[key][0] += fl; tot[key][1] += t
        print(f"{name:20s} GF={fl/1e9:7.1f}  fprop {tf*1e3:7.1f}us {fl/tf/1e9:6.1f}TF | dgrad {td*1e3:7.1f}us {fl/td/1e9:6.1f}TF | "
              f"wgrad {tw*1e3:7.1f}us {fl/tw/1e9:6.1f}TF" + extra, flush=True)
    for key, (fl, t) in tot.items():
        if t:
            print(f"TOTAL {key}: {fl/t/1e9:.1f} TF")


if __name__ == "__main__":
    main()
