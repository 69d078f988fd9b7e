#!/usr/bin/env python3
"""Micro-benchmark of the bf16 COMPUTE mode conv kernels (csrc/conv_b16.hip) on the layer shapes of the step: fprop (as is / through
the BatchNorm + ReLU loader), dgrad (bf16 gradient, plain / BatchNorm-backward epilogue), wgrad (plain / loader).  Prints microseconds,
TFLOP/s against the 2.5 PF dense bf16 peak and algorithmic GB/s against the 8 TB/s HBM peak."""
import sys
import torch
sys.path.insert(0, ".")
from seghiero_amd import ops
from tools.bench_conv import SHAPES, timeit

DEV = "cuda:0"


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    grad32 = len(sys.argv) > 2 and sys.argv[2] == "g32"
    gdt = torch.float32 if grad32 else torch.bfloat16
    with ops.compute_as(torch.bfloat16):
        for name, n, h, w, cin, cout, k, s, p in SHAPES:
            if (only and only not in name) or cin % 8 or s != 1:
                continue
            ho, wo = ops.conv_out_hw(h, w, k, k, s, p, 1)
            m = n * ho * wo
            x = ops.new_act(n, cin, h, w, DEV, dtype=torch.bfloat16); x.normal_()
            wt = torch.randn(cout, cin, k, k, device=DEV).contiguous(memory_format=torch.channels_last) * 0.05
            y = ops.new_act(n, cout, ho, wo, DEV, dtype=torch.bfloat16)
            part = ops.conv_partials(m, cout, DEV)
            dy = ops.new_act(n, cout, ho, wo, DEV, ld=ops.pad8(cout), dtype=gdt); dy.normal_()
            dx = ops.new_act(n, cin, h, w, DEV, dtype=gdt)
            dw = torch.empty_like(wt)
            coefs = torch.rand(4, cin, device=DEV) + 0.5
            bp = torch.empty((-(-n * h * w // 64), 2, cin), device=DEV)
            ops.weights_bf16(wt)
            cache = {}
            ops.prepare_bf16_weights([wt], cache)
            fl = 2.0 * m * cout * cin * k * k
            r = {}
            r["fprop"] = timeit(lambda: ops._fprop_b16(x, None, wt, None, y, part, s, p, 1))
            r["fprop_aff"] = timeit(lambda: ops._fprop_b16(x, coefs, wt, None, y, part, s, p, 1))
            r["dgrad"] = timeit(lambda: ops._dgrad_b16(dy, wt, dx, s, p, 1))
            r["dgrad_bnb"] = timeit(lambda: ops._dgrad_b16(dy, wt, dx, s, p, 1, bnb=(x, coefs, True, bp, None)))
            r["wgrad"] = timeit(lambda: ops._wgrad_b16(x, dy, dw, s, p, 1, False, None)) if cout % 8 == 0 else float("nan")
            r["wgrad_aff"] = timeit(lambda: ops._wgrad_b16(x, dy, dw, s, p, 1, False, coefs)) if cout % 8 == 0 else float("nan")
            gb = 2 if not grad32 else 4
            byt = {"fprop": 2 * n * h * w * cin + 2 * m * cout, "dgrad": gb * (m * cout + n * h * w * cin), "wgrad": 2 * n * h * w * cin + gb * m * cout}
            ops.release_dgrad_weights()
            print(f"{name:20s} GF={fl/1e9:7.1f} " + " | ".join(
                f"{key} {t*1e3:6.1f}us {fl/t/1e9:6.0f}TF {byt[key.split('_')[0]] * (1.5 if key == 'dgrad_bnb' else 1) / t / 1e6:5.0f}GB/s" for key, t in r.items()), flush=True)


if __name__ == "__main__":
    main()
