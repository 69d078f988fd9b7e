"""Measurement of the north-star unit (BASELINE.json): the ASPP depthwise-separable branch forward at B=16, 512^2 input, i.e.
c4 = [16, 2048, 16, 16] -> depthwise 3x3 (dilation d) -> BN -> ReLU -> pointwise 2048->512 -> BN (train-mode statistics), reference
``models/head/sep_aspp_contrast_head.py:55-62, 125-131``.  Algorithmic cost per branch: 46.2 MB, 8.74 GF => 5.8 us at
8 TB/s, 21 us at the 416.7 TF fp32-equivalent peak of the 6-product bf16 MFMA path: MFMA-bound by 3.6x at fp32 accuracy.

The product runs the four GEMM branches of the ASPP (1x1 branch + three DS branches) as ONE unit (``head._aspp_branches_grouped``):
shared statistics of c4 for the centre-tap branches, one depthwise kernel for the dilation that still touches the image, one
grouped pointwise launch, per-branch finalize, one BatchNorm+ReLU pass into the concat buffer.  ``measure()`` times exactly that
call with HIP events on the launch stream and reports the per-branch share next to both bounds.
"""
import torch

from . import head as H
from . import ops

HBM_US, MFMA_US = 46.2e6 / 8.0e12 * 1e6, 8.74e9 / (2500e12 / 6.0) * 1e6


HBM_US_BF16, MFMA_US_BF16 = 23.1e6 / 8.0e12 * 1e6, 8.74e9 / 2500e12 * 1e6      # bf16 storage, one MFMA product


def measure(device="cuda:0", batch=16, cin=2048, hw=16, channels=512, dilations=(1, 12, 24, 36), iters=30, bf16=False):
    """bf16=True: the same unit in bf16 compute mode (csrc/conv_b16.hip: c4 and the depthwise output read as bf16 tensors, operands rounded
    once to bf16, one MFMA product per tile) against the bf16 bounds -- 23.1 MB => 2.9 us at 8 TB/s, 3.5 us at 2.5 PF."""
    if bf16:
        with ops.compute_as(torch.bfloat16):
            return _measure(device, batch, cin, hw, channels, dilations, iters, True)
    return _measure(device, batch, cin, hw, channels, dilations, iters, False)


def _measure(device, batch, cin, hw, channels, dilations, iters, bf16):
    dev = torch.device(device)
    torch.manual_seed(0)
    aspp = H.DepthwiseSeparableASPPModule(dilations=dilations, in_channels=cin, channels=channels).to(dev).train()
    c4 = ops.new_act(batch, cin, hw, hw, dev)
    c4.normal_().relu_()
    c4b = c4.to(torch.bfloat16) if bf16 else None
    nb = len(dilations)
    cat = ops.new_act(batch, channels * (nb + 1), hw, hw, dev, dtype=torch.bfloat16 if bf16 else torch.float32)
    hbm_us, mfma_us = (HBM_US_BF16, MFMA_US_BF16) if bf16 else (HBM_US, MFMA_US)

    if bf16:          # the bf16 weight copies are made once per training step for the whole model (ops.prepare_bf16_weights), not per unit
        wcache = {}
        ops.prepare_bf16_weights([aspp.branches[0][0].weight] + [aspp.branches[i][0].pointwise.weight for i in range(1, nb)], wcache)

    def unit(grouped):
        keep = H.ASPP_GROUPED
        H.ASPP_GROUPED = grouped
        try:
            R = {}
            if not H._aspp_branches_grouped(aspp, c4, cat, channels, True, R, c4b):
                from . import layers as L
                L.cba_fwd(c4, aspp.branches[0][0].weight, H.G1, aspp.branches[0][1], True, True, out=cat[:, channels:2 * channels])
                for i, d in enumerate(dilations[1:], start=1):
                    ds = aspp.branches[i][0]
                    t, _ = L.dw_fwd(c4, ds.depthwise.weight, d, ds.bn_dw, True, lazy=True)
                    L.cba_fwd(t, ds.pointwise.weight, H.G1, ds.bn_pw, True, True, out=cat[:, (i + 1) * channels:(i + 2) * channels])
        finally:
            H.ASPP_GROUPED = keep

    out = {}
    for name, grouped in (("grouped", True), ("branch_by_branch", False)):
        for _ in range(3):
            unit(grouped)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            unit(grouped)
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) * 1e3 / iters
    with ops.profile() as prof:
        unit(True)
    out["kernels_us"] = {k.replace("sh_", ""): round(v["ms"] * 1e3, 1) for k, v in prof.rows.items()}
    if bf16:
        ops.release_dgrad_weights()
    us = out["grouped"] / nb
    # the branches are not alike: at 16 x 16 only the dilation-12 branch runs a real depthwise kernel (the others are centre-tap scalings of
    # c4 folded into BatchNorm coefficients).  Per kind, from the kernel times of the unit: the shared launches (grouped GEMM, its
    # finalize, the BatchNorm + ReLU pass) split by nb; the depthwise kernel + its finalize to the real branch; the c4 statistics pass +
    # its finalize over the centre-tap branches
    kk = {k: v["ms"] * 1e3 for k, v in prof.rows.items()}
    gemm = kk.get("sh_conv1x1_grouped_fprop_b16", 0.0) + kk.get("sh_conv1x1_grouped_fprop_x6", 0.0)
    shared = (gemm + kk.get("sh_bn_act", 0.0)) / nb
    n_centre = sum(1 for d in dilations[1:] if d >= hw)
    fin = kk.get("sh_bn_finalize_multi", 0.0) + kk.get("sh_bn_finalize", 0.0)
    real = shared + kk.get("sh_dwconv_fprop", 0.0) + fin / 3
    centre = shared + (kk.get("sh_channel_stats", 0.0) + fin / 3) / max(n_centre, 1)
    return {"unit": "ASPP depthwise-separable branch forward (dw3x3 -> BN -> ReLU -> pw 2048->512 -> BN), c4 = [%d,%d,%d,%d]%s" % (
                batch, cin, hw, hw, ", bf16 compute mode" if bf16 else ""),
            "us": round(us, 1), "us_all_%d_gemm_branches_one_unit" % nb: round(out["grouped"], 1),
            "us_per_branch_launched_one_by_one": round(out["branch_by_branch"] / nb, 1),
            "us_real_depthwise_branch_kernel_time": round(real, 1), "us_centre_tap_branch_kernel_time": round(centre, 1),
            "kernels_us": out["kernels_us"], "bound_hbm_us": round(hbm_us, 1), "bound_mfma_us": round(mfma_us, 1),
            "frac_hbm": round(hbm_us / us, 3), "frac_mfma": round(mfma_us / us, 3),
            "frac_hbm_real_depthwise_branch": round(hbm_us / real, 3) if real else None,
            "note": ("bf16 storage, one MFMA product: HBM- and MFMA-bound within 20 % of each other (8.74 GF per 23.1 MB)" if bf16 else
                     "MFMA-bound at fp32 accuracy (8.74 GF per 46.2 MB)") +
                    "; `us` = one quarter of the grouped unit (1x1 branch + 3 DS branches: identical pointwise work each); the real-depthwise "
                    "and centre-tap figures are sums of kernel times, without the launch gaps `us` includes"}
