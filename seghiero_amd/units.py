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


def measure(device="cuda:0", batch=16, cin=2048, hw=16, channels=512, dilations=(1, 12, 24, 36), iters=30):
    dev = torch.device(device)
    torch.manual_seed(0)
    aspp = H.DepthwiseSeparableASPPModule(dilations=dilations, in_channels=cin, channels=channels).to(dev).train()
    c4 = ops.new_act(batch, cin, hw, hw, dev)
    c4.normal_().relu_()
    nb = len(dilations)
    cat = ops.new_act(batch, channels * (nb + 1), hw, hw, dev)

    def unit(grouped):
        keep = H.ASPP_GROUPED
        H.ASPP_GROUPED = grouped
        try:
            R = {}
            if not H._aspp_branches_grouped(aspp, c4, cat, channels, True, R):
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
    us = out["grouped"] / nb
    return {"unit": "ASPP depthwise-separable branch forward (dw3x3 -> BN -> ReLU -> pw 2048->512 -> BN), c4 = [%d,%d,%d,%d]" % (batch, cin, hw, hw),
            "us": round(us, 1), "us_all_%d_gemm_branches_one_unit" % nb: round(out["grouped"], 1),
            "us_per_branch_launched_one_by_one": round(out["branch_by_branch"] / nb, 1),
            "kernels_us": out["kernels_us"], "bound_hbm_us": round(HBM_US, 1), "bound_mfma_us": round(MFMA_US, 1),
            "frac_hbm": round(HBM_US / us, 3), "frac_mfma": round(MFMA_US / us, 3),
            "note": "MFMA-bound at fp32 accuracy (8.74 GF per 46.2 MB); per-branch time = one quarter of the grouped unit (1x1 branch + 3 DS "
                    "branches: identical pointwise work each)"}
