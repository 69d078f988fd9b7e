"""Every bf16-compute forward conv of the head, checked call by call against the fp32-accurate kernel on the same (bf16) inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from seghiero_amd import ops
from seghiero_amd.head import DepthwiseSeparableASPPContrastHead

DEV = "cuda:0"
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
orig_f, orig_g = ops._fprop_b16, ops.conv1x1_grouped_fprop


def fprop_checked(x, in_coefs, weight, bias, y, partials, stride, pad, dil):
    ok = orig_f(x, in_coefs, weight, bias, y, partials, stride, pad, dil)
    if ok:
        keep, ops._B16 = ops._B16, False
        y2 = torch.empty_like(y.float()) if True else None
        y2 = ops.new_act(*y.shape[:1], y.shape[1], y.shape[2], y.shape[3], y.device, ld=ops.pm(y.float())[1] if False else ops.pad4(y.shape[1]), zero=True)
        p2 = None if partials is None else torch.empty_like(partials)
        if in_coefs is None:
            ops.conv_fprop(x, weight, bias, y2, p2, stride, pad, dil)
        else:
            assert ops.conv_fprop_aff(x, in_coefs, weight, bias, y2, p2, stride, pad, dil)
        ops._B16 = keep
        print(f"fprop_b16 x{tuple(x.shape)} ld{ops.pmx(x)[1]} -> {weight.shape[0]} k{weight.shape[2]} aff={in_coefs is not None} y:{y.dtype} rel {rel(y.float(), y2):.2e}"
              + ("" if partials is None else f"  partials rel {rel(partials, p2):.2e}"))
    return ok


def grouped_checked(sources, weights, y, partials):
    ok = orig_g(sources, weights, y, partials)
    if ok and ops._B16:
        keep, ops._B16 = ops._B16, False
        y2, p2 = torch.empty_like(y), torch.empty_like(partials)
        assert orig_g([(x.float(), c) for x, c in sources], weights, y2, p2)
        ops._B16 = keep
        a = weights[0].shape[0]
        print("grouped_b16:", [f"group {g} rel {rel(y[:, g * a:(g + 1) * a], y2[:, g * a:(g + 1) * a]):.2e}" for g in range(len(sources))], f"partials rel {rel(partials, p2):.2e}")
    return ok


ops._fprop_b16, ops.conv1x1_grouped_fprop = fprop_checked, grouped_checked
torch.manual_seed(4)
kw = dict(in_channels=256, c1_in_channels=64, c1_channels=48, aspp_channels=128, dilations=(1, 12, 24, 36), num_classes=13, proj_dim=64, proj_type="convmlp")
b = DepthwiseSeparableASPPContrastHead(**kw).to(DEV).train()
b.act_dtype = b.compute_dtype = torch.bfloat16
g = torch.Generator().manual_seed(9)
c1 = torch.randn(2, 64, 64, 64, generator=g).relu().bfloat16().float().to(DEV)
c4 = torch.randn(2, 256, 16, 16, generator=g).relu().bfloat16().float().to(DEV)
with torch.no_grad():
    lo, em = b([c1, None, None, c4])
print("done", lo.shape, em.shape)
