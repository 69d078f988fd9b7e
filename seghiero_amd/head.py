"""DepthwiseSeparableASPPContrastHead -- drop-in for reference ``models/head/sep_aspp_contrast_head.py:6-254``.

Same class names (``ProjectionHead``, ``DepthwiseSeparableConv``, ``DepthwiseSeparableASPPModule``,
``DepthwiseSeparableASPPContrastHead``), constructor signatures, attribute names and hence the same 94 state_dict
entries, same ``forward(inputs: list of 4) -> (logits, embedding)`` and the same ``ValueError`` on an unknown
``proj_type``.  The torch.nn modules hold parameters only; ``_HeadFn`` runs the HIP kernels:

* the five ASPP branches write their BN+ReLU output straight into channel slices of one 5*A-channel buffer (the
  ``torch.cat`` of ``:113`` is virtual), the x8 bilinear upsample and the C1 skip write into slices of the
  (A+48)-channel decoder buffer (``:235-240``);
* every conv emits its BatchNorm (sum, sum^2) partials from its epilogue, so no tensor is re-read for statistics;
* backward is hand-scheduled: the six gradients flowing into C4 are summed inside dgrad epilogues / accumulating
  depthwise dgrads instead of separate add kernels.

Constraint of the kernels (met by every configuration of the reference): all channel counts except ``num_classes``
are multiples of 4, and depthwise convs use padding == dilation (``:129-130, 200-203``).
"""
import torch
import torch.nn as nn

from . import layers as L
from . import ops
from ._lib import SegHieroHipError

G1 = (1, 0, 1)      # 1x1 conv geometry (stride, pad, dil)


def _cbr(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class ProjectionHead(nn.Module):
    def __init__(self, dim_in, proj_dim=256, proj="convmlp"):
        super().__init__()
        if proj == "linear":
            self.proj = nn.Conv2d(dim_in, proj_dim, kernel_size=1, bias=False)
        elif proj == "convmlp":
            self.proj = nn.Sequential(nn.Conv2d(dim_in, dim_in, kernel_size=1, bias=False), nn.BatchNorm2d(dim_in),
                                      nn.ReLU(inplace=True), nn.Conv2d(dim_in, proj_dim, kernel_size=1, bias=False))
        else:
            raise ValueError(f"Unknown proj type: {proj}")


class DepthwiseSeparableConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, dilation=1, padding=1, bias=False):
        super().__init__()
        if kernel_size != 3 or padding != dilation or bias:
            raise SegHieroHipError("HIP depthwise kernel: 3x3, padding == dilation, no bias (all the reference uses)")
        self.depthwise = nn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, padding=padding,
                                   dilation=dilation, groups=in_channels, bias=bias)
        self.bn_dw = nn.BatchNorm2d(in_channels)
        self.act_dw = nn.ReLU(inplace=True)
        self.pointwise = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=bias)
        self.bn_pw = nn.BatchNorm2d(out_channels)
        self.act_pw = nn.ReLU(inplace=True)


class DepthwiseSeparableASPPModule(nn.Module):
    """Branch order of the (virtual) concat: [image_pool, branches[0] (1x1), branches[1..] (DS, dilated)]
    -- reference ``:100-114``.  The reference builds dense dilated 3x3 convs first and then replaces them
    (``:84-90, 125-131``); the throw-away construction is repeated so default init matches seed for seed."""

    def __init__(self, dilations, in_channels, channels):
        super().__init__()
        self.dilations = dilations
        self.branches = nn.ModuleList([_cbr(in_channels, channels)])
        for d in dilations[1:]:
            nn.Conv2d(in_channels, channels, kernel_size=3, padding=d, dilation=d, bias=False)   # discarded RNG draw
            self.branches.append(None)
        self.image_pool = nn.AdaptiveAvgPool2d(1)
        self.image_pool_conv = _cbr(in_channels, channels)
        for i, d in enumerate(dilations[1:], start=1):
            self.branches[i] = nn.Sequential(DepthwiseSeparableConv(in_channels, channels, kernel_size=3,
                                                                    dilation=d, padding=d, bias=False))


ASPP_GROUPED = __import__("os").environ.get("SEGHIERO_ASPP_GROUPED", "1") != "0"


def _aspp_branches_grouped(aspp, c4, cat, A, training, R, c4b=None):
    """The 1x1 branch and the depthwise-separable branches of the ASPP (``sep_aspp_contrast_head.py:100-131``) as ONE unit:

    * a branch whose dilation is >= the feature map (24 / 36 at 16 x 16) is ``w_centre * c4`` -- no depthwise kernel
      runs: its BatchNorm coefficients come from the per-channel statistics of c4 (computed once for all such branches) and are
      folded into the pointwise conv's loader (``ops.bn_finalize_scaled``);
    * the other depthwise convs run as before, their BatchNorm + ReLU deferred to the loader too;
    * all pointwise convs (+ the 1x1 branch) are one grouped launch (``ops.conv1x1_grouped_fprop``): a 512-tile GEMM instead of
      four under-filled ones with K slices and slab reduces; one BatchNorm + ReLU pass writes the four slices of the concat.

    c4b (bf16 compute mode): the bf16 copy of c4 -- the grouped launch then reads bf16 sources (c4b as is / through the centre-tap
    coefficients, a bf16 copy of the dilated depthwise output) with one MFMA product per tile; statistics, coefficient tables and the
    raw outputs stay fp32.
    Fills R["b0"], R["pw<i>"] and R["dw<i>"] / R["dwc<i>"]; False = geometry not eligible (the caller runs the branches one by one)."""
    n, cin, h, w = c4.shape
    nbr = len(aspp.dilations)
    if not (ASPP_GROUPED and training and L.FUSE_BN and ops.CONV_IMPL == "x6" and not ops._sync_on() and A % 128 == 0 and cin % 16 == 0
            and nbr <= 6):
        return False
    dev = c4.device
    m = n * h * w
    src4 = c4 if c4b is None else c4b
    sources, weights, bns = [(src4, None)], [aspp.branches[0][0].weight], [aspp.branches[0][1]]
    centre = [i for i, d in enumerate(aspp.dilations[1:], start=1) if d >= h and d >= w]      # only the centre tap touches the image
    if centre:                                           # y = w_centre * c4: statistics of c4 once, all such BatchNorms in one launch
        x_stats = ops.channel_stats(c4)
        cfs = [torch.empty((4, cin), device=dev, dtype=torch.float32) for _ in centre]
        isys = [torch.empty((cin,), device=dev, dtype=torch.float32) for _ in centre]
        ops.bn_finalize_multi(x_stats, m, ops.LIB.raw("sh_stats_tile_rows")(), [aspp.branches[i][0].bn_dw for i in centre], cfs,
                              dw_weights=[aspp.branches[i][0].depthwise.weight for i in centre], isy_list=isys)
        for i, cf, isy in zip(centre, cfs, isys):
            R[f"dwc{i}"] = (cf, isy)
    for i, d in enumerate(aspp.dilations[1:], start=1):
        ds = aspp.branches[i][0]
        if i in centre:
            sources.append((src4, R[f"dwc{i}"][0]))
        else:
            with ops.stored_as(torch.float32):          # (the dilated depthwise kernels are fp32)
                t, R[f"dw{i}"] = L.dw_fwd(c4, ds.depthwise.weight, d, ds.bn_dw, training, lazy=True)
            sources.append((t.y if c4b is None else t.y.to(torch.bfloat16), t.coefs))
        weights.append(ds.pointwise.weight)
        bns.append(ds.bn_pw)
    ycat = ops.new_act(n, nbr * A, h, w, dev)
    partials = ops.conv_partials(m, nbr * A, dev)
    if not ops.conv1x1_grouped_fprop(sources, weights, ycat, partials):
        raise SegHieroHipError("grouped ASPP launch rejected an eligible geometry")
    coefs_cat = torch.empty((4, nbr * A), device=dev, dtype=torch.float32)
    ops.bn_finalize_multi(partials, m, 64, bns, [coefs_cat[:, g * A:(g + 1) * A] for g in range(nbr)], partials_ld=nbr * A,
                          cols=[g * A for g in range(nbr)])
    ops.bn_act(ycat, coefs_cat, cat[:, A:(nbr + 1) * A], True)
    for g in range(nbr):
        rec = L.CBARec()
        src, cf = sources[g]
        rec.x = src if cf is None else L.Lazy(src, cf, grad32=True)      # (gradients into c4 / the depthwise branch are accumulated in fp32)
        rec.y, rec.out, rec.coefs = ycat[:, g * A:(g + 1) * A], None, coefs_cat[:, g * A:(g + 1) * A]
        rec.relu, rec.geom, rec.weight, rec.has_res, rec.mask = True, G1, weights[g], False, None
        R["b0" if g == 0 else f"pw{g}"] = rec
    return True


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, c1, c4, *params):
        # compute_dtype = torch.bfloat16: the convolutions whose input is a bf16-stored tensor (the decoder with act_dtype = bfloat16)
        # run in bf16 compute mode (ops.compute_as / csrc/conv_b16.hip)
        with ops.compute_as(mod.compute_dtype):
            return _HeadFn._forward(ctx, mod, c1, c4, *params)

    @staticmethod
    def backward(ctx, dlogits, demb):
        with ops.compute_as(ctx.mod.compute_dtype):
            return _HeadFn._backward(ctx, dlogits, demb)

    @staticmethod
    def _forward(ctx, mod, c1, c4, *params):
        training = mod.training
        c4 = ops.to_nhwc(c4)
        if ops.pm(c4)[1] != c4.shape[1]:
            c4 = ops.dense_copy(c4)
        n, cin, h, w = c4.shape
        dev = c4.device
        R = {}
        # bf16 compute mode: the 2048-channel consumers of c4 (projection head, ASPP pointwise convs) read a bf16 copy of it (8 M elements:
        # the trunk hands its stage outputs on as fp32 tensors); pooling, statistics and the dilated depthwise conv keep the fp32 tensor
        b16 = ops.b16() and training and L.FUSE_BN and mod.act_dtype == torch.bfloat16 and cin % 8 == 0
        c4b = c4.to(torch.bfloat16) if b16 else None
        # ---- projection head -> l2-normalised embedding (:12-30)
        proj = mod.proj_head.proj
        with ops.stored_as(torch.bfloat16 if b16 else torch.float32):
            if isinstance(proj, nn.Sequential):
                p1, R["p1"] = L.cba_fwd(c4b if b16 else c4, proj[0].weight, G1, proj[1], True, training, lazy=True)
                e_raw = L.conv_fwd(p1, proj[3].weight, None, G1)
                R["p1_out"] = p1
            else:
                e_raw = L.conv_fwd(c4b if b16 else c4, proj.weight, None, G1)
                R["emb_in"] = c4b if b16 else c4
        if ops.pm(e_raw)[1] != e_raw.shape[1]:
            e_raw = ops.dense_copy(e_raw)
        emb, norm = ops.l2norm_fwd(e_raw)
        R["emb"], R["norm"] = emb, norm
        # ---- ASPP: five branches write into channel slices of one buffer (:100-114)
        aspp = mod.aspp
        A = aspp.image_pool_conv[0].out_channels
        nb = len(aspp.dilations) + 1
        cat = ops.new_act(n, A * nb, h, w, dev, dtype=torch.bfloat16 if b16 else torch.float32)
        pooled = ops.avgpool_fwd(c4)
        ip, R["ip"] = L.cba_fwd(pooled, aspp.image_pool_conv[0].weight, G1, aspp.image_pool_conv[1], True, training)
        if b16:
            cat[:, 0:A].copy_(ip.expand(n, A, h, w))       # (sh_broadcast_hw writes fp32; [N, A] values into a 16 x 16 bf16 slice)
        else:
            ops.broadcast_hw(ip, cat[:, 0:A])
        if not _aspp_branches_grouped(aspp, c4, cat, A, training, R, c4b):
            _, R["b0"] = L.cba_fwd(c4, aspp.branches[0][0].weight, G1, aspp.branches[0][1], True, training, out=cat[:, A:2 * A])
            for i, d in enumerate(aspp.dilations[1:], start=1):
                ds = aspp.branches[i][0]
                t, R[f"dw{i}"] = L.dw_fwd(c4, ds.depthwise.weight, d, ds.bn_dw, training, lazy=True)
                _, R[f"pw{i}"] = L.cba_fwd(t, ds.pointwise.weight, G1, ds.bn_pw, True, training, out=cat[:, (i + 1) * A:(i + 2) * A])
        b, R["bt"] = L.cba_fwd(cat, mod.bottleneck[0].weight, G1, mod.bottleneck[1], True, training)
        # ---- decoder: x8 bilinear + C1 skip into one buffer (:231-242)
        # act_dtype = torch.bfloat16 (training, strip-walk geometry): the decoder's 128 x 128 tensors -- concat buffer, raw depthwise and
        # pointwise outputs -- are STORED as bf16 like the trunk's (ResNetBackbone.act_dtype; arithmetic / statistics / gradients fp32)
        H1, W1 = (c1.shape[2:] if mod.c1_bottleneck is not None else (h, w))
        stored = mod.act_dtype if (training and L.FUSE_BN and ops.CONV_IMPL == "x6" and ops.dw_lin_ok((n, A, H1, W1), 1)) else torch.float32
        with ops.stored_as(stored):
            if mod.c1_bottleneck is not None:
                c1 = ops.to_nhwc(c1)
                c1ch = mod.c1_bottleneck[0].out_channels
                cat2 = ops.new_act(n, A + c1ch, H1, W1, dev, dtype=stored)
                _, R["c1"] = L.cba_fwd(c1, mod.c1_bottleneck[0].weight, G1, mod.c1_bottleneck[1], True, training, out=cat2[:, A:])
                ops.bilinear_fwd(b, cat2[:, :A])
                xin = cat2
            else:
                xin = b
            for j, ds in enumerate(mod.sep_bottleneck):
                t, R[f"sdw{j}"] = L.dw_fwd(xin, ds.depthwise.weight, 1, ds.bn_dw, training, lazy=True)
                xin, R[f"spw{j}"] = L.cba_fwd(t, ds.pointwise.weight, G1, ds.bn_pw, True, training, lazy=True)
        logits = L.conv_fwd(xin, mod.cls_seg.weight, mod.cls_seg.bias, G1)
        R["cls_in"] = xin
        if training:
            bns = mod.__dict__.get("_bn_list")
            if bns is None:
                bns = mod.__dict__["_bn_list"] = [m for m in mod.modules() if isinstance(m, nn.BatchNorm2d)]
            L.bump_bn_counters(bns)
        ctx.mod, ctx.R, ctx.params, ctx.training = mod, R, params, training
        ctx.hw = (h, w)
        ctx.c1_needs = c1 is not None and torch.is_tensor(c1) and ctx.needs_input_grad[1]
        return logits, emb

    @staticmethod
    def _backward(ctx, dlogits, demb):
        if not ctx.training:
            raise SegHieroHipError("backward through eval-mode BatchNorm is not on the SegHiero hot path")
        mod, R = ctx.mod, ctx.R
        aspp = mod.aspp
        A = aspp.image_pool_conv[0].out_channels
        h, w = ctx.hw
        gm = L.GradMap()

        def put_cba(conv_w, bn, res):
            gm.put(conv_w, res[1]); gm.put(bn.weight, res[2]); gm.put(bn.bias, res[3])

        dc1 = dc4 = None
        if dlogits is not None:
            C = mod.cls_seg.out_channels
            dl = L.grad_as_nhwc_padded(dlogits, C)
            dx, dw = L.conv_bwd(R["cls_in"], mod.cls_seg.weight, G1, dl)
            gm.put(mod.cls_seg.weight, dw)
            part = ops.channel_stats(dl)
            red = torch.empty((4, C), device=dl.device, dtype=torch.float32)
            ops._call("sh_bn_bwd_finalize", part.data_ptr(), part.shape[0], C, None, None, 1.0, red[0].data_ptr(),
                      red[1].data_ptr(), red[2].data_ptr(), red[3].data_ptr(), None, None, ops._st())
            gm.put(mod.cls_seg.bias, red[1])              # "dbeta" slot = sum over pixels of dlogits
            for j in range(len(mod.sep_bottleneck) - 1, -1, -1):
                ds = mod.sep_bottleneck[j]
                res = L.cba_bwd(R[f"spw{j}"], ds.bn_pw, dx)
                put_cba(ds.pointwise.weight, ds.bn_pw, res)
                dx, dww, dg, db = L.dw_bwd(R[f"sdw{j}"], ds.bn_dw, res[0])
                gm.put(ds.depthwise.weight, dww); gm.put(ds.bn_dw.weight, dg); gm.put(ds.bn_dw.bias, db)
            if mod.c1_bottleneck is not None:
                dbt = ops.bilinear_bwd(dx[:, :A], h, w)
                res = L.cba_bwd(R["c1"], mod.c1_bottleneck[1], dx[:, A:], need_dx=ctx.c1_needs)
                put_cba(mod.c1_bottleneck[0].weight, mod.c1_bottleneck[1], res)
                dc1 = res[0]
            else:
                dbt = dx
            res = L.cba_bwd(R["bt"], mod.bottleneck[1], dbt)
            put_cba(mod.bottleneck[0].weight, mod.bottleneck[1], res)
            dcat = res[0]
            # gradients into C4: 1x1 branch first (allocates), the rest accumulate into it
            res = L.cba_bwd(R["b0"], aspp.branches[0][1], dcat[:, A:2 * A], dx32=True)      # dc4 is accumulated in fp32
            put_cba(aspp.branches[0][0].weight, aspp.branches[0][1], res)
            dc4 = res[0]
            for i in range(1, len(aspp.dilations)):
                ds = aspp.branches[i][0]
                res = L.cba_bwd(R[f"pw{i}"], ds.bn_pw, dcat[:, (i + 1) * A:(i + 2) * A])
                put_cba(ds.pointwise.weight, ds.bn_pw, res)
                if f"dwc{i}" in R:
                    # centre-tap branch (y = w_centre * c4 never existed): the BatchNorm backward on c4 with the x-domain
                    # coefficients yields d/dc4 directly; the depthwise weight gradient is closed-form (sh_dw_center_wgrad)
                    coefs_x, isy = R[f"dwc{i}"]
                    xin = R[f"pw{i}"].x.y
                    dxc, dg, db, _ = ops.bn_backward(res[0], None, xin, coefs_x, ds.bn_dw.weight, 2, grad32=True)
                    ops._call("sh_axpy", dc4.data_ptr(), dxc.data_ptr(), 1.0, dc4.numel(), ops._st())
                    dww = L.new_grad(ds.depthwise.weight)
                    # (a centre tap that is exactly 0 leaves no trace in dgamma: those channels read the masked gradient itself)
                    gmask = res[0].g if isinstance(res[0], L.GradPack) and xin.dtype == torch.float32 else None
                    ops.dw_center_wgrad(dg, ds.bn_dw.weight, isy, ds.depthwise.weight, ds.bn_dw.eps, dww, g=gmask, x=xin, mean_x=coefs_x[0])
                else:
                    _, dww, dg, db = L.dw_bwd(R[f"dw{i}"], ds.bn_dw, res[0], dx_accumulate_into=dc4)
                gm.put(ds.depthwise.weight, dww); gm.put(ds.bn_dw.weight, dg); gm.put(ds.bn_dw.bias, db)
            dip = ops.sum_hw(ops.f32(dcat[:, 0:A]))
            res = L.cba_bwd(R["ip"], aspp.image_pool_conv[1], dip)
            put_cba(aspp.image_pool_conv[0].weight, aspp.image_pool_conv[1], res)
            ops.avgpool_bwd(res[0], dc4, accumulate=True)
        if demb is not None:
            de = ops.l2norm_bwd(demb, R["emb"], R["norm"])
            proj = mod.proj_head.proj
            if isinstance(proj, nn.Sequential):
                dp1, dw3 = L.conv_bwd(R["p1_out"], proj[3].weight, G1, L.grad_as_nhwc_padded(de, de.shape[1]))
                gm.put(proj[3].weight, dw3)
                res = L.cba_bwd(R["p1"], proj[1], dp1, addend=dc4, dx32=True)
                put_cba(proj[0].weight, proj[1], res)
                dc4 = res[0]
            else:
                dc4n, dwp = L.conv_bwd(R["emb_in"], proj.weight, G1, L.grad_as_nhwc_padded(de, de.shape[1]), addend=dc4, dx32=True)
                gm.put(proj.weight, dwp)
                dc4 = dc4n
        ctx.R = None
        gm.flush(ctx.params)
        return (None, dc1, dc4) + gm.ordered(ctx.params)


class DepthwiseSeparableASPPContrastHead(nn.Module):
    def __init__(self, in_channels: int, c1_in_channels: int, c1_channels: int, aspp_channels: int, dilations: tuple,
                 num_classes: int, proj_dim: int = 256, proj_type: str = "convmlp"):
        super().__init__()
        self.proj_head = ProjectionHead(dim_in=in_channels, proj_dim=proj_dim, proj=proj_type)
        self.register_buffer("step", torch.zeros(1, dtype=torch.long))
        self.aspp = DepthwiseSeparableASPPModule(dilations=dilations, in_channels=in_channels, channels=aspp_channels)
        total_aspp_ch = aspp_channels * (len(dilations) + 1)
        self.bottleneck = _cbr(total_aspp_ch, aspp_channels)
        if c1_in_channels > 0:
            self.c1_bottleneck = _cbr(c1_in_channels, c1_channels)
        else:
            self.c1_bottleneck = None
            c1_channels = 0
        self.sep_bottleneck = nn.Sequential(
            DepthwiseSeparableConv(aspp_channels + c1_channels, aspp_channels, kernel_size=3, padding=1, bias=False),
            DepthwiseSeparableConv(aspp_channels, aspp_channels, kernel_size=3, padding=1, bias=False))
        self.cls_seg = nn.Conv2d(aspp_channels, num_classes, kernel_size=1)
        self.act_dtype = torch.float32          # torch.bfloat16: the decoder stores its activations as bf16 (see _HeadFn.forward)
        self.compute_dtype = torch.float32      # torch.bfloat16 (with act_dtype = bfloat16): bf16 compute mode for those convolutions
        self.align_corners = False
        for name, v in (("in_channels", in_channels), ("aspp_channels", aspp_channels), ("c1_channels", c1_channels),
                        ("proj_dim", proj_dim), ("c1_in_channels", max(c1_in_channels, 0))):
            if v % 4:
                raise SegHieroHipError(f"{name}={v}: the HIP kernels need channel counts that are multiples of 4")

    def forward(self, inputs: list):
        """inputs: [C1, C2, C3, C4] (only inputs[0] and inputs[-1] are read, as in the reference).
        Returns (logits [B,num_classes,H/4,W/4], embedding [B,proj_dim,H/32,W/32])."""
        ops._require_gpu(inputs[-1])
        self.step += 1
        c1 = inputs[0] if self.c1_bottleneck is not None else None
        return _HeadFn.apply(self, c1, inputs[-1], *L.params_of(self))


# ----------------------------------------------------------------------------- aux head (train.py:169-173)
class _AuxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, c3, *params):
        c3 = ops.to_nhwc(c3)
        out, rec = L.cba_fwd(c3, mod[0].weight, G1, mod[1], True, mod.training)
        if mod.training:
            L.bump_bn_counters([mod[1]])
        ctx.mod, ctx.rec, ctx.params, ctx.training = mod, rec, params, mod.training
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise SegHieroHipError("backward through eval-mode BatchNorm is not on the SegHiero hot path")
        mod = ctx.mod
        d = L.grad_as_nhwc_padded(dout, dout.shape[1])
        dx, dw, dg, db, _ = L.cba_bwd(ctx.rec, mod[1], d, need_dx=ctx.needs_input_grad[1])
        gm = L.GradMap()
        gm.put(mod[0].weight, dw); gm.put(mod[1].weight, dg); gm.put(mod[1].bias, db)
        ctx.rec = None
        gm.flush(ctx.params)
        return (None, dx) + gm.ordered(ctx.params)


class AuxHead(nn.Sequential):
    """``nn.Sequential(Conv2d(c3, n_fine, 1, bias=False), BatchNorm2d(n_fine), ReLU)`` of reference
    ``train.py:169-173`` (same state_dict keys ``0.weight, 1.weight, ...``) running on the HIP kernels."""

    def __init__(self, in_channels: int, n_fine: int):
        super().__init__(nn.Conv2d(in_channels, n_fine, kernel_size=1, bias=False), nn.BatchNorm2d(n_fine),
                         nn.ReLU(inplace=True))

    def forward(self, c3):
        ops._require_gpu(c3)
        return _AuxFn.apply(self, c3, *L.params_of(self))
