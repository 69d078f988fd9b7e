"""Forward / backward building blocks shared by the backbone and the head.

These are plain functions over torch tensors (device memory) that launch the HIP kernels through ``ops``; the
``torch.nn.Conv2d`` / ``BatchNorm2d`` objects they receive are used purely as parameter / buffer containers (so that
state_dict keys and shapes equal the reference's) -- their ``forward`` is never called.

The backward pass is scheduled by hand (``backbone._BackboneFn``, ``head._HeadFn``) instead of being traced by autograd:
that is what allows residual / concat gradients to be summed inside the dgrad epilogue, BN statistics to be produced by
the conv epilogue, and concat buffers to be written in place.
"""
import torch
import torch.nn as nn

from . import ops


def conv_geom(conv):
    return conv.stride[0], conv.padding[0], conv.dilation[0]


def W(m):
    """m.weight / m.bias (Bi) without nn.Module.__getattr__ (~1 us per access, several per layer and step); always the live Parameter."""
    return m._parameters["weight"]


def Bi(m):
    return m._parameters["bias"]


def to_native_layout(module):
    """Put every dense KxK conv weight of `module` into channels_last (OHWI) memory, in place.  1x1 / depthwise /
    the 3-channel stem weight are layout-neutral or handled separately.  Idempotent; call after .to(device) or
    load_state_dict (both preserve it)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d) and m.groups == 1 and m.kernel_size != (1, 1) and m.in_channels % 4 == 0:
            if not m.weight.data.is_contiguous(memory_format=torch.channels_last):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return module


def params_of(mod):
    """list(mod.parameters()), walked once per module (the parameter set of these modules is fixed after construction)."""
    ps = mod.__dict__.get("_param_list")
    if ps is None:
        ps = mod.__dict__["_param_list"] = list(mod.parameters())
    return ps


def bump_bn_counters(bns):
    """num_batches_tracked += 1 for all train-mode BN layers of one forward, in one launch."""
    ctrs = [b.num_batches_tracked for b in bns if b.num_batches_tracked is not None]
    if ctrs:
        torch._foreach_add_(ctrs, 1)


# ----------------------------------------------------------------------------- deferred BatchNorm + ReLU
FUSE_BN = __import__("os").environ.get("SEGHIERO_FUSE_BN", "1") != "0"      # 0: every BatchNorm + ReLU is its own pass (round-1 path)


class Lazy:
    """relu(y * scale + shift) of a raw conv output, NOT materialised: the consumer convolution applies it in its loader
    (ops.conv_fprop_aff / conv_wgrad(aff=...)), so conv -> BN -> ReLU -> conv chains never write or re-read the activated
    tensor.  `materialize()` is the fallback for consumers without a fused loader."""
    __slots__ = ("y", "coefs", "_out", "grad32")

    def __init__(self, y, coefs, grad32=False):
        # grad32: the gradient w.r.t. this activation goes to a kernel that only takes fp32 (a depthwise producer's backward)
        self.y, self.coefs, self._out, self.grad32 = y, coefs, None, grad32

    @property
    def shape(self):
        return self.y.shape

    @property
    def device(self):
        return self.y.device

    def materialize(self):
        if self._out is None:
            n, c, h, w = self.y.shape
            ld = ops.pad4(c)
            self._out = ops.new_act(n, c, h, w, self.y.device, ld=ld, zero=ld != c, dtype=self.y.dtype)
            ops.bn_act(self.y, self.coefs, self._out, True)
        return self._out


class LazyAffine:
    """y * scale + shift (BatchNorm, NO ReLU) of a raw conv output, not materialised: the downsample branch of a residual block,
    consumed only by the block's final BatchNorm + add + ReLU pass (ops.bn_act(..., res_coefs=...))."""
    __slots__ = ("y", "coefs")

    def __init__(self, y, coefs):
        self.y, self.coefs = y, coefs


class GradPack:
    """Gradient w.r.t. a deferred activation as the dgrad epilogue leaves it (ops.conv_dgrad_bnb): g = relumask * dx and the
    (sum g, sum g*xhat) partials per 64 rows -- the producer's BatchNorm backward starts from its finalize step."""
    __slots__ = ("g", "partials")

    def __init__(self, g, partials):
        self.g, self.partials = g, partials


def dense(x):
    return x.materialize() if isinstance(x, Lazy) else x


# ----------------------------------------------------------------------------- conv + BN (+res) (+ReLU)
class CBARec:
    __slots__ = ("x", "y", "out", "coefs", "relu", "geom", "weight", "has_res", "mask")      # mask: ReLU quad mask of `out` (ops.new_relu_mask) or None


def _bn_coefs(bn, partials, count, training, c, device):
    if training:
        mom = 0.1 if bn.momentum is None else bn.momentum
        pr, bf = bn._parameters, bn._buffers
        return ops.bn_finalize(partials, count, pr["weight"], pr["bias"], bn.eps, mom, bf["running_mean"], bf["running_var"], c, device)
    return ops.bn_eval_coefs(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def _fprop(x, weight, bias, y, partials, s, p, d):
    """conv forward of a tensor or a deferred activation (fused loader where the geometry has one)."""
    if isinstance(x, Lazy):
        if ops.conv_fprop_aff(x.y, x.coefs, weight, bias, y, partials, s, p, d):
            return
        x = x.materialize()
    ops.conv_fprop(x, weight, bias, y, partials, s, p, d)


def _wgrad(x, dy, dw, s, p, d):
    """dy: tensor or ops.DeferredDy (cba_bwd defers only where ops.lin_ok holds; the x-through-BatchNorm loader composes with it)."""
    if isinstance(dy, ops.DeferredDy) and isinstance(x, Lazy) and not (x._out is None and ops.wgrad_aff_ok(x.y, dw, s, p, d)):
        dy = dy.materialize()
    if isinstance(x, Lazy):
        if x._out is None and ops.wgrad_aff_ok(x.y, dw, s, p, d):
            ops.conv_wgrad(x.y, dy, dw, s, p, d, side=True, aff=x.coefs)
            return
        x = x.materialize()
    ops.conv_wgrad(x, dy, dw, s, p, d, side=True)


def _dgrad(rec_x, dy, weight, s, p, d, addend=None, pack_for=None, grad32=False):
    """Gradient w.r.t. a conv input.  For a deferred activation the dgrad epilogue also runs the front half of the producer's
    BatchNorm backward (-> GradPack); otherwise a plain tensor.  pack_for: the CBARec of the residual block whose OUTPUT is this
    conv's input (out = relu(bn(y) + identity)): the same fusion with the mask taken from `out`.
    In bf16 compute mode the result is stored like the input it belongs to (ops.grad_dtype); fp32 when its consumer only takes fp32
    (grad32, or a depthwise producer: Lazy.grad32) or when only the fp32-accurate kernels have the geometry."""
    like = rec_x.y if isinstance(rec_x, Lazy) else rec_x
    g32 = grad32 or (isinstance(rec_x, Lazy) and rec_x.grad32)
    if ops.grad_dtype(like, g32) == torch.bfloat16:
        out = _dgrad_as(torch.bfloat16, rec_x, dy, weight, s, p, d, addend, pack_for)
        if out is not None:
            return out
    return _dgrad_as(torch.float32, rec_x, dy, weight, s, p, d, addend, pack_for)


def _dgrad_as(gdt, rec_x, dy, weight, s, p, d, addend, pack_for):
    """_dgrad with its result stored as `gdt`.  bf16 results exist in the bf16 compute kernels only: None when they have no
    instantiation for the geometry (nothing launched; the caller repeats with fp32)."""
    n, c, h, w = rec_x.shape
    dev = (rec_x.y if isinstance(rec_x, Lazy) else rec_x).device
    strict = gdt != torch.float32
    new = lambda: ops.new_act(n, c, h, w, dev, dtype=gdt)
    parts = lambda: torch.empty((-(-n * h * w // 64), 2, c), device=dev, dtype=torch.float32)

    def plain(cur, dx):
        if not strict:
            ops.conv_dgrad(cur, weight, dx, s, p, d, addend=addend)
            return dx
        return dx if ops._dgrad_b16(cur, weight, dx, s, p, d, addend=addend) else None

    if isinstance(dy, ops.DeferredDy):
        if pack_for is None:
            if isinstance(rec_x, Lazy) and FUSE_BN:
                g, partials = new(), parts()
                if ops.conv_dgrad_lin(dy, weight, g, addend=addend, bnb=(rec_x.y, rec_x.coefs, partials)):
                    return GradPack(g, partials)
            elif not isinstance(rec_x, Lazy):
                dx = new()
                if ops.conv_dgrad_lin(dy, weight, dx, addend=addend):
                    return dx
        dy = dy.materialize()
    if pack_for is not None and FUSE_BN and not isinstance(rec_x, Lazy):
        g, partials = new(), parts()
        if ops.conv_dgrad_bnb(dy, weight, g, pack_for.y, pack_for.coefs, True, partials, s, p, d, addend=addend,
                              out_prev=pack_for.out if pack_for.mask is None else pack_for.mask):
            return GradPack(g, partials)
        return plain(dy, g)
    if isinstance(rec_x, Lazy) and FUSE_BN:
        g, partials = new(), parts()
        if ops.conv_dgrad_bnb(dy, weight, g, rec_x.y, rec_x.coefs, True, partials, s, p, d, addend=addend):
            return GradPack(g, partials)
        return plain(dy, g)
    return plain(dy, new())


def cba_fwd(x, weight, geom, bn, relu, training, residual=None, out=None, lazy=False):
    """out = [relu](BN(conv(x, weight)) [+ residual]); `out` may be a channel slice of a concat buffer.
    x may be a deferred activation (Lazy).  lazy=True (training, ReLU, no residual, no `out`): the BatchNorm + ReLU of THIS
    layer is deferred too -- a Lazy is returned and no elementwise pass runs."""
    s, p, d = geom
    n, _, h, w = x.shape
    o, _, kh, kw = weight.shape
    ho, wo = ops.conv_out_hw(h, w, kh, kw, s, p, d)
    ld = ops.pad4(o)
    m = n * ho * wo
    mask = None
    if not training and ops.FUSE_EVAL and ops.CONV_IMPL == "x6":
        # inference: eval-mode BN (+ residual) (+ ReLU) in the conv epilogue; y is never materialised
        coefs = _bn_coefs(bn, None, m, False, o, x.device)
        if out is None:
            out = ops.new_act(n, o, ho, wo, x.device, ld=ld, zero=ld != o)
        ops.conv_fprop_act(dense(x), weight, coefs, out, relu, residual, s, p, d)
        y = None
    else:
        y = ops.new_act(n, o, ho, wo, x.device, ld=ld, zero=ld != o, dtype=ops.stored_dtype())      # (bf16 inside ops.stored_as)
        partials = ops.conv_partials(m, o, x.device) if training else None
        _fprop(x, weight, None, y, partials, s, p, d)
        coefs = _bn_coefs(bn, partials, m, training, o, x.device)
        if lazy and FUSE_BN and training and relu and residual is None and out is None and ops.CONV_IMPL == "x6":
            out = Lazy(y, coefs)
        elif lazy and FUSE_BN and training and not relu and residual is None and out is None:
            out = LazyAffine(y, coefs)
        else:
            if out is None:
                out = ops.new_act(n, o, ho, wo, x.device, ld=ld, zero=ld != o, dtype=ops.stored_dtype())
            # a residual block's output: its backward needs only the ReLU mask of `out` -- kept as a quad mask (1/16 of the bytes)
            if training and relu and residual is not None and ops.RELU_MASK and FUSE_BN and o % 4 == 0 and ops.CONV_IMPL == "x6":
                mask = ops.new_relu_mask(n, o, ho, wo, x.device)
            if isinstance(residual, LazyAffine):
                ops.bn_act(y, coefs, out, relu, residual.y, res_coefs=residual.coefs, mask=mask)
            else:
                ops.bn_act(y, coefs, out, relu, residual, mask=mask)
    rec = CBARec()
    rec.x, rec.y, rec.coefs, rec.relu, rec.geom, rec.weight = x, y, coefs, relu, geom, weight
    rec.out = None if isinstance(out, (Lazy, LazyAffine)) else out
    rec.has_res = residual is not None
    rec.mask = mask
    return out, rec


def new_grad(param):
    """Output buffer for the gradient of `param`: its slice of the data-parallel gradient arena when an exchange is in
    flight (ddp.GradSync: the all-reduce then needs no staging copy), else a fresh tensor laid out like the parameter."""
    from . import ddp
    buf = ddp.grad_buffer(param)
    return torch.empty_like(param) if buf is None else buf


def cba_bwd(rec, bn, dout, need_dx=True, addend=None, want_dres=False, scatter_into=None, pack_for=None, dx32=False):
    """-> (dx, dweight, dgamma, dbeta, dres).  `addend` is summed into dx by the dgrad epilogue;
    `scatter_into` (1x1 strided convs) accumulates the result into an existing dx instead.  dx32: dx goes to an fp32-only consumer."""
    s, p, d = rec.geom
    # bf16 compute mode: dy (the gradient w.r.t. this conv's raw output) is stored like that output, except where its consumers are
    # the fp32-accurate kernels: convs whose input is an fp32 tensor, and strided input gradients the one-product kernel does not take
    xin = rec.x.y if isinstance(rec.x, Lazy) else rec.x
    strided = s > 1 or scatter_into is not None
    dy32 = xin.dtype == torch.float32 or (strided and not ops.strided_dgrad_b16_ok(rec.weight, xin.shape[1], s, p, d, scatter_into is not None))
    # ReLU mask: from `out` only where a residual was added; otherwise recomputed from y (one activation read less).
    # dout may be a GradPack (mask applied, statistics partials done by the consumer's dgrad epilogue).
    mode = 0 if not rec.relu else (1 if rec.has_res else 2)
    # second half of the BatchNorm backward in the loaders of this conv's dgrad / wgrad (1x1 convs): no apply pass, no dy tensor
    defer = FUSE_BN and scatter_into is None and pack_for is None and ops.lin_ok(rec.x.shape, rec.weight, s, p, d)
    dy, dgamma, dbeta, dres = ops.bn_backward(dout, (rec.out if rec.mask is None else rec.mask) if mode == 1 else None, rec.y, rec.coefs,
                                              W(bn), mode, want_dres, defer=defer, grad32=dy32)
    dw = new_grad(rec.weight)
    if not ops.WGRAD_AFTER_DGRAD:
        _wgrad(rec.x, dy, dw, s, p, d)
    dx = None
    if scatter_into is not None:
        ops.conv_dgrad(dy, rec.weight, scatter_into, s, p, d, mode=1)
        dx = scatter_into
    elif need_dx:
        dx = _dgrad(rec.x, dy, rec.weight, s, p, d, addend=addend, pack_for=pack_for, grad32=dx32)
    if ops.WGRAD_AFTER_DGRAD:
        # enqueued after the dgrad: the side stream then starts this (MFMA-bound) wgrad when the dgrad has finished, i.e.
        # next to the HBM-bound BatchNorm backward of the previous layer instead of next to another MFMA-bound kernel
        _wgrad(rec.x, dy, dw, s, p, d)
    return dx, dw, dgamma, dbeta, dres


# ----------------------------------------------------------------------------- depthwise 3x3 + BN + ReLU
class DWRec:
    __slots__ = ("x", "y", "out", "coefs", "dil", "weight")


def dw_fwd(x, weight, dil, bn, training, lazy=False):
    """depthwise 3x3 -> BN -> ReLU; lazy=True defers the BatchNorm + ReLU to the pointwise conv's loader (-> Lazy)."""
    n, c, h, w = x.shape
    y = ops.new_act(n, c, h, w, x.device, dtype=ops.stored_dtype())
    partials = torch.empty((ops.dw_partials_rows(n, h, w), 2, c), device=x.device) if training else None
    if isinstance(x, Lazy) and x._out is None:
        ops.dwconv_fprop(x.y, weight, y, partials, dil, aff=x.coefs)      # the producer's BN + ReLU in the depthwise loader
    else:
        x = dense(x)
        ops.dwconv_fprop(x, weight, y, partials, dil)
    coefs = _bn_coefs(bn, partials, n * h * w, training, c, x.device)
    if lazy and FUSE_BN and training and ops.CONV_IMPL == "x6":
        out = Lazy(y, coefs, grad32=True)              # the depthwise backward kernels take fp32 gradients
    else:
        out = ops.new_act(n, c, h, w, x.device)
        ops.bn_act(y, coefs, out, True)
    rec = DWRec()
    rec.x, rec.y, rec.coefs, rec.dil, rec.weight = x, y, coefs, dil, weight
    rec.out = None if isinstance(out, Lazy) else out
    return out, rec


def dw_bwd(rec, bn, dout, dx_accumulate_into=None):
    """-> (dx, dweight, dgamma, dbeta); with dx_accumulate_into the input gradient is added into that tensor."""
    # second half of the BatchNorm backward in the loaders of the depthwise dgrad / wgrad (strip-walk kernels): no apply pass
    if hasattr(dout, "partials"):
        dout.g = ops.f32(dout.g)
    else:
        dout = ops.f32(dout)
    dy, dgamma, dbeta, _ = ops.bn_backward(dout, None, rec.y, rec.coefs, W(bn), 2, defer=FUSE_BN and ops.dw_lin_ok(rec.y.shape, rec.dil),
                                           grad32=True)
    dw = new_grad(rec.weight)
    if isinstance(rec.x, Lazy) and rec.x._out is None:
        ops.dwconv_wgrad(rec.x.y, dy, dw, rec.dil, side=True, aff=rec.x.coefs)
    else:
        ops.dwconv_wgrad(dense(rec.x), dy, dw, rec.dil, side=True)
    if dx_accumulate_into is not None:
        ops.dwconv_dgrad(dy, rec.weight, dx_accumulate_into, rec.dil, accumulate=True)
        dx = dx_accumulate_into
    else:
        n, c, h, w = rec.x.shape
        dx = ops.new_act(n, c, h, w, rec.x.device)
        if isinstance(rec.x, Lazy) and FUSE_BN:
            # the input is a deferred activation: its BatchNorm backward starts in this dgrad's epilogue (-> GradPack)
            partials = torch.empty((ops.dw_partials_rows(n, h, w), 2, c), device=dx.device, dtype=torch.float32)
            ops.dwconv_dgrad_bnb(dy, rec.weight, dx, rec.x.y, rec.x.coefs, partials, rec.dil)
            dx = GradPack(dx, partials)
        else:
            ops.dwconv_dgrad(dy, rec.weight, dx, rec.dil)
    return dx, dw, dgamma, dbeta


# ----------------------------------------------------------------------------- plain conv (+bias), no BN
def conv_fwd(x, weight, bias, geom):
    s, p, d = geom
    n, _, h, w = x.shape
    o, _, kh, kw = weight.shape
    ho, wo = ops.conv_out_hw(h, w, kh, kw, s, p, d)
    ld = ops.pad8(o) if ops.b16() else ops.pad4(o)          # bf16 compute mode: the one-product kernels move 8 columns per lane
    y = ops.new_act(n, o, ho, wo, x.device, ld=ld, zero=ld != o)
    _fprop(x, weight, bias, y, None, s, p, d)
    return y


def conv_bwd(x, weight, geom, dy, need_dx=True, addend=None, dx32=False):
    """dy must be NHWC with its padding lanes zeroed.  -> (dx (tensor, or GradPack for a deferred input), dweight)."""
    s, p, d = geom
    dw = new_grad(weight)
    _wgrad(x, dy, dw, s, p, d)
    dx = None
    if need_dx:
        dx = _dgrad(x, dy, weight, s, p, d, addend=addend, grad32=dx32)
    return dx, dw


def grad_as_nhwc_padded(g, c):
    """Incoming autograd gradient -> NHWC tensor whose pixel stride is a multiple of 4 with zeroed padding."""
    g = ops.to_nhwc(g)
    ld = ops.pm(g)[1]
    q = 8 if ops.b16() else 4                       # bf16 compute mode: rows of pad8(c) for the one-product kernels' 8-channel loads
    if ld == c and c % q == 0:
        return g
    return ops.dense_copy(g, ld=ops.pad8(c) if q == 8 else ops.pad4(c))


class GradMap:
    """Collects parameter gradients produced by a hand-scheduled backward and hands them back in parameter order."""

    def __init__(self):
        self.g = {}

    def put(self, param, grad):
        if grad is None:
            return
        if grad.shape != param.shape:
            grad = grad.reshape(param.shape)
        key = id(param)
        if key in self.g:
            ops.join_wgrad()             # accumulating on the compute stream: the addends may come from the wgrad stream
            grad = self.g[key] + grad
        self.g[key] = grad

    def ordered(self, params):
        ops.join_wgrad()                 # weight gradients run on a side stream (ops.conv_wgrad); order them before the consumers
        return tuple(self.g.get(id(p)) for p in params)

    def flush(self, params):
        """Hand the (final) gradients of `params` to the data-parallel exchange while backward continues (ddp.GradSync)."""
        from . import ddp
        if ddp._ACTIVE is not None:
            ops.join_wgrad()
            ddp.early_flush([(p, self.g.get(id(p))) for p in params])
