"""Typed Python wrappers over the C ABI (include/seghiero_hip.h) for torch tensors.

Everything here is plumbing: PyTorch provides device memory (the caching allocator) and the current
HIP stream; all arithmetic happens in libseghiero_hip.so.  Activations are logical NCHW tensors whose
memory is NHWC (``channels_last``), possibly a channel slice of a wider buffer (pixel stride ``ld``).
"""
import ctypes

import torch

from ._lib import LIB, SegHieroHipError, status_text

import os

# "x6": fp32-accurate 6 x bf16-split MFMA path (default); "f32": v_mfma_f32_32x32x2_f32 path.
CONV_IMPL = os.environ.get("SEGHIERO_CONV", "x6")

_PROF = None          # when set (see `profile()`), every C-ABI call is bracketed by HIP events on its launch stream


def _call(name, *args, cost=None, key=None):
    """Launch one C-ABI entry point on the current stream.  `cost` = (algorithmic flops, algorithmic bytes); `key` = shape label
    (profiling only: per-shape rows next to the per-family ones)."""
    if _PROF is None:
        return LIB.call(name, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = LIB.call(name, *args)
    e1.record()
    _PROF.append((name, cost, e0, e1, key))
    return rc


class profile:
    """Context manager: per-kernel-family device time measured with HIP events recorded on the launch stream
    (torch's current stream), plus the algorithmic flops / bytes the callers declare.  Used by bench.py for the
    roofline line; adds event overhead, so never wrap the timed region with it.  The weight-gradient side stream is switched
    off inside the context: kernels that co-run share the CUs, which would inflate every per-kernel duration."""

    def __enter__(self):
        global _PROF, WGRAD_ASYNC
        _PROF = []
        self._async = WGRAD_ASYNC
        WGRAD_ASYNC = False
        return self

    def __exit__(self, *exc):
        global _PROF, WGRAD_ASYNC
        WGRAD_ASYNC = self._async
        rec, _PROF = _PROF, None
        self.rec = rec
        torch.cuda.synchronize()
        self.rows, self.shapes = {}, {}
        for name, cost, e0, e1, key in rec:
            ms = e0.elapsed_time(e1)
            for table, k in ((self.rows, name), (self.shapes, (name, key))):
                if k[1] is None and table is self.shapes:
                    continue
                r = table.setdefault(k, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
                r["calls"] += 1
                r["ms"] += ms
                if cost:
                    r["flops"] += cost[0]
                    r["bytes"] += cost[1]
        return False


_STREAM = None        # explicit launch stream (raw hipStream_t) while the weight-gradient side stream is the target: see conv_wgrad


def _st():
    # raw hipStream_t of torch's current stream on the current device (the C call: no Stream object is built per launch)
    if _STREAM is not None:
        return _STREAM
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _rp(t, k):
    """Pointer of row k of a dense 2-D tensor (coefficient tables): `t[k].data_ptr()` without building a view per launch argument."""
    return t.data_ptr() + k * t.stride(0) * t.element_size()


def _require_gpu(t):
    if not t.is_cuda:
        raise SegHieroHipError("seghiero_amd runs on the MI355X only: tensor is on %s (no CPU fallback)" % t.device)


# ----------------------------------------------------------------------------- tensor helpers
def new_act(n, c, h, w, device, ld=None, zero=False, dtype=torch.float32):
    """Logical [n,c,h,w] tensor over NHWC memory with pixel stride ld (>= c, default c); fp32, or bf16 for STORED activations
    (stored_dtype())."""
    if (ld is None or ld == c) and not zero:        # the common case in one call (no permute / slice views: ~2 us of host time each)
        return torch.empty_strided((n, c, h, w), (h * w * c, 1, w * c, c), device=device, dtype=dtype)
    ld = c if ld is None else ld
    buf = (torch.zeros if zero else torch.empty)((n, h, w, ld), device=device, dtype=dtype)
    t = buf.permute(0, 3, 1, 2)
    return t if ld == c else t[:, :c]


# bf16 activation storage (BASELINE configs[4]; include/seghiero_hip.h "bf16 ACTIVATION STORAGE"): inside `with stored_as(torch.bfloat16)`
# -- the training forward of the ResNet trunk when its act_dtype says so -- the raw conv outputs and block outputs that layers.py
# allocates are bf16; every kernel computes in fp32, BatchNorm statistics come from the fp32 accumulators, gradients stay fp32.
_STORED = torch.float32


def stored_dtype():
    return _STORED


class stored_as:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _STORED
        self.prev, _STORED = _STORED, self.dtype

    def __exit__(self, *exc):
        global _STORED
        _STORED = self.prev


def pad4(c):
    return (c + 3) & ~3


def pad8(c):
    return (c + 7) & ~7


def f32(t):
    """fp32 view of a (gradient) tensor for the fp32-only kernels: the tensor itself, or a widened copy of a bf16 one."""
    return t if t is None or t.dtype == torch.float32 else t.float()


# bf16 COMPUTE mode (BASELINE configs[4]; include/seghiero_hip.h "bf16 COMPUTE mode", csrc/conv_b16.hip): inside `with compute_as(torch.bfloat16)`
# every dense convolution whose geometry has an instantiation runs ONE bf16 MFMA product per tile on operands rounded once to bf16 in the
# loader (fp32 accumulate, fp32 BatchNorm statistics, fp32 weight gradients) instead of the fp32-accurate six-product plan; the others
# (the 4-channel stem, strided input gradients, channel counts that are not multiples of 8) stay on the fp32-accurate kernels.
_B16 = False


def b16():
    return _B16


class compute_as:
    def __init__(self, dtype):
        self.on = dtype == torch.bfloat16

    def __enter__(self):
        global _B16
        self.prev, _B16 = _B16, self.on and CONV_IMPL == "x6"

    def __exit__(self, *exc):
        global _B16
        _B16 = self.prev


def pm(t):
    """(data_ptr, ld) of a logical-NCHW tensor with NHWC memory; raises if the strides are anything else.

    Element (n,c,h,w) must live at ((n*H + h)*W + w)*ld + c; strides of size-1 dims are ignored."""
    return _pm(t, False)[:2]


def pmx(t):
    """(data_ptr, ld, 1 if bf16 else 0): as pm, for the entry points that take bf16-stored activations (act_flags)."""
    return _pm(t, True)


_F32, _BF16 = torch.float32, torch.bfloat16


def _pm(t, allow_bf16):
    # (hot: ~1 800 calls per ResNet-101 step -- one pass over shape / strides, no helper calls)
    dt = t.dtype
    if not t.is_cuda:
        _require_gpu(t)
    if t.dim() != 4 or not (dt is _F32 or (allow_bf16 and dt is _BF16)):
        raise SegHieroHipError(f"expected a 4-D fp32 tensor, got {tuple(t.shape)} {t.dtype}")
    n, c, h, w = t.shape
    s0, s1, s2, s3 = t.stride()
    if w > 1:
        ld = s3
    elif h > 1:
        ld = s2
    elif n > 1:
        ld = s0
    else:
        ld = c
    if not ((c == 1 or s1 == 1) and (w == 1 or s3 == ld) and (h == 1 or s2 == w * ld) and (n == 1 or s0 == h * w * ld) and ld >= c):
        raise SegHieroHipError(f"tensor is not NHWC-strided: shape {tuple(t.shape)} strides {t.stride()}")
    return t.data_ptr(), ld, 1 if dt is _BF16 else 0


def is_nhwc(t):
    try:
        pm(t)
        return True
    except SegHieroHipError:
        return False


def to_nhwc(t, cpad=None):
    """Any-layout logical NCHW fp32 tensor -> NHWC-memory tensor (copy only when needed)."""
    _require_gpu(t)
    n, c, h, w = t.shape
    cpad = c if cpad is None else cpad
    if is_nhwc(t) and pm(t)[1] >= cpad:
        return t
    if t.dtype != torch.float32:
        raise SegHieroHipError("fp32 only")
    out = new_act(n, c, h, w, t.device, ld=cpad, zero=cpad != c)
    if t.is_contiguous():
        _call("sh_nchw_to_nhwc", t.data_ptr(), out.data_ptr(), n, c, h, w, cpad, _st())
    else:
        out.copy_(t)       # exotic strides: let torch do the gather (never on the hot path)
    return out


# ----------------------------------------------------------------------------- conv
def _ckey(n, h, w, cin, o, kh, stride, dil):
    if _PROF is None:
        return None
    return f"{n}x{h}x{w} {cin}->{o} k{kh}" + (f" s{stride}" if stride != 1 else "") + (f" d{dil}" if dil != 1 else "")


def conv_out_hw(h, w, kh, kw, stride, pad, dil):
    return (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1, (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1


def w_ohwi(weight):
    """[O,I,KH,KW] parameter -> pointer to OHWI memory (the parameter is kept in channels_last)."""
    o, i, kh, kw = weight.shape
    if kh * kw == 1 or i == 1:
        if not weight.is_contiguous() and not weight.is_contiguous(memory_format=torch.channels_last):
            raise SegHieroHipError("weight must be dense")
        return weight
    if not weight.is_contiguous(memory_format=torch.channels_last):
        raise SegHieroHipError("conv weight must be in channels_last (OHWI) memory; call module.to_native_layout()")
    return weight


UNSUPPORTED = -3          # SH_EUNSUPPORTED: a fused entry point has no instantiation for the geometry (nothing launched)


def _call_fused(name, *args, cost=None, key=None):
    """Fused entry points: True = launched, False = SH_EUNSUPPORTED (the caller runs the unfused sequence); raises otherwise."""
    if _PROF is None:
        rc = LIB.raw(name)(*args)
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = LIB.raw(name)(*args)
        e1.record()
        if rc == 0:
            _PROF.append((name, cost, e0, e1, key))
    if rc == 0:
        return True
    if rc == UNSUPPORTED:
        return False
    raise SegHieroHipError(f"{name} failed with status {rc} ({status_text(rc)})")


def conv_fprop_aff(x, in_coefs, weight, bias, y, partials, stride, pad, dil):
    """y = conv(relu(x * scale + shift), weight): the producer's train-mode BatchNorm + ReLU applied in the conv's loader
    (in_coefs = the (4, C) coefficient tensor of bn_finalize).  -> False when the geometry has no fused kernel."""
    if CONV_IMPL != "x6":
        return False
    n, cin, h, w = x.shape
    o, _, kh, kw = weight.shape
    xp, ldx, xb = pmx(x)
    yp, ldy, yb = pmx(y)
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    m = n * ho * wo
    if _B16 and xb and _fprop_b16(x, in_coefs, weight, bias, y, partials, stride, pad, dil):
        return True
    cost = (2.0 * m * o * cin * kh * kw, 4.0 * (n * h * w * cin + m * o + o * cin * kh * kw))
    ws, nb = _splitk_ws(0, n, h, w, cin, o, kh, kw, stride, pad, dil, 0, x.device)
    return _call_fused("sh_conv_fprop_x6_aff", xp, ldx, _rp(in_coefs, 2), _rp(in_coefs, 3), w_ohwi(weight).data_ptr(),
                       None if bias is None else bias.data_ptr(), yp, ldy, None if partials is None else partials.data_ptr(),
                       n, h, w, cin, o, kh, kw, stride, pad, dil, ws, nb, xb | (yb << 1), _st(), cost=cost,
                       key=_ckey(n, h, w, cin, o, kh, stride, dil))


def _bcost(n, h, w, cin, o, kh, kw, m, xbytes=2, ybytes=2):
    return (2.0 * m * o * cin * kh * kw, float(xbytes * n * h * w * cin + ybytes * m * o + 2 * o * cin * kh * kw))


def _fprop_b16(x, in_coefs, weight, bias, y, partials, stride, pad, dil):
    """bf16 compute mode forward (x a bf16 tensor, read as is or through in_coefs); False = no instantiation (nothing launched)."""
    n, cin, h, w = x.shape
    o, _, kh, kw = weight.shape
    xp, ldx, _ = pmx(x)
    yp, ldy, yb = pmx(y)
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    m = n * ho * wo
    if o % 8:
        # the classifier: columns to pad8(o) (see sh_conv_fprop_b16) -- a padded row stride, no statistics, the bias zero-padded
        if ldy < pad8(o) or partials is not None:
            return False
        if bias is not None:
            bias = torch.cat([bias.detach(), bias.new_zeros(pad8(o) - o)])
    ws, nb = _splitk_ws(0, n, h, w, cin, o, kh, kw, stride, pad, dil, 0, x.device)
    wb, _ = weights_bf16(weight)
    return _call_fused("sh_conv_fprop_b16", xp, ldx, None if in_coefs is None else _rp(in_coefs, 2),
                       None if in_coefs is None else _rp(in_coefs, 3), wb.data_ptr(), None if bias is None else bias.data_ptr(),
                       yp, ldy, None if partials is None else partials.data_ptr(), n, h, w, cin, o, kh, kw, stride, pad, dil, ws, nb,
                       yb << 1, _st(), cost=_bcost(n, h, w, cin, o, kh, kw, m, 2, 2 if yb else 4), key=_ckey(n, h, w, cin, o, kh, stride, dil))


def conv_fprop(x, weight, bias, y, partials, stride, pad, dil):
    n, cin, h, w = x.shape
    o, _, kh, kw = weight.shape
    xp, ldx, xb = pmx(x)
    yp, ldy, yb = pmx(y)
    if _B16 and xb and _fprop_b16(x, None, weight, bias, y, partials, stride, pad, dil):
        return
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    m = n * ho * wo
    x6 = CONV_IMPL == "x6"
    if (xb or yb) and not x6:
        raise SegHieroHipError("bf16-stored activations need the x6 convolution path")
    wptr = w_ohwi(weight).data_ptr()
    cost = (2.0 * m * o * cin * kh * kw, 4.0 * (n * h * w * cin + m * o + o * cin * kh * kw))
    args = (xp, ldx, wptr, None if bias is None else bias.data_ptr(), yp, ldy,
            None if partials is None else partials.data_ptr(), n, h, w, cin, o, kh, kw, stride, pad, dil)
    if x6:
        ws, nb = _splitk_ws(0, n, h, w, cin, o, kh, kw, stride, pad, dil, 0, x.device)
        _call("sh_conv_fprop_x6", *args, ws, nb, xb | (yb << 1), _st(), cost=cost, key=_ckey(n, h, w, cin, o, kh, stride, dil))
    else:
        _call("sh_conv_fprop", *args, _st(), cost=cost, key=_ckey(n, h, w, cin, o, kh, stride, dil))


SPLIT_K = os.environ.get("SEGHIERO_SPLITK", "1") != "0"      # debugging knob: 0 = never hand the conv kernels a split-K workspace


FUSE_EVAL = os.environ.get("SEGHIERO_FUSE_EVAL", "1") != "0"      # inference: BN (+residual) (+ReLU) in the conv epilogue


def conv_fprop_act(x, weight, coefs, out, relu, residual, stride, pad, dil):
    """out = [relu](conv(x, weight) * scale + shift [+ residual]) in one kernel (eval-mode BN folded into the epilogue)."""
    n, cin, h, w = x.shape
    o, _, kh, kw = weight.shape
    xp, ldx = pm(x)
    op, ldo = pm(out)
    rp, ldr = (None, 0) if residual is None else pm(residual)
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    m = n * ho * wo
    _call("sh_conv_fprop_x6_act", xp, ldx, w_ohwi(weight).data_ptr(), _rp(coefs, 2), _rp(coefs, 3), rp, ldr,
          int(bool(relu)), op, ldo, n, h, w, cin, o, kh, kw, stride, pad, dil, _st(),
          cost=(2.0 * m * o * cin * kh * kw, 4.0 * (n * h * w * cin + m * o + o * cin * kh * kw)))


def _splitk_ws(which, n, h, w, cin, o, kh, kw, stride, pad, dil, mode, device):
    """(pointer, bytes) of the split-K slab workspace the x6 fprop / dgrad may use for this shape; (None, 0) = no split."""
    if not SPLIT_K:
        return None, 0
    need = LIB.raw("sh_conv_x6_workspace")(which, n, h, w, cin, o, kh, kw, stride, pad, dil, mode)
    if need <= 0:
        return None, 0
    return workspace(need, device, "splitk").data_ptr(), need


_WT_ACTIVE = None         # {id(weight): (weight, transposed copy)} of the training step in flight, else None


def weight_transpose(weight):
    """[O,I,KH,KW] (OHWI memory) -> [KH*KW, I, pad4(O)] buffer: the K-contiguous B operand of the x6 dgrad."""
    if _WT_ACTIVE is not None:
        hit = _WT_ACTIVE.get(id(weight))
        if hit is not None and hit[0] is weight:
            return hit[1]
    o, i, kh, kw = weight.shape
    wt = torch.empty((kh * kw, i, pad4(o)), device=weight.device, dtype=torch.float32)
    _call("sh_weight_transpose", w_ohwi(weight).data_ptr(), wt.data_ptr(), o, kh, kw, i, _st())
    return wt


def prepare_dgrad_weights(weights, cache):
    """Transpose all of `weights` (the dense conv weights a backward pass will run dgrad on) in a few launches instead of one
    per layer.  `cache` is a dict owned by the caller (the trainer) that keeps the buffers across steps.  The copies are
    handed out by weight_transpose() until release_dgrad_weights(); the caller guarantees that the weights do not change in
    between (the training step: prepare -> forward -> backward -> release -> SGD)."""
    global _WT_ACTIVE
    # The launch tables (pointer / shape arrays) are kept with the buffers: the step boundary is the one place where the GPU queue is
    # empty, so host time spent here is idle device time (measured: 250 us before this launch, tools/gap_analysis.sh).  They are rebuilt
    # when any weight's storage has moved.
    sig = tuple(w.data_ptr() for w in weights)
    plan = cache.get("__plan__")
    if plan is None or plan[0] != sig:
        todo = []
        for w in weights:
            hit = cache.get(id(w))
            if hit is None or hit[0] is not w or hit[1].device != w.device:
                o, i, kh, kw = w.shape
                hit = (w, torch.empty((kh * kw, i, pad4(o)), device=w.device, dtype=torch.float32))
                cache[id(w)] = hit
            todo.append(hit)
        cap = 40                                                   # SH_WT_MAX
        launches = []
        for a in range(0, len(todo), cap):
            chunk = todo[a:a + cap]
            k = len(chunk)
            launches.append((k, (ctypes.c_void_p * k)(*[w_ohwi(w).data_ptr() for w, _ in chunk]), (ctypes.c_void_p * k)(*[t.data_ptr() for _, t in chunk]),
                             (ctypes.c_int * k)(*[w.shape[0] for w, _ in chunk]), (ctypes.c_int * k)(*[w.shape[2] * w.shape[3] for w, _ in chunk]),
                             (ctypes.c_int * k)(*[w.shape[1] for w, _ in chunk])))
        plan = cache["__plan__"] = (sig, launches)
    st = _st()
    for k, wa, ta, co, tp, ci in plan[1]:
        _call("sh_weight_transpose_multi", k, wa, ta, co, tp, ci, st)
    _WT_ACTIVE = cache


def release_dgrad_weights():
    global _WT_ACTIVE, _WB_ACTIVE
    _WT_ACTIVE = None
    _WB_ACTIVE = None


_WB_ACTIVE = None         # {id(weight): (weight, bf16 forward copy, bf16 transposed copy)} of the training step in flight, else None


def _wb_buffers(w):
    o, i, kh, kw = w.shape
    return (torch.empty((o, kh * kw, i), device=w.device, dtype=torch.bfloat16),
            torch.empty((kh * kw, i, pad8(o)), device=w.device, dtype=torch.bfloat16))


def _wb_tables(items):
    cap = 40                                                   # SH_WT_MAX
    out = []
    for a in range(0, len(items), cap):
        chunk = items[a:a + cap]
        k = len(chunk)
        vp = ctypes.c_void_p
        out.append((k, (vp * k)(*[w_ohwi(w).data_ptr() for w, _, _ in chunk]), (vp * k)(*[b.data_ptr() for _, b, _ in chunk]),
                    (vp * k)(*[t.data_ptr() for _, _, t in chunk]), (ctypes.c_int * k)(*[w.shape[0] for w, _, _ in chunk]),
                    (ctypes.c_int * k)(*[w.shape[2] * w.shape[3] for w, _, _ in chunk]), (ctypes.c_int * k)(*[w.shape[1] for w, _, _ in chunk])))
    return out


def _wb_launch(items):
    st = _st()
    for args in _wb_tables(items):
        _call("sh_weights_to_bf16_multi", *args, st)


def prepare_bf16_weights(weights, cache):
    """bf16 compute mode: the bf16 copies (forward operand and transposed input-gradient operand) of all dense conv `weights`, made
    once per step in a few launches; handed out by weights_bf16() until release_dgrad_weights().  Contract as prepare_dgrad_weights."""
    global _WB_ACTIVE
    sig = tuple(w.data_ptr() for w in weights)                 # (launch tables kept across steps: see prepare_dgrad_weights)
    plan = cache.get("__plan__")
    if plan is None or plan[0] != sig:
        todo = []
        for w in weights:
            hit = cache.get(id(w))
            if hit is None or hit[0] is not w or hit[1].device != w.device:
                hit = (w,) + _wb_buffers(w)
                cache[id(w)] = hit
            todo.append(hit)
        plan = cache["__plan__"] = (sig, _wb_tables(todo))
    st = _st()
    for args in plan[1]:
        _call("sh_weights_to_bf16_multi", *args, st)
    _WB_ACTIVE = cache


def weights_bf16(weight):
    """-> (bf16 [O][taps][I] forward copy, bf16 [taps][I][pad8(O)] transposed copy) of a dense conv weight."""
    if _WB_ACTIVE is not None:
        hit = _WB_ACTIVE.get(id(weight))
        if hit is not None and hit[0] is weight:
            return hit[1], hit[2]
    item = (weight,) + _wb_buffers(weight)
    _wb_launch([item])
    return item[1], item[2]


def _dgrad_b16(dy, weight, dx, stride, pad, dil, addend=None, lin=None, bnb=None, scatter=False):
    """bf16 compute mode input gradient.  dy: fp32 or bf16 tensor (lin = (y, lin coefficients): dy is the masked gradient g, bf16);
    bnb = (y_prev, coefs, relu, partials, out_prev or None): BatchNorm-backward epilogue.  Strided convs (no hooks): stride-2 KxK, and
    scatter=True -- a 1x1 strided conv whose result is added to dx at the strided pixels.  False = no instantiation."""
    n, cin, h, w = dx.shape
    o, _, kh, kw = weight.shape
    dyp, lddy, dyb = pmx(dy)
    if lddy < pad8(o):
        return False
    if (stride != 1 or scatter) and (addend is not None or lin is not None or bnb is not None):
        return False
    dxp, lddx, dxb = pmx(dx)
    ap, lda, ab = (None, 0, 0) if addend is None else pmx(addend)
    flags = dyb | (dxb << 1) | (ab << 2) | (64 if scatter else 0)
    ylp, ldyl, linp = None, 0, None
    if lin is not None:
        ylp, ldyl, ylb = pmx(lin[0])
        if not (ylb and dyb):
            return False
        linp = lin[1].data_ptr()
    bargs = (None, 0, None, 0, None, None, None, None, 0, None)
    if bnb is not None:
        y_prev, cf, relu, partials, out_prev = bnb
        ypp, ldyp, ypb = pmx(y_prev)
        flags |= ypb << 3
        if out_prev is not None and out_prev.dtype == torch.uint8:
            opp, ldop = out_prev.data_ptr(), cin // 4
            flags |= 32
        elif out_prev is not None:
            opp, ldop, opb = pmx(out_prev)
            flags |= opb << 4
        else:
            opp, ldop = None, 0
        bargs = (ypp, ldyp, opp, ldop, _rp(cf, 0), _rp(cf, 1), _rp(cf, 2), _rp(cf, 3), int(bool(relu)), partials.data_ptr())
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    m = n * ho * wo
    ws, nb = _splitk_ws(1, n, h, w, cin, o, kh, kw, stride, pad, dil, 0, dx.device)
    _, wtb = weights_bf16(weight)
    cost = (2.0 * m * o * cin * kh * kw, float((2 if dxb else 4) * n * h * w * cin * (1 + (addend is not None)) + (2 if dyb else 4) * m * o * (2 if lin else 1)
                                                + 2 * o * cin * kh * kw + (2 * n * h * w * cin if bnb else 0)))
    return _call_fused("sh_conv_dgrad_b16", dyp, lddy, ylp, ldyl, linp, wtb.data_ptr(), ap, lda, dxp, lddx, *bargs, n, h, w, cin, o, kh, kw,
                       stride, pad, dil, ws, nb, flags, _st(), cost=cost, key=_ckey(n, h, w, cin, o, kh, stride, dil))


def conv_dgrad(dy, weight, dx, stride, pad, dil, addend=None, mode=0):
    n, cin, h, w = dx.shape
    o, _, kh, kw = weight.shape
    if _B16 and mode == 0 and _dgrad_b16(dy, weight, dx, stride, pad, dil, addend=addend):
        return
    if _B16 and mode == 1 and addend is None and _dgrad_b16(dy, weight, dx, stride, pad, dil, scatter=True):
        return
    if dx.dtype != torch.float32:
        raise SegHieroHipError("the fp32-accurate input-gradient kernels write fp32 tensors")
    dy, addend = f32(dy), f32(addend)
    dyp, lddy = pm(dy)
    dxp, lddx = pm(dx)
    ap, lda = (None, 0) if addend is None else pm(addend)
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    m = n * ho * wo
    x6 = CONV_IMPL == "x6" and lddy >= pad4(o)
    wptr = weight_transpose(weight).data_ptr() if x6 else w_ohwi(weight).data_ptr()
    cost = (2.0 * m * o * cin * kh * kw, 4.0 * (n * h * w * cin * (2 if addend is not None or mode else 1) + m * o + o * cin * kh * kw))
    args = (dyp, lddy, wptr, ap, lda, dxp, lddx, n, h, w, cin, o, kh, kw, stride, pad, dil, mode)
    if x6:
        ws, nb = _splitk_ws(1, n, h, w, cin, o, kh, kw, stride, pad, dil, mode, dx.device)
        _call("sh_conv_dgrad_x6", *args, ws, nb, _st(), cost=cost, key=_ckey(n, h, w, cin, o, kh, stride, dil))
    else:
        _call("sh_conv_dgrad", *args, _st(), cost=cost, key=_ckey(n, h, w, cin, o, kh, stride, dil))


def conv_dgrad_bnb(dy, weight, g, y_prev, coefs, relu, partials, stride, pad, dil, addend=None, out_prev=None):
    """Input gradient + front half of the producer layer's BatchNorm backward in the dgrad epilogue: g <- relumask * (dx [+ addend]),
    partials[ceil(M/64), 2, Cin] <- (sum g, sum g*xhat) per 64 rows.  -> False when the geometry has no fused kernel."""
    if CONV_IMPL != "x6":
        return False
    n, cin, h, w = g.shape
    o, _, kh, kw = weight.shape
    if _B16 and _dgrad_b16(dy, weight, g, stride, pad, dil, addend=addend, bnb=(y_prev, coefs, relu, partials, out_prev)):
        return True
    if g.dtype != torch.float32:
        return False
    dy, addend = f32(dy), f32(addend)
    dyp, lddy = pm(dy)
    gp, ldg = pm(g)
    ypp, ldyp, ypb = pmx(y_prev)
    if lddy < pad4(o):
        return False
    ap, lda = (None, 0) if addend is None else pm(addend)
    if out_prev is not None and out_prev.dtype == torch.uint8:      # ReLU quad mask of the previous block's output (new_relu_mask)
        opp, ldop, opb = out_prev.data_ptr(), cin // 4, 2
    else:
        opp, ldop, opb = (None, 0, 0) if out_prev is None else pmx(out_prev)
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    m = n * ho * wo
    cost = (2.0 * m * o * cin * kh * kw, 4.0 * (n * h * w * cin * (3 if addend is not None else 2) + m * o + o * cin * kh * kw))
    ws, nb = _splitk_ws(1, n, h, w, cin, o, kh, kw, stride, pad, dil, 0, g.device)
    return _call_fused("sh_conv_dgrad_x6_bnb", dyp, lddy, weight_transpose(weight).data_ptr(), ap, lda, gp, ldg, ypp, ldyp, opp, ldop,
                       _rp(coefs, 0), _rp(coefs, 1), _rp(coefs, 2), _rp(coefs, 3), int(bool(relu)),
                       partials.data_ptr(), n, h, w, cin, o, kh, kw, stride, pad, dil, ws, nb, ypb | (opb << 1), _st(), cost=cost,
                       key=_ckey(n, h, w, cin, o, kh, stride, dil))


def conv_dgrad_lin(dd, weight, dx, addend=None, bnb=None):
    """1x1 stride-1 input gradient with the dy operand evaluated in the loader from a DeferredDy.  bnb = (y_prev, coefs, partials):
    BatchNorm-backward epilogue for the producer of the conv's input (as conv_dgrad_bnb; dx then receives g).  -> False when the
    geometry has no fused kernel (the caller materialises dy)."""
    if CONV_IMPL != "x6":
        return False
    n, cin, h, w = dx.shape
    o = weight.shape[0]
    if _B16 and dd.g.dtype == torch.bfloat16:
        return _dgrad_b16(dd.g, weight, dx, 1, 0, 1, addend=addend, lin=(dd.y, dd.lin),
                          bnb=None if bnb is None else (bnb[0], bnb[1], True, bnb[2], None))
    if dx.dtype != torch.float32 or dd.g.dtype != torch.float32:
        return False
    addend = f32(addend)
    gp, ldg = pm(dd.g)
    yp, ldy, yb = pmx(dd.y)
    dxp, lddx = pm(dx)
    ap, lda = (None, 0) if addend is None else pm(addend)
    ypb = 0
    if bnb is None:
        bargs = (None, 0, None, None, None, None, 0, None)
    else:
        y_prev, cf, partials = bnb
        ypp, ldyp, ypb = pmx(y_prev)
        bargs = (ypp, ldyp, _rp(cf, 0), _rp(cf, 1), _rp(cf, 2), _rp(cf, 3), 1, partials.data_ptr())
    m = n * h * w
    cost = (2.0 * m * o * cin, 4.0 * (m * cin * (1 + (addend is not None) + (bnb is not None)) + 2 * m * o + o * cin))
    ws, nb = _splitk_ws(1, n, h, w, cin, o, 1, 1, 1, 0, 1, 0, dx.device)
    return _call_fused("sh_conv_dgrad_x6_lin", gp, ldg, yp, ldy, dd.lin.data_ptr(), weight_transpose(weight).data_ptr(), ap, lda, dxp, lddx,
                       *bargs, n, h, w, cin, o, ws, nb, yb | (ypb << 1), _st(), cost=cost, key=_ckey(n, h, w, cin, o, 1, 1, 1))


_WS = {}


def workspace(nbytes, device, tag="ws"):
    """Grow-only scratch buffer per (device, tag, launch stream): kernels of one stream run in order, so a buffer is never
    shared by kernels that may overlap (the weight-gradient stream gets its own)."""
    key = (device, tag, torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
           if device.type == "cuda" else 0)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# Weight gradients run on a side HIP stream: dgrad (the critical path of backward) and the wgrad of the same layer only share
# read-only inputs, so the two kernels can co-reside on a CU (98 KB + 61 KB of LDS) and fill each other's load / split /
# epilogue phases and grid tails.  Every hand-scheduled backward node joins the side stream before it returns its gradients
# (layers.GradMap.ordered / flush), so consumers of p.grad are ordered after the wgrads without knowing about the stream.
WGRAD_ASYNC = os.environ.get("SEGHIERO_WGRAD_STREAM", "1") != "0"
WGRAD_NSTREAMS = int(os.environ.get("SEGHIERO_WGRAD_NSTREAMS", "1"))
WGRAD_AFTER_DGRAD = os.environ.get("SEGHIERO_WGRAD_AFTER_DGRAD", "1") != "0"
_WG_STREAMS = {}          # device index -> [[streams], [pending flags], next]


class _WgEnt:
    def __init__(self, device):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(1, WGRAD_NSTREAMS))]
        self.pending = [False] * len(self.streams)
        self.next = 0

    def take(self):
        k = self.next
        self.next = (k + 1) % len(self.streams)
        self.pending[k] = True
        return k, self.streams[k]


def _wgrad_side(device):
    ent = _WG_STREAMS.get(device.index)
    if ent is None:
        ent = _WgEnt(device)
        _WG_STREAMS[device.index] = ent
    return ent


def _launch_on_side(device, launch, tagfmt, tensors):
    """launch(tag) with the weight-gradient side stream as the launch stream of its C-ABI calls (ops._STREAM: the kernels take their
    stream as an argument, so torch's current stream is not switched -- a `with torch.cuda.stream(...)` costs more host time than the
    launch itself); the side stream first waits for what the compute stream has queued, and the allocator is told about the tensors."""
    global _STREAM
    k, st = _wgrad_side(device).take()
    st.wait_stream(torch.cuda.current_stream(device))     # dy / x were produced on the compute stream
    prev, _STREAM = _STREAM, st.cuda_stream
    try:
        launch(tagfmt % k)
    finally:
        _STREAM = prev
    for t in tensors:
        t.record_stream(st)                                 # the allocator must not recycle them under the side stream



def join_wgrad():
    """Make the current stream wait for every weight gradient launched on the side stream(s) so far."""
    for idx, ent in _WG_STREAMS.items():
        for k, st in enumerate(ent.streams):
            if ent.pending[k]:
                torch.cuda.current_stream(idx).wait_stream(st)
                ent.pending[k] = False


def wgrad_aff_ok(x, dweight, stride, pad, dil):
    """Can conv_wgrad(..., aff=...) read this x through the producer's BatchNorm + ReLU?  (x6 kernel, output width >= 16)"""
    n, cin, h, w = x.shape
    o, _, kh, kw = dweight.shape
    return CONV_IMPL == "x6" and _wgrad_pipelined_ok(x, o, kh, kw, stride, pad, dil)


def _wgrad_pipelined_ok(x, o, kh, kw, stride, pad, dil):
    """Does the pipelined weight-gradient kernel (the only one with the BatchNorm loader / bf16 streams) take this geometry?
    16-byte channel chunks, output width >= 16, and both operands below the 2 GiB range of its 32-bit buffer offsets -- a
    256-channel conv3 gradient at 512^2 crosses that at batch 128, which fits the 288 GB part."""
    n, cin, h, w = x.shape
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    ldx = _pm(x, True)[1] if x.dim() == 4 else cin
    return (cin % 4 == 0 and wo >= 16 and n * h * w * ldx * 4 < (1 << 31) and n * ho * wo * pad4(o) * 4 < (1 << 31))


def conv_wgrad(x, dy, dweight, stride, pad, dil, side=False, aff=None):
    """dweight <- grad_weight.  side=True (the backward nodes): launched on the weight-gradient stream; the caller must
    run join_wgrad() before anything consumes dweight.  aff = (4, C) BatchNorm coefficients: x holds the producer conv's raw
    output and the loader applies relu(x * scale + shift) (check wgrad_aff_ok first)."""
    n, cin, h, w = x.shape
    o, _, kh, kw = dweight.shape
    dd = dy if isinstance(dy, DeferredDy) else None          # dy evaluated in the loader (check lin_ok before deferring)
    if _B16 and x.dtype == torch.bfloat16 and _wgrad_b16(x, dy, dweight, stride, pad, dil, side, aff):
        return
    if dd is not None and dd.g.dtype != torch.float32:
        dy, dd = dd.materialize(), None                      # the fp32-accurate lin loaders take an fp32 g
    if dd is None:
        dy = f32(dy)
    if (x.dtype == torch.bfloat16 or dd is not None) and not (CONV_IMPL == "x6" and _wgrad_pipelined_ok(x, o, kh, kw, stride, pad, dil)):
        # bf16-stored x / deferred dy where the pipelined wgrad has no instantiation (output width < 16: tiny test inputs; operands
        # of 2 GiB or more): widen / materialise once and run the plain kernel
        if x.dtype == torch.bfloat16:
            x = x.float()
        if dd is not None:
            dy, dd = dd.materialize(), None
    if dd is not None and (dd.y.dtype == torch.bfloat16) != (x.dtype == torch.bfloat16):
        dy, dd = dd.materialize(), None                      # the kernel takes both stored streams in one element type
    if dd is not None:
        dy = dd.g
    # tiny output-channel counts (cls_seg, aux head) stay on the f32-MFMA kernel
    x6 = CONV_IMPL == "x6" and o >= 32 and (cin * kh * kw >= 128 or (cin * kh * kw == 64 and o >= 128))
    if aff is not None or dd is not None or x.dtype == torch.bfloat16:
        x6 = True
    need = LIB.raw("sh_conv_wgrad_x6_workspace" if x6 else "sh_conv_wgrad_workspace")(n, h, w, cin, o, kh, kw, stride, pad, dil)
    if need < 0:
        raise SegHieroHipError("sh_conv_wgrad_workspace rejected the geometry")
    xp, ldx, xb = pmx(x)
    dyp, lddy = pm(dy)
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    m = n * ho * wo
    cost = (2.0 * m * o * cin * kh * kw, 4.0 * (n * h * w * cin + m * o + o * cin * kh * kw))
    name = "sh_conv_wgrad_x6" if x6 else "sh_conv_wgrad"

    def launch(tag="wgrad"):
        ws = workspace(need, x.device, tag)              # one workspace per stream: its kernels stay in that stream's order
        if dd is not None:
            y2p, ldy2, y2b = pmx(dd.y)
            if not _call_fused("sh_conv_wgrad_x6_lin", xp, ldx, None if aff is None else _rp(aff, 2), None if aff is None else _rp(aff, 3),
                               dyp, lddy, y2p, ldy2, dd.lin.data_ptr(), dweight.data_ptr(), ws.data_ptr(), n, h, w, cin, o, kh, kw, stride, pad, dil,
                               xb | (y2b << 1), _st(), cost=cost, key=_ckey(n, h, w, cin, o, kh, stride, dil)):
                raise SegHieroHipError("sh_conv_wgrad_x6_lin: unsupported geometry (check lin_ok before deferring the BatchNorm-backward apply)")
            return
        if aff is not None:
            if not _call_fused("sh_conv_wgrad_x6_aff", xp, ldx, _rp(aff, 2), _rp(aff, 3), dyp, lddy, dweight.data_ptr(),
                               ws.data_ptr(), n, h, w, cin, o, kh, kw, stride, pad, dil, xb, _st(), cost=cost,
                               key=_ckey(n, h, w, cin, o, kh, stride, dil)):
                raise SegHieroHipError("sh_conv_wgrad_x6_aff: unsupported geometry (check wgrad_aff_ok before deferring the activation)")
            return
        _call(name, xp, ldx, dyp, lddy, dweight.data_ptr(), ws.data_ptr(), n, h, w, cin, o, kh, kw, stride, pad, dil, *((xb,) if x6 else ()), _st(),
              cost=cost, key=_ckey(n, h, w, cin, o, kh, stride, dil))

    if not (side and WGRAD_ASYNC and x.is_cuda):
        launch()
        return
    _launch_on_side(x.device, launch, "wgrad%d", (x, dy, dweight) + (() if dd is None else (dd.y, dd.lin)))


def _wgrad_b16(x, dy, dweight, stride, pad, dil, side, aff):
    """bf16 compute mode weight gradient (x bf16; dy fp32 / bf16 tensor, or a DeferredDy whose g and y are bf16).  False = no
    instantiation (nothing launched, the caller runs the fp32-accurate kernel)."""
    n, cin, h, w = x.shape
    o, _, kh, kw = dweight.shape
    dd = dy if isinstance(dy, DeferredDy) else None
    g = dd.g if dd is not None else dy
    if dd is not None and not (dd.g.dtype == torch.bfloat16 and dd.y.dtype == torch.bfloat16):
        return False
    if cin % 8 or not dweight.is_contiguous(memory_format=torch.channels_last if kh * kw > 1 and cin > 1 else torch.contiguous_format):
        return False
    need = LIB.raw("sh_conv_wgrad_x6_workspace")(n, h, w, cin, o, kh, kw, stride, pad, dil)
    if need < 0:
        return False
    xp, ldx, _ = pmx(x)
    gp, ldg, gb = pmx(g)
    if ldg < pad8(o) or (o % 8 and dd is not None):
        return False                                   # the loader reads 8 channels of dy at a time (padding lanes zero)
    ylp, ldyl = (None, 0) if dd is None else pmx(dd.y)[:2]
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    m = n * ho * wo
    cost = (2.0 * m * o * cin * kh * kw, float(2 * n * h * w * cin + (2 if gb else 4) * m * o * (2 if dd is not None else 1) + 4 * o * cin * kh * kw))
    ok = [True]

    def launch(tag):
        ws = workspace(need, x.device, tag)
        ok[0] = _call_fused("sh_conv_wgrad_b16", xp, ldx, None if aff is None else _rp(aff, 2), None if aff is None else _rp(aff, 3),
                            gp, ldg, ylp, ldyl, None if dd is None else dd.lin.data_ptr(), dweight.data_ptr(), ws.data_ptr(), n, h, w, cin, o,
                            kh, kw, stride, pad, dil, gb, _st(), cost=cost, key=_ckey(n, h, w, cin, o, kh, stride, dil))

    if not (side and WGRAD_ASYNC and x.is_cuda):
        launch("wgrad")
        return ok[0]
    _launch_on_side(x.device, launch, "wgrad%d", (x, g, dweight) + (() if dd is None else (dd.y, dd.lin)))
    return ok[0]


def conv_partials(m, cout, device):
    return torch.empty((-(-m // 64), 2, cout), device=device, dtype=torch.float32)


# ----------------------------------------------------------------------------- depthwise
def dw_partials_rows(n, h, w):
    return LIB.raw("sh_dw_partials")(n, h, w)


def dwconv_fprop(x, weight, y, partials, dil, aff=None):
    """aff = (4, C) BatchNorm coefficients: x is the producer conv's raw output, read as relu(x * scale + shift)."""
    n, c, h, w = x.shape
    xp, ldx, xb = pmx(x)
    yp, ldy, yb = pmx(y)
    _call("sh_dwconv_fprop", xp, ldx, None if aff is None else _rp(aff, 2), None if aff is None else _rp(aff, 3),
          weight.data_ptr(), yp, ldy, None if partials is None else partials.data_ptr(), n, h, w, c, dil, xb | (yb << 1), _st(),
          key=(None if _PROF is None else f"{n}x{h}x{w} C{c} d{dil}"))


def dw_lin_ok(shape, dil):
    """Can the depthwise dgrad / wgrad read their dy operand as a DeferredDy?  (strip-walk kernels: dilation 1, H and W multiples of 8)"""
    n, c, h, w = shape
    return DEFER_APPLY and dil == 1 and h % 8 == 0 and w % 8 == 0 and c % 4 == 0


def _lin_args(dy):
    """((g pointer, ldg, y pointer, ldy, lin pointer), y-is-bf16) of a DeferredDy, or ((dy pointer, lddy, None, 0, None), 0) of a tensor."""
    if isinstance(dy, DeferredDy):
        yp, ldy, yb = pmx(dy.y)
        return pm(dy.g) + (yp, ldy, dy.lin.data_ptr()), yb
    return pm(dy) + (None, 0, None), 0


def dwconv_dgrad(dy, weight, dx, dil, accumulate=False):
    """dy: tensor or DeferredDy (check dw_lin_ok first)."""
    n, c, h, w = dx.shape
    dxp, lddx = pm(dx)
    largs, yb = _lin_args(dy)
    _call("sh_dwconv_dgrad", *largs, weight.data_ptr(), dxp, lddx, n, h, w, c, dil, int(accumulate), yb, _st(), key=(None if _PROF is None else f"{n}x{h}x{w} C{c} d{dil}"))


def dwconv_dgrad_bnb(dy, weight, g, y_prev, coefs, partials, dil):
    """depthwise input gradient + front half of the producer's BatchNorm backward: g <- relumask * dx, partials <- (sum g, sum g*xhat)."""
    n, c, h, w = g.shape
    gp, ldg = pm(g)
    ypp, ldyp, ypb = pmx(y_prev)
    largs, yb = _lin_args(dy)
    _call("sh_dwconv_dgrad_bnb", *largs, weight.data_ptr(), gp, ldg, ypp, ldyp, _rp(coefs, 0), _rp(coefs, 1),
          _rp(coefs, 2), _rp(coefs, 3), partials.data_ptr(), n, h, w, c, dil, yb | (ypb << 1), _st(), key=(None if _PROF is None else f"{n}x{h}x{w} C{c} d{dil}"))


def dwconv_wgrad(x, dy, dweight, dil, side=False, aff=None):
    """side=True: on the weight-gradient stream (see conv_wgrad); join_wgrad() before dweight is consumed.
    aff: as dwconv_fprop."""
    n, c, h, w = x.shape
    p = dw_partials_rows(n, h, w)
    xp, ldx, xb = pmx(x)
    largs, yb = _lin_args(dy)
    dd = dy if isinstance(dy, DeferredDy) else None
    if dd is not None:
        dy = dd.g

    def launch(tag="dwwgrad"):
        ws = workspace(p * 9 * c * 4, x.device, tag)
        _call("sh_dwconv_wgrad", xp, ldx, None if aff is None else _rp(aff, 2), None if aff is None else _rp(aff, 3),
              *largs, ws.data_ptr(), dweight.data_ptr(), n, h, w, c, dil, xb | (yb << 1), _st(), key=(None if _PROF is None else f"{n}x{h}x{w} C{c} d{dil}"))

    if not (side and WGRAD_ASYNC and x.is_cuda):
        launch()
        return
    _launch_on_side(x.device, launch, "dwwgrad%d", (x, dy, dweight) + (() if dd is None else (dd.y, dd.lin)))


# ----------------------------------------------------------------------------- batch norm
SYNC_BN = False       # True: BatchNorm statistics (forward and backward) are all-reduced over the default process group


def _sync_on():
    if not SYNC_BN:
        return False
    from . import ddp
    return ddp.collectives_on()


def _all_reduce_sq(sq):
    """Sum the f64 vector [sum, sum-of-squares (or sum g*xhat), local count] over the ranks, on the small-message group."""
    from . import ddp
    ddp.all_reduce_small(sq)
    return sq


FOLD_MIN = int(os.environ.get("SEGHIERO_FOLD_MIN", "4096"))      # partial lists at least this long are folded 64:1 before the finalize
FOLD_CHUNK = 64


def _fold_partials(partials, c, count, rows):
    """(partials', rows') for the finalize kernels: a long list is folded FOLD_CHUNK:1 by sh_bn_fold_partials (same format), else as is."""
    p = partials.shape[0]
    if p < FOLD_MIN or partials.shape[-1] != c or not partials.is_contiguous():
        return partials, rows
    out = torch.empty((-(-p // FOLD_CHUNK), 2, c), device=partials.device, dtype=torch.float32)
    if not _call_fused("sh_bn_fold_partials", partials.data_ptr(), p, c, float(count), rows, FOLD_CHUNK, out.data_ptr(), _st(), key=(None if _PROF is None else f"P{p} C{c}")):
        return partials, rows
    return out, rows * FOLD_CHUNK


def bn_finalize(partials, count, gamma, beta, eps, momentum, running_mean, running_var, c, device, rows=64, coefs=None,
                partials_ld=0, partials_col=0):
    """coefs (optional): (4, c) tensor / strided view to fill.  partials_ld / partials_col: this layer's columns are
    [col, col + c) of a wider partials tensor with rows of partials_ld floats (the grouped ASPP launch)."""
    if coefs is None:
        coefs = torch.empty((4, c), device=device, dtype=torch.float32)      # mean, invstd, scale, shift
    pptr = partials.data_ptr() + 4 * partials_col
    if _sync_on():
        if partials_ld:
            raise SegHieroHipError("SyncBN does not take column slices of grouped partials")
        # sq = [sum x (C), sum x^2 (C), local count]: the count travels with the sums, so ranks may hold different numbers of
        # pixels (uneven last batch, different crops) -- torch.nn.SyncBatchNorm all-gathers the counts for the same reason
        sq = torch.empty((2 * c + 1,), device=device, dtype=torch.float64)
        _call("sh_bn_reduce_partials", partials.data_ptr(), partials.shape[0], c, float(count), rows, sq.data_ptr(), _st())
        _all_reduce_sq(sq)
        _call("sh_bn_finalize_sq", sq.data_ptr(), c, None if gamma is None else gamma.data_ptr(),
              None if beta is None else beta.data_ptr(), eps, momentum,
              None if running_mean is None else running_mean.data_ptr(),
              None if running_var is None else running_var.data_ptr(),
              _rp(coefs, 0), _rp(coefs, 1), _rp(coefs, 2), _rp(coefs, 3), _st())
        return coefs
    if not partials_ld and not partials_col:
        partials, rows = _fold_partials(partials, c, count, rows)
        pptr = partials.data_ptr()
    _call("sh_bn_finalize", pptr, partials.shape[0], c, float(count),
          None if gamma is None else gamma.data_ptr(), None if beta is None else beta.data_ptr(), eps, momentum,
          None if running_mean is None else running_mean.data_ptr(),
          None if running_var is None else running_var.data_ptr(),
          _rp(coefs, 0), _rp(coefs, 1), _rp(coefs, 2), _rp(coefs, 3), rows, int(partials_ld), _st(),
          key=(None if _PROF is None else f"P{partials.shape[0]} C{c}"))
    return coefs


def bn_finalize_scaled(partials, count, dw_weight, gamma, beta, eps, momentum, running_mean, running_var, c, device, rows):
    """BatchNorm of y = w_centre[c] * x from the statistics partials of x (centre-tap depthwise conv, see sh_bn_finalize_scaled).
    -> (coefs (4, c) in the x domain, isy (c,))."""
    coefs = torch.empty((4, c), device=device, dtype=torch.float32)
    isy = torch.empty((c,), device=device, dtype=torch.float32)
    _call("sh_bn_finalize_scaled", partials.data_ptr(), partials.shape[0], c, float(count), dw_weight.data_ptr() + 16, 9,
          gamma.data_ptr(), beta.data_ptr(), eps, momentum, None if running_mean is None else running_mean.data_ptr(),
          None if running_var is None else running_var.data_ptr(), _rp(coefs, 0), _rp(coefs, 1), _rp(coefs, 2),
          _rp(coefs, 3), isy.data_ptr(), rows, _st())
    return coefs, isy


def bn_finalize_multi(partials, count, rows, bns, coefs_list, partials_ld=0, cols=None, dw_weights=None, isy_list=None):
    """k BatchNorm layers (same channel count, eps, momentum) finalized in one launch.  bns: the nn.BatchNorm2d modules; coefs_list:
    (4, C) tensors / views to fill; cols: first partials column of each layer; dw_weights: per layer a depthwise weight [C,1,3,3]
    whose centre tap multiplies the statistics (or None); isy_list: (C,) outputs for those."""
    k = len(bns)
    c = coefs_list[0].shape[1]
    eps, mom = bns[0].eps, 0.1 if bns[0].momentum is None else bns[0].momentum
    if any(b.eps != eps or (0.1 if b.momentum is None else b.momentum) != mom for b in bns) or k > 8:
        raise SegHieroHipError("bn_finalize_multi needs <= 8 layers with one eps / momentum")
    vp = ctypes.c_void_p
    arr = lambda vals: (vp * k)(*vals)
    ptr = lambda t: None if t is None else t.data_ptr()
    cols = [0] * k if cols is None else cols
    dw_weights = [None] * k if dw_weights is None else dw_weights
    isy_list = [None] * k if isy_list is None else isy_list
    _call("sh_bn_finalize_multi", k, partials.data_ptr(), partials.shape[0], c, float(count), rows, int(partials_ld),
          (ctypes.c_int * k)(*cols), arr([b.weight.data_ptr() for b in bns]), arr([b.bias.data_ptr() for b in bns]),
          arr([ptr(b.running_mean) for b in bns]), arr([ptr(b.running_var) for b in bns]),
          arr([_rp(cf, 0) for cf in coefs_list]), arr([_rp(cf, 1) for cf in coefs_list]),
          arr([_rp(cf, 2) for cf in coefs_list]), arr([_rp(cf, 3) for cf in coefs_list]),
          arr([None if w is None else w.data_ptr() + 16 for w in dw_weights]), 9, arr([ptr(t) for t in isy_list]),
          float(eps), float(mom), _st())


def dw_center_wgrad(dgamma, gamma, isy, dw_weight, eps, dweight, g=None, x=None, mean_x=None):
    """g (masked gradient w.r.t. the BatchNorm output), x (the conv input), mean_x: channels whose centre tap is exactly 0 get their
    gradient from these (their dgamma is 0 whatever the true gradient is)."""
    gp, ldg = (None, 0) if g is None else pm(g)
    xp, ldx = (None, 0) if g is None else pm(x)
    m = 0 if g is None else g.shape[0] * g.shape[2] * g.shape[3]
    _call("sh_dw_center_wgrad", dgamma.data_ptr(), gamma.data_ptr(), isy.data_ptr(), dw_weight.data_ptr(), float(eps),
          dweight.data_ptr(), dw_weight.shape[0], gp, ldg, xp, ldx, None if g is None else mean_x.data_ptr(), m, _st())


def conv1x1_grouped_fprop(sources, weights, y, partials):
    """One launch for several pointwise convs of one geometry (the ASPP branches): sources = [(x, coefs or None)], group g writes
    y[:, g*A:(g+1)*A].  -> False when the geometry has no grouped kernel."""
    if CONV_IMPL != "x6":
        return False
    k = len(sources)
    n, cin, h, w = sources[0][0].shape
    a = weights[0].shape[0]
    if _B16 and all(x.dtype == torch.bfloat16 for x, _ in sources):      # bf16 compute mode: every group's input is a bf16 tensor
        xs, lds = zip(*[pmx(x)[:2] for x, _ in sources])
        yp, ldy, yb = pmx(y)
        vp = ctypes.c_void_p
        arr = lambda vals: (vp * k)(*vals)
        cost = (2.0 * n * h * w * cin * a * k, float(n * h * w * (2 * cin + (2 if yb else 4) * a) * k + 2 * a * cin * k))
        wbs = [weights_bf16(wt)[0] for wt in weights]         # (held until the launch is enqueued: copies made on demand are fresh tensors)
        return _call_fused("sh_conv1x1_grouped_fprop_b16", k, arr(xs), (ctypes.c_int * k)(*lds),
                           arr([None if c is None else _rp(c, 2) for _, c in sources]),
                           arr([None if c is None else _rp(c, 3) for _, c in sources]),
                           arr([t.data_ptr() for t in wbs]), yp, ldy, partials.data_ptr(), n, h, w, cin, a, yb << 1, _st(),
                           cost=cost, key=(None if _PROF is None else f"{k} x ({n}x{h}x{w} {cin}->{a} k1)"))
    xs, lds = zip(*[pm(x) for x, _ in sources])
    yp, ldy = pm(y)
    vp = ctypes.c_void_p
    arr = lambda vals: (vp * k)(*vals)
    cost = (2.0 * n * h * w * cin * a * k, 4.0 * (n * h * w * (cin + a) * k + a * cin * k))
    return _call_fused("sh_conv1x1_grouped_fprop_x6", k, arr(xs), (ctypes.c_int * k)(*lds),
                       arr([None if c is None else _rp(c, 2) for _, c in sources]),
                       arr([None if c is None else _rp(c, 3) for _, c in sources]),
                       arr([w_ohwi(wt).data_ptr() for wt in weights]), yp, ldy, partials.data_ptr(), n, h, w, cin, a, _st(),
                       cost=cost, key=(None if _PROF is None else f"{k} x ({n}x{h}x{w} {cin}->{a} k1)"))



def bn_eval_coefs(gamma, beta, running_mean, running_var, eps):
    c = running_mean.numel()
    coefs = torch.empty((4, c), device=running_mean.device, dtype=torch.float32)
    _call("sh_bn_eval_coefs", gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), eps, c,
          _rp(coefs, 2), _rp(coefs, 3), _st())
    return coefs


def channel_stats(y):
    n, c, h, w = y.shape
    m = n * h * w
    p = LIB.raw("sh_stats_partials_count")(m)
    partials = torch.empty((p, 2, c), device=y.device, dtype=torch.float32)
    yp, ldy = pm(y)
    _call("sh_channel_stats", yp, ldy, m, c, partials.data_ptr(), _st())
    return partials


RELU_MASK = os.environ.get("SEGHIERO_RELU_MASK", "1") != "0"      # residual blocks keep a 1/16-size ReLU quad mask of their output for the backward


def new_relu_mask(n, c, h, w, device):
    """Buffer for sh_bn_act's ReLU quad mask of an [n, c, h, w] output: one byte per (pixel, 4 channels)."""
    return torch.empty((n, h, w, c // 4), device=device, dtype=torch.uint8)


def bn_act(y, coefs, out, relu, residual=None, res_coefs=None, mask=None):
    """res_coefs: `residual` is a raw conv output whose BatchNorm (no ReLU) is applied on the fly (the downsample branch).
    mask (new_relu_mask): also receives the ReLU quad mask of `out`."""
    n, c, h, w = y.shape
    yp, ldy, yb = pmx(y)
    op, ldo, ob = pmx(out)
    rp, ldr, rb = (None, 0, 0) if residual is None else pmx(residual)
    _call("sh_bn_act", yp, ldy, _rp(coefs, 2), _rp(coefs, 3), rp, ldr,
          None if res_coefs is None else _rp(res_coefs, 2), None if res_coefs is None else _rp(res_coefs, 3),
          op, ldo, n * h * w, c, int(relu), None if mask is None else mask.data_ptr(), yb | (rb << 1) | (ob << 2), _st(),
          key=(None if _PROF is None else f"{n}x{h}x{w} C{c}" + (" +res" if residual is not None else "")))


class DeferredDy:
    """Gradient w.r.t. a raw conv output, NOT materialised: dy = lin[0]*g + lin[1]*(y - lin[2]) + lin[3] per channel (the second half
    of the BatchNorm backward) of the masked gradient g and the raw conv output y.  The 1x1 consumers evaluate it in their loaders
    (conv_dgrad_lin, conv_wgrad(dy=DeferredDy)); materialize() runs the sh_bn_bwd_apply pass for anything else."""
    __slots__ = ("g", "y", "lin", "coefs", "gamma", "red", "_dy")

    def __init__(self, g, y, lin, coefs, gamma, red):
        self.g, self.y, self.lin, self.coefs, self.gamma, self.red, self._dy = g, y, lin, coefs, gamma, red, None

    @property
    def shape(self):
        return self.y.shape

    def materialize(self):
        if self._dy is None:
            n, c, h, w = self.y.shape
            gdt = self.g.dtype
            ld = pad8(c) if gdt == torch.bfloat16 else pad4(c)
            dy = new_act(n, c, h, w, self.y.device, ld=ld, zero=ld != c, dtype=gdt)
            gp, ldg, gb = pmx(self.g)
            yp, ldy, yb = pmx(self.y)
            dyp, lddy = pmx(dy)[:2]
            cf = self.coefs
            _call("sh_bn_bwd_apply", gp, ldg, None, 0, yp, ldy, _rp(cf, 0), _rp(cf, 1), _rp(cf, 2), _rp(cf, 3),
                  None if self.gamma is None else self.gamma.data_ptr(), _rp(self.red, 2), _rp(self.red, 3), dyp, lddy,
                  None, 0, n * h * w, c, 0, yb | (gb << 2) | (gb << 3), _st(), key=(None if _PROF is None else f"{n}x{h}x{w} C{c} (deferred)"))
            self._dy = dy
        return self._dy


DEFER_APPLY = os.environ.get("SEGHIERO_DEFER_APPLY", "1") != "0"     # BatchNorm-backward apply in the 1x1 consumers' loaders
# The dgrad re-evaluates dy once per 128-column tile of its OUTPUT (Cin / 128 times), the apply pass once: deferring pays where the
# dy tensor is wide and the dgrad output narrow (Bottleneck conv3: 4P -> P), not the other way round (conv1: P -> 4P).  Deferred
# when Cout * DEFER_RATIO >= Cin; measured (same-box A/B, ms per step): always 33.85, ratio 1 (also 512 -> 512) 33.7 / 34.0,
# ratio 0.5 (Cout >= 2 Cin) 33.6 / 33.8 -- the two-stream loader costs the square pointwise convs what the apply pass saved.
DEFER_RATIO = float(os.environ.get("SEGHIERO_DEFER_RATIO", "0.5"))
# bf16 compute mode: a gradient w.r.t. an activation is stored like the activation itself -- bf16 where the forward tensor is bf16
# (trunk and decoder), fp32 elsewhere -- except where its consumer only takes fp32 (depthwise / pooling / resampling kernels, the
# fp32-accurate fallbacks): grad_dtype(like, fp32_consumer)
GRAD_BF16 = os.environ.get("SEGHIERO_GRAD_BF16", "1") != "0"


def grad_dtype(like, fp32_consumer=False):
    if _B16 and GRAD_BF16 and not fp32_consumer and like is not None and like.dtype == torch.bfloat16:
        return torch.bfloat16
    return torch.float32


def strided_dgrad_b16_ok(weight, cin, stride, pad, dil, scatter):
    """bf16 compute mode: does the one-product kernel take this strided input gradient (so that dy and dx may be bf16 tensors)?
    scatter: the 1x1 strided conv whose gradient is added into an existing dx; otherwise stride-2 KxK by parity class."""
    o, _, kh, kw = weight.shape
    if not (_B16 and GRAD_BF16 and CONV_IMPL == "x6" and cin % 8 == 0 and o % 8 == 0):
        return False
    if scatter:
        return kh == 1 and kw == 1 and pad == 0
    return stride == 2 and dil == 1 and kh * kw > 1 and o % 64 == 0


def lin_ok(x_shape, weight, stride, pad, dil):
    """Can the dgrad AND the wgrad of this conv read their dy operand as a DeferredDy?  (x6 kernels, 1x1 stride 1, 16-byte channel
    counts, output width >= 16 for the pipelined wgrad, operands below the 2 GiB buffer-descriptor range)"""
    n, cin, h, w = x_shape
    o, _, kh, kw = weight.shape
    if _B16 and not GRAD_BF16:
        return False                   # bf16 compute mode: lin(g, y) takes a bf16 g (GRAD_BF16); with fp32 gradients dy is materialised
    return (DEFER_APPLY and CONV_IMPL == "x6" and kh == 1 and kw == 1 and stride == 1 and pad == 0 and o % 4 == 0 and cin % 4 == 0
            and o >= 32 and cin >= 64 and w >= 16 and n * h * w * max(pad4(o), pad4(cin)) * 4 < (1 << 31)
            and o * DEFER_RATIO >= cin)


def bn_backward(dout, out, y, coefs, gamma, relu, want_dres=False, dy_ld=None, defer=False, grad32=False):
    """-> (dy, dgamma, dbeta, dres).  relu: 0 none, 1 mask from `out` (residual blocks), 2 mask recomputed from y and
    the forward coefficients (out may be None).  dout may be a layers.GradPack: the ReLU mask is already applied and the
    (sum g, sum g*xhat) partials were produced by the consumer's dgrad epilogue, so the statistics pass is skipped.
    defer=True: dy is returned as a DeferredDy (no apply pass, no dy tensor) -- the statistics pass then also stores the masked
    gradient g where a mask applies.  grad32: the gradient tensors made here are fp32 whatever y's storage (their consumer takes
    fp32 only); otherwise they are stored like y in bf16 compute mode (grad_dtype)."""
    relu = int(relu)
    if relu == 2:
        out = None
    n, c, h, w = y.shape
    m = n * h * w
    dev = y.device
    yp, ldy, yb = pmx(y)
    if out is not None and out.dtype == torch.uint8:          # the ReLU quad mask of the output (new_relu_mask) instead of the output
        if relu != 1:
            raise SegHieroHipError("a ReLU quad mask stands in for `out` in mode 1 only")
        op, ldo, ob, relu = out.data_ptr(), c // 4, 0, 3
    else:
        op, ldo, ob = (None, 0, 0) if out is None else pmx(out)
    af = yb | (ob << 1)
    gdt = grad_dtype(y, grad32)                # storage of the gradient tensors made here (dy, g, dres): like y, or fp32
    packed = hasattr(dout, "partials")
    if packed:
        partials, dout, relu, op, ldo = dout.partials, dout.g, 0, None, 0
        p = partials.shape[0]
        dop, lddo, dob = pmx(dout)
    else:
        dop, lddo, dob = pmx(dout)
        p = LIB.raw("sh_stats_partials_count")(m)
        partials = torch.empty((p, 2, c), device=dev, dtype=torch.float32)
        gmask = None
        if defer and relu:
            gmask = new_act(n, c, h, w, dev, dtype=gdt)
            if not _call_fused("sh_bn_bwd_reduce", dop, lddo, op, ldo, yp, ldy, _rp(coefs, 0), _rp(coefs, 1), _rp(coefs, 2),
                               _rp(coefs, 3), partials.data_ptr(), m, c, relu, gmask.data_ptr(), c, af | (dob << 2) | (int(gdt == torch.bfloat16) << 3),
                               _st(), key=(None if _PROF is None else f"{n}x{h}x{w} C{c} +g")):
                gmask, defer = None, False
        if gmask is None:
            _call("sh_bn_bwd_reduce", dop, lddo, op, ldo, yp, ldy, _rp(coefs, 0), _rp(coefs, 1), _rp(coefs, 2),
                  _rp(coefs, 3), partials.data_ptr(), m, c, relu, None, 0, af | (dob << 2), _st(), key=(None if _PROF is None else f"{n}x{h}x{w} C{c}"))
        else:
            dout, relu, op, ldo = gmask, 0, None, 0          # from here on as a packed gradient: mask applied
            dop, lddo, dob = pmx(dout)
            packed = True
    if defer and (lddo % 4 or dop % 16 or c % 4):
        defer = False
    if defer and _B16 and dout.dtype != y.dtype:
        defer = False                  # the bf16 lin loaders take g and y in one element type
    red = torch.empty((4, c), device=dev, dtype=torch.float32)           # dgamma, dbeta, c1, c2
    lin = torch.empty((4, c), device=dev, dtype=torch.float32) if defer else None
    linp = None if lin is None else lin.data_ptr()
    if _sync_on():
        local = torch.empty((2 * c + 1,), device=dev, dtype=torch.float64)
        _call("sh_bn_reduce_partials", partials.data_ptr(), p, c, float(m), 0, local.data_ptr(), _st())
        glob = _all_reduce_sq(local.clone())
        _call("sh_bn_bwd_finalize_sq", local.data_ptr(), glob.data_ptr(), c,
              _rp(red, 0), _rp(red, 1), _rp(red, 2), _rp(red, 3),
              None if gamma is None else gamma.data_ptr(), _rp(coefs, 1), _rp(coefs, 0), linp, _st())
    else:
        partials, _ = _fold_partials(partials, c, m, 0)
        p = partials.shape[0]
        _call("sh_bn_bwd_finalize", partials.data_ptr(), p, c, None if gamma is None else gamma.data_ptr(),
              _rp(coefs, 1), float(m), _rp(red, 0), _rp(red, 1), _rp(red, 2), _rp(red, 3),
              _rp(coefs, 0), linp, _st(), key=(None if _PROF is None else f"P{p} C{c}"))
    if defer:
        # relu == 0 here: dout is the masked gradient (packed, or stored by the statistics pass), or no mask applies
        return DeferredDy(dout, y, lin, coefs, gamma, red), red[0], red[1], (dout if want_dres else None)
    ld = (pad8(c) if gdt == torch.bfloat16 else pad4(c)) if dy_ld is None else dy_ld
    dy = new_act(n, c, h, w, dev, ld=ld, zero=ld != c, dtype=gdt)
    dres = None
    if want_dres:
        dres = dout if packed else new_act(n, c, h, w, dev, dtype=gdt)      # packed: g (mask applied) IS the identity path's gradient
    dyp, lddy = pmx(dy)[:2]
    drp, lddr = (None, 0) if (dres is None or packed) else pmx(dres)[:2]
    gb = int(gdt == torch.bfloat16)
    _call("sh_bn_bwd_apply", dop, lddo, op, ldo, yp, ldy, _rp(coefs, 0), _rp(coefs, 1), _rp(coefs, 2),
          _rp(coefs, 3), None if gamma is None else gamma.data_ptr(), _rp(red, 2), _rp(red, 3), dyp, lddy,
          drp, lddr, m, c, relu, af | (dob << 2) | (gb << 3) | (gb << 4), _st(), key=(None if _PROF is None else f"{n}x{h}x{w} C{c}" + (" +dres" if drp else "")))
    return dy, red[0], red[1], dres


# ----------------------------------------------------------------------------- pooling / resampling
def maxpool_fwd(x, want_argmax=True, aff=None):
    """-> (y, argmax): argmax uint8 [N,Ho,Wo,C] (window position of the first maximum) or None.
    aff = (4, C) BatchNorm coefficients: x is the stem conv's raw output, read as relu(x * scale + shift)."""
    n, c, h, w = x.shape
    xp, ldx, xb = pmx(x)
    if ldx != c:
        raise SegHieroHipError("maxpool needs a dense NHWC tensor")
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    y = new_act(n, c, ho, wo, x.device, dtype=x.dtype)                # stored like its input
    am = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8) if want_argmax else None
    _call("sh_maxpool_fwd", xp, None if aff is None else _rp(aff, 2), None if aff is None else _rp(aff, 3), y.data_ptr(),
          None if am is None else am.data_ptr(), n, h, w, c, xb | (xb << 1), _st())
    return y, am


def maxpool_bwd(argmax, dy, h, w):
    """dx [N,C,h,w] from the recorded argmax and dy (the pooled tensor's gradient)."""
    n, c = dy.shape[:2]
    dyp, ld = pm(dy)
    if ld != c:
        raise SegHieroHipError("maxpool backward needs a dense NHWC gradient")
    dx = new_act(n, c, h, w, dy.device)
    _call("sh_maxpool_bwd", argmax.data_ptr(), dyp, dx.data_ptr(), n, h, w, c, _st())
    return dx


def avgpool_fwd(x):
    n, c, h, w = x.shape
    xp, ldx = pm(x)
    y = new_act(n, c, 1, 1, x.device)
    _call("sh_avgpool_fwd", xp, ldx, y.data_ptr(), n, h * w, c, _st())
    return y


def avgpool_bwd(dy, dx, accumulate):
    n, c, h, w = dx.shape
    dxp, lddx = pm(dx)
    _call("sh_avgpool_bwd", dy.data_ptr(), dxp, lddx, n, h * w, c, 1.0 / (h * w), int(accumulate), _st())


def broadcast_hw(x, out):
    n, c, h, w = out.shape
    op, ldo = pm(out)
    _call("sh_broadcast_hw", x.data_ptr(), op, ldo, n, h * w, c, _st())


def sum_hw(dy):
    n, c, h, w = dy.shape
    dyp, ld = pm(dy)
    dx = new_act(n, c, 1, 1, dy.device)
    _call("sh_sum_hw", dyp, ld, dx.data_ptr(), n, h * w, c, _st())
    return dx


def bilinear_fwd(x, out):
    n, c, h, w = x.shape
    _, _, H, W = out.shape
    xp, ldx = pm(x)
    op, ldo, ob = pmx(out)
    _call("sh_bilinear_fwd", xp, ldx, op, ldo, n, h, w, H, W, c, ob, _st())


def bilinear_bwd(dy, h, w):
    n, c, H, W = dy.shape
    dyp, ld = pm(dy)
    dx = new_act(n, c, h, w, dy.device)
    need = LIB.raw("sh_bilinear_bwd_workspace")(n, h, w, c)
    ws = workspace(need, dy.device, "bilinear_bwd")
    _call("sh_bilinear_bwd", dyp, ld, dx.data_ptr(), c, n, h, w, H, W, c, ws.data_ptr(), need, _st())
    return dx


def l2norm_fwd(x):
    n, c, h, w = x.shape
    xp, ld = pm(x)
    if ld != c:
        raise SegHieroHipError("l2norm needs a dense NHWC tensor")
    y = new_act(n, c, h, w, x.device)
    norm = torch.empty((n * h * w,), device=x.device, dtype=torch.float32)
    _call("sh_l2norm_fwd", xp, y.data_ptr(), norm.data_ptr(), n * h * w, c, _st())
    return y, norm


def l2norm_bwd(dy, y, norm):
    n, c, h, w = y.shape
    dy = to_nhwc(dy)
    dyp, ld = pm(dy)
    if ld != c:
        dy = dense_copy(dy)
        dyp = dy.data_ptr()
    dx = new_act(n, c, h, w, y.device)
    _call("sh_l2norm_bwd", dyp, y.data_ptr(), norm.data_ptr(), dx.data_ptr(), n * h * w, c, _st())
    return dx


def dense_copy(t, ld=None):
    """Copy a (possibly channel-sliced) NHWC tensor into a fresh buffer with pixel stride ld (padding zeroed)."""
    n, c, h, w = t.shape
    ld = c if ld is None else ld
    out = new_act(n, c, h, w, t.device, ld=ld, zero=ld != c)
    ones = torch.ones((2, c), device=t.device, dtype=torch.float32)
    ones[1].zero_()
    tp, ldt = pm(t)
    op, ldo = pm(out)
    _call("sh_bn_act", tp, ldt, _rp(ones, 0), _rp(ones, 1), None, 0, None, None, op, ldo, n * h * w, c, 0, None, 0, _st())
    return out


def fill_(t, v):
    _call("sh_fill", t.data_ptr(), float(v), t.numel(), _st())


# ----------------------------------------------------------------------------- labels / losses
def labels_u8(label):
    _require_gpu(label)
    if label.dtype == torch.uint8:
        return label.contiguous()
    if label.dtype != torch.int64:
        raise SegHieroHipError("labels must be int64 or uint8")
    label = label.contiguous()
    out = torch.empty(label.shape, device=label.device, dtype=torch.uint8)
    _call("sh_labels_to_u8", label.data_ptr(), out.data_ptr(), label.numel(), _st())
    return out


def _buckets(hiera_index):
    flat = [int(v) for pair in hiera_index for v in (pair[0], pair[-1])]
    return (ctypes.c_int * max(len(flat), 1))(*flat)


STEP_SCOPE = None         # set by SegHieroTrainer.train_step: {tag: tensor} of buffers that live from the forward to the backward of ONE
                          # step and are handed out again in the next one (one graph in flight at a time -- the trainer's contract)


def _loss_grad_out(want_grad, n, c, h, w, H, W, device, tag="lossgrad"):
    """Buffer for the per-pixel gradient a loss forward leaves for its backward ([N*H*W][pad4(C)] fp32; it lives from the forward to
    the backward of one graph), when the logits are upsampled.  Inside a trainer step (STEP_SCOPE) the same buffer serves every
    step; a standalone loss call gets a fresh tensor (several graphs may be alive at once)."""
    if not (want_grad and LOSS_FWD_GRAD and LOSS_BWD_TWO_PASS) or not (h <= H and w <= W and (h < H or w < W)):
        return None, 0, 0
    ldg = pad4(c)
    need = LIB.raw("sh_loss_bwd_workspace")(n, H, W, ldg)
    if STEP_SCOPE is not None:
        buf = STEP_SCOPE.get(tag)
        if buf is None or buf.numel() < need // 4 or buf.device != device:
            buf = STEP_SCOPE[tag] = torch.empty((need // 4,), device=device, dtype=torch.float32)
        return buf[:need // 4], need, ldg
    return torch.empty((need // 4,), device=device, dtype=torch.float32), need, ldg


def label_counts(labels8, n_fine, hiera_index):
    """-> device int64[3] = {pixels with a valid fine label, with a valid coarse label, all pixels} of this shard: the loss normalisers
    that the exact data-parallel mode all-reduces (ddp.exact_counts)."""
    counts = torch.empty((3,), device=labels8.device, dtype=torch.int64)
    _call("sh_label_counts", labels8.data_ptr(), _buckets(hiera_index), n_fine, len(hiera_index), labels8.numel(), counts.data_ptr(), _st())
    return counts


def hiera2_fwd(logits, labels8, n_fine, hiera_index, want_coarse=False, want_grad=False, norm=None):
    """-> (loss[1] f32, sums[8] f64, coarse u8 or None, grad workspace or None).  want_grad: the same pass leaves the per-pixel
    gradient for hiera2_bwd(grad_ws=...), which is then only the adjoint of the resize.  norm: device int64[3] global normalisers
    (label_counts summed over the ranks): local numerators over global denominators."""
    n, c, h, w = logits.shape
    _, H, W = labels8.shape
    lp, ldl = pm(logits)
    dev = logits.device
    nblk = LIB.raw("sh_hiera2_partials")(n, H, W)
    partials = torch.empty((nblk, 8), device=dev, dtype=torch.float32)
    sums = torch.empty((8,), device=dev, dtype=torch.float64)
    loss = torch.empty((1,), device=dev, dtype=torch.float32)
    coarse = torch.empty_like(labels8) if want_coarse else None
    gw, gbytes, ldg = _loss_grad_out(want_grad, n, c, h, w, H, W, dev)
    _call("sh_hiera2_loss_fwd", lp, ldl, labels8.data_ptr(), _buckets(hiera_index), n_fine, len(hiera_index),
          sums.data_ptr(), loss.data_ptr(), partials.data_ptr(), None if coarse is None else coarse.data_ptr(),
          n, h, w, H, W, None if gw is None else gw.data_ptr(), gbytes, ldg, None if norm is None else norm.data_ptr(), _st())
    return loss, sums, coarse, gw


LOSS_BWD_TWO_PASS = os.environ.get("SEGHIERO_LOSS_TWO_PASS", "1") != "0"
LOSS_FWD_GRAD = os.environ.get("SEGHIERO_LOSS_FWD_GRAD", "1") != "0"    # the loss forward also emits the per-pixel gradient


def _loss_bwd_ws(n, h, w, H, W, ldd, device):
    """Workspace of the two-pass loss backward (full-resolution gradient, [N*H*W][ldd] fp32) when the logits are upsampled."""
    if not LOSS_BWD_TWO_PASS or (h >= H and w >= W) or ldd % 4:
        return None, 0
    need = LIB.raw("sh_loss_bwd_workspace")(n, H, W, ldd)
    return workspace(need, device, "lossbwd").data_ptr(), need


def hiera2_bwd(logits, labels8, n_fine, hiera_index, sums, gscale_dev, gscale, grad_ws=None):
    n, c, h, w = logits.shape
    _, H, W = labels8.shape
    lp, ldl = pm(logits)
    d = new_act(n, c, h, w, logits.device, ld=pad4(c))
    dp, ldd = pm(d)
    if grad_ws is not None:
        ws, nb, has = grad_ws.data_ptr(), grad_ws.numel() * 4, 1
    else:
        (ws, nb), has = _loss_bwd_ws(n, h, w, H, W, ldd, logits.device), 0
    _call("sh_hiera2_loss_bwd", lp, ldl, labels8.data_ptr(), _buckets(hiera_index), n_fine, len(hiera_index),
          sums.data_ptr(), None if gscale_dev is None else gscale_dev.data_ptr(), float(gscale), dp, ldd,
          n, h, w, H, W, ws, nb, has, _st())
    return d


def ce_fwd(logits, labels8, want_grad=False, norm=None):
    """-> (loss[1], sums[2] f64, grad workspace or None); want_grad as hiera2_fwd; norm: device int64[>=1], [0] = global valid count."""
    n, c, h, w = logits.shape
    _, H, W = labels8.shape
    lp, ldl = pm(logits)
    dev = logits.device
    nblk = LIB.raw("sh_hiera2_partials")(n, H, W)
    partials = torch.empty((nblk, 8), device=dev, dtype=torch.float32)
    sums = torch.empty((2,), device=dev, dtype=torch.float64)
    loss = torch.empty((1,), device=dev, dtype=torch.float32)
    gw, gbytes, ldg = _loss_grad_out(want_grad, n, c, h, w, H, W, dev, tag="lossgrad_ce")
    _call("sh_ce_loss_fwd", lp, ldl, labels8.data_ptr(), c, sums.data_ptr(), loss.data_ptr(), partials.data_ptr(),
          n, h, w, H, W, None if gw is None else gw.data_ptr(), gbytes, ldg, None if norm is None else norm.data_ptr(), _st())
    return loss, sums, gw


def ce_bwd(logits, labels8, sums, gscale_dev, gscale, grad_ws=None):
    n, c, h, w = logits.shape
    _, H, W = labels8.shape
    lp, ldl = pm(logits)
    d = new_act(n, c, h, w, logits.device, ld=pad4(c))
    dp, ldd = pm(d)
    if grad_ws is not None:
        ws, nb, has = grad_ws.data_ptr(), grad_ws.numel() * 4, 1
    else:
        (ws, nb), has = _loss_bwd_ws(n, h, w, H, W, ldd, logits.device), 0
    _call("sh_ce_loss_bwd", lp, ldl, labels8.data_ptr(), c, sums.data_ptr(),
          None if gscale_dev is None else gscale_dev.data_ptr(), float(gscale), dp, ldd, n, h, w, H, W, ws, nb, has, _st())
    return d


def triplet_fwd(emb, labels8, masks, anchor_ok, max_triplet=200, margin=0.6):
    """emb: dense NHWC [n,d,h,w].  -> (out[2] f32 = {loss, class_count}, workspace)."""
    n, d, h, w = emb.shape
    _, H, W = labels8.shape
    ep, ld = pm(emb)
    if ld != d:
        raise SegHieroHipError("triplet needs a dense NHWC embedding")
    need = LIB.raw("sh_triplet_workspace")(n * h * w)
    ws = torch.empty((need,), device=emb.device, dtype=torch.uint8)
    out = torch.empty((2,), device=emb.device, dtype=torch.float32)
    _call("sh_triplet_fwd", ep, labels8.data_ptr(), masks.data_ptr(), anchor_ok.data_ptr(), max_triplet, margin,
          out.data_ptr(), ws.data_ptr(), n, h, w, d, H, W, _st())
    return out, ws


def triplet_bwd(emb, ws, out, gscale_dev, gscale):
    n, d, h, w = emb.shape
    demb = new_act(n, d, h, w, emb.device, zero=True)
    _call("sh_triplet_bwd", emb.data_ptr(), ws.data_ptr(), out.data_ptr(),
          None if gscale_dev is None else gscale_dev.data_ptr(), float(gscale), demb.data_ptr(), n, h, w, d, _st())
    return demb


def combine_loss(main, trip_out, ready_count, factor, loss_weight):
    out = torch.empty((1,), device=main.device, dtype=torch.float32)
    _call("sh_combine_loss", main.data_ptr(), trip_out.data_ptr(), None if ready_count is None else ready_count.data_ptr(),
          float(factor), float(loss_weight), out.data_ptr(), _st())
    return out


def pixel_metrics(logits, labels8, n_fine, counts=None):
    n, c, h, w = logits.shape
    _, H, W = labels8.shape
    lp, ldl = pm(logits)
    if counts is None:
        counts = torch.zeros((2 + n_fine * n_fine,), device=logits.device, dtype=torch.int64)
    _call("sh_pixel_metrics", lp, ldl, labels8.data_ptr(), n_fine, counts.data_ptr(), n, h, w, H, W, _st())
    return counts


# ----------------------------------------------------------------------------- optimizer
SGD_MAX = 48
WEIGHT_EPOCH = 0          # bumped by every sgd_step launch: the kernel writes the parameters through raw pointers, which torch's
                          # tensor version counter does not see -- derived copies (the padded stem weight) are keyed on both


def weights_key(w):
    return (w.data_ptr(), w._version, WEIGHT_EPOCH)


_SGD_PLAN = {}          # (first parameter's address, count) -> (signature, launch tables): one entry per optimizer in the process


def sgd_step(params, grads, bufs, lr, momentum, weight_decay, first_step, gscale=1.0):
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1
    st = _st()
    # the weight / momentum / size tables are the same every step (only the gradient tensors are new): kept until a pointer moves
    sig = (tuple(p.data_ptr() for p in params), tuple(v.data_ptr() for v in bufs))
    key = (sig[0][0], len(params))
    plan = _SGD_PLAN.get(key)
    if plan is None or plan[0] != sig:
        chunks = []
        for i in range(0, len(params), SGD_MAX):
            ps, vs = params[i:i + SGD_MAX], bufs[i:i + SGD_MAX]
            k = len(ps)
            chunks.append((i, k, (ctypes.c_void_p * k)(*[p.data_ptr() for p in ps]), (ctypes.c_void_p * k)(*[v.data_ptr() for v in vs]),
                           (ctypes.c_longlong * k)(*[p.numel() for p in ps])))
        if len(_SGD_PLAN) > 16:
            _SGD_PLAN.clear()
        plan = _SGD_PLAN[key] = (sig, chunks)
    for i, k, wa, va, na in plan[1]:
        ga = (ctypes.c_void_p * k)(*[g.data_ptr() for g in grads[i:i + k]])
        _call("sh_sgd_step", k, wa, ga, va, na, float(lr), float(momentum), float(weight_decay), int(first_step),
              float(gscale), st)


# ----------------------------------------------------------------------------- 3-level loss + RMI
def _ints(values):
    vals = [int(v) for v in values]
    return (ctypes.c_int * max(len(vals), 1))(*vals)


def hiera3_fwd(logits, labels8, n_fine, n_mid, n_high, f2m, f2h, want_probs, want_targets=False, want_grad=False):
    """-> (loss_rest[1] f32, sums[8] f64, probs planar [N,C,H,W] or None[, (mid u8, high u8) target maps]).  want_grad: the same pass
    leaves the per-pixel gradient of everything but the RMI term and a fourth value is returned: (workspace or None, its row stride) for
    hiera3_bwd(grad_ws=...)."""
    n, c, h, w = logits.shape
    _, H, W = labels8.shape
    lp, ldl = pm(logits)
    dev = logits.device
    nblk = LIB.raw("sh_hiera2_partials")(n, H, W)
    partials = torch.empty((nblk, 8), device=dev, dtype=torch.float32)
    sums = torch.empty((8,), device=dev, dtype=torch.float64)
    loss = torch.empty((1,), device=dev, dtype=torch.float32)
    probs = torch.empty((n, c, H, W), device=dev, dtype=torch.float32) if want_probs else None
    mid = torch.empty_like(labels8) if want_targets else None
    high = torch.empty_like(labels8) if want_targets else None
    gw, gbytes, ldg = (None, 0, 0)
    if want_grad and LOSS_FWD_GRAD and LOSS_BWD_TWO_PASS and (h < H or w < W) and c <= 32:
        ldg = 16 if c <= 16 else 32
        gbytes = n * H * W * ldg * 4
        gw = STEP_SCOPE.get("lossgrad3") if STEP_SCOPE is not None else None
        if gw is None or gw.numel() * 4 < gbytes or gw.device != dev:
            gw = torch.empty((gbytes // 4,), device=dev, dtype=torch.float32)
            if STEP_SCOPE is not None:
                STEP_SCOPE["lossgrad3"] = gw
    _call("sh_hiera3_loss_fwd", lp, ldl, labels8.data_ptr(), _ints(f2m), _ints(f2h), n_fine, n_mid, n_high, sums.data_ptr(),
          loss.data_ptr(), partials.data_ptr(), None if probs is None else probs.data_ptr(),
          None if mid is None else mid.data_ptr(), None if high is None else high.data_ptr(), n, h, w, H, W,
          None if gw is None else gw.data_ptr(), gbytes, ldg, _st())
    if want_targets:
        return loss, sums, probs, (mid, high)
    if want_grad:
        return loss, sums, probs, (gw, ldg)
    return loss, sums, probs


def rmi_loss(probs, labels8, n_fine, n_mid, n_high, f2m, f2h, want_grad):
    """-> (rmi[1] f32, dprob planar or None)."""
    n, c, H, W = probs.shape
    need = LIB.raw("sh_rmi_workspace")(n, c, H, W)
    ws = workspace(need, probs.device, "rmi")
    out = torch.empty((1,), device=probs.device, dtype=torch.float32)
    dprob = torch.empty_like(probs) if want_grad else None
    _call("sh_rmi_loss", probs.data_ptr(), labels8.data_ptr(), _ints(f2m), _ints(f2h), n_fine, n_mid, n_high, ws.data_ptr(),
          out.data_ptr(), None if dprob is None else dprob.data_ptr(), n, H, W, _st())
    return out, dprob


def rmi_values(n, c, H, W, device):
    """f64 [n, c] rmi_now = 0.5*logdet per (image, channel) left in the workspace by the last rmi_loss call on this stream."""
    off = LIB.raw("sh_rmi_values_offset")(n, c, H, W)
    ws = workspace(LIB.raw("sh_rmi_workspace")(n, c, H, W), device, "rmi")
    return ws[off:off + 8 * n * c].view(torch.float64).reshape(n, c).clone()


def hiera3_bwd(logits, labels8, n_fine, n_mid, n_high, f2m, f2h, sums, dprob, rmi_coef, gscale_dev, gscale, grad_ws=None, probs=None):
    """grad_ws = (workspace, ldg) left by hiera3_fwd(want_grad=True): the gather-only backward (+ the RMI term from dprob and probs)."""
    n, c, h, w = logits.shape
    _, H, W = labels8.shape
    lp, ldl = pm(logits)
    if grad_ws is not None and grad_ws[0] is not None:
        gw, ldg = grad_ws
        d = new_act(n, c, h, w, logits.device, ld=ldg)
        dp, ldd = pm(d)
        _call("sh_hiera3_loss_bwd", lp, ldl, labels8.data_ptr(), _ints(f2m), _ints(f2h), n_fine, n_mid, n_high, sums.data_ptr(),
              None if dprob is None else dprob.data_ptr(), float(rmi_coef), None if gscale_dev is None else gscale_dev.data_ptr(),
              float(gscale), dp, ldd, n, h, w, H, W, gw.data_ptr(), gw.numel() * 4, 1, None if probs is None else probs.data_ptr(), _st())
        return d
    d = new_act(n, c, h, w, logits.device, ld=pad4(c))
    dp, ldd = pm(d)
    ws, nb = _loss_bwd_ws(n, h, w, H, W, ldd, logits.device)
    _call("sh_hiera3_loss_bwd", lp, ldl, labels8.data_ptr(), _ints(f2m), _ints(f2h), n_fine, n_mid, n_high, sums.data_ptr(),
          None if dprob is None else dprob.data_ptr(), float(rmi_coef), None if gscale_dev is None else gscale_dev.data_ptr(),
          float(gscale), dp, ldd, n, h, w, H, W, ws, nb, 0, None, _st())
    return d


def scalar_axpy(a, b, alpha):
    out = torch.empty((1,), device=a.device, dtype=torch.float32)
    _call("sh_scalar_axpy", a.data_ptr(), b.data_ptr(), float(alpha), out.data_ptr(), _st())
    return out
