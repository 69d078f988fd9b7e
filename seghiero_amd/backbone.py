"""ResNetBackbone -- drop-in for reference ``models/backbone/resnet.py:6-75`` on the HIP kernels.

Same constructor (``depth``, ``pretrained``), same attribute names (``stem_conv, stem_bn, stem_relu, stem_pool,
layer1..layer4``) and therefore the same state_dict keys / shapes as the reference (which takes them from
torchvision's ResNet), same forward contract ``x[B,3,H,W] -> (c1, c2, c3, c4)`` at strides 4/8/16/32.
The torch.nn modules below only hold parameters and buffers; the forward runs the implicit-GEMM conv, BN and
pooling kernels and the backward is hand-scheduled (``_BackboneFn``), so autograd sees ONE node for the trunk.

Superset: depth 18/34/152 are accepted too (README.md:95 promises them; BASELINE config 1 needs 18).
``pretrained=True`` is a network download in the reference (``resnet.py:35,37``); there is no network here, so it
warns and keeps the random initialisation (load weights with ``load_state_dict``).
"""
import warnings

import torch
import torch.nn as nn

from . import layers as L
from . import ops
from ._lib import SegHieroHipError

_SPECS = {18: ("basic", (2, 2, 2, 2)), 34: ("basic", (3, 4, 6, 3)), 50: ("bottleneck", (3, 4, 6, 3)),
          101: ("bottleneck", (3, 4, 23, 3)), 152: ("bottleneck", (3, 8, 36, 3))}


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)


class _Block(nn.Module):
    """Parameter container for one residual block (torchvision naming: conv1/bn1/.../downsample)."""

    def __init__(self, kind, cin, width, stride, downsample):
        super().__init__()
        self.kind = kind
        if kind == "basic":
            self.conv1, self.bn1 = _conv(cin, width, 3, stride, 1), nn.BatchNorm2d(width)
            self.relu = nn.ReLU(inplace=True)
            self.conv2, self.bn2 = _conv(width, width, 3, 1, 1), nn.BatchNorm2d(width)
        else:
            self.conv1, self.bn1 = _conv(cin, width, 1), nn.BatchNorm2d(width)
            self.conv2, self.bn2 = _conv(width, width, 3, stride, 1), nn.BatchNorm2d(width)
            self.conv3, self.bn3 = _conv(width, width * 4, 1), nn.BatchNorm2d(width * 4)
            self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        # the hand-scheduled forward / backward walk these per step: plain instance attributes (nn.Module resolves `self.conv1` through
        # __getattr__ and three dict probes -- ~1 us each, ~1 000 of them per ResNet-101 step); the module tree itself is unchanged
        m = self._modules
        self.__dict__["_chain"] = [(m["conv1"], m["bn1"]), (m["conv2"], m["bn2"])] + ([] if kind == "basic" else [(m["conv3"], m["bn3"])])
        self.__dict__["_ds"] = None if downsample is None else (downsample[0], downsample[1])

    def chain(self):
        return self._chain

    def forward(self, x):
        raise SegHieroHipError("blocks run inside ResNetBackbone.forward (hand-scheduled HIP path)")


def _stage(kind, cin, width, n, stride):
    exp = 1 if kind == "basic" else 4
    cout = width * exp
    ds = None
    if stride != 1 or cin != cout:
        ds = nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))
    blocks = [_Block(kind, cin, width, stride, ds)] + [_Block(kind, cout, width, 1, None) for _ in range(n - 1)]
    return nn.Sequential(*blocks), cout


# Residual-block form of the fused BatchNorm backward (the next block's first dgrad epilogue masks with this block's output and
# emits its bn3 statistics).  Measured at the headline shape (same-box A/B): while the epilogue spilled 85 registers this cost more
# (+1.6 ms of dgrad) than the statistics pass it replaces (-0.7 ms); with the epilogue processed in row-group chunks (no spills) it
# wins: dgrad +0.6 ms, statistics pass -1.0 ms, step 33.94 -> 33.45 ms -- on by default.
BNB_RESIDUAL = __import__("os").environ.get("SEGHIERO_BNB_RESIDUAL", "1") != "0"
# The gradient a stage output receives from outside the trunk (c1 <- decoder, c3 <- aux head) is summed by the first dgrad epilogue of the
# next stage's first block instead of an axpy pass over the 268 MB / 67 MB tensor: same-box A/B 33.00 -> 32.85 ms per step.
STAGE_GRAD_IN_EPILOGUE = __import__("os").environ.get("SEGHIERO_STAGE_GRAD_EPI", "1") != "0"


# ----------------------------------------------------------------------------- hand-scheduled fwd / bwd
def _block_fwd(blk, x, training):
    chain = blk.chain()
    recs = []
    idt, ds_rec = x, None
    if blk._ds is not None:
        dconv, dbn = blk._ds
        # the downsample BatchNorm is applied inside the block's final BatchNorm + add + ReLU pass (LazyAffine)
        idt, ds_rec = L.cba_fwd(x, L.W(dconv), L.conv_geom(dconv), dbn, False, training, lazy=True)
    h = x
    for i, (conv, bn) in enumerate(chain):
        last = i == len(chain) - 1
        # inner convs hand their BatchNorm + ReLU to the next conv's loader (Lazy); the block output (+ identity) is materialised
        h, rec = L.cba_fwd(h, L.W(conv), L.conv_geom(conv), bn, True, training, residual=idt if last else None, lazy=not last)
        recs.append(rec)
    return h, (recs, ds_rec)


def _block_bwd(blk, saved, dout, gm, prev_rec=None, extra=None, in32=False):
    """dout: gradient w.r.t. the block output (tensor, or a GradPack made by the NEXT block's first conv).  prev_rec: last CBARec of
    the previous block when this block's input is that block's output and no downsample path adds into the input gradient -- the
    first conv's dgrad epilogue then runs the front half of that block's bn3 backward (mask from its `out`) and the identity
    gradient needs no separate tensor.  extra (blocks with a downsample path only): a tensor of the input's shape summed into the
    input gradient by the first conv's dgrad epilogue -- the previous stage's own output gradient (c1 / c3 feed the head).
    in32 (bf16 compute mode): the block's input gradient goes to an fp32-only consumer (the max-pool backward)."""
    recs, ds_rec = saved
    chain = blk.chain()
    # a strided downsample branch scatters into the input gradient: where only the fp32-accurate kernel takes it, that tensor is fp32
    if ds_rec is not None and blk._ds[0].stride[0] > 1:
        dc = blk._ds[0]
        in32 = in32 or not ops.strided_dgrad_b16_ok(L.W(dc), dc.in_channels, dc.stride[0], dc.padding[0], dc.dilation[0], True)
    # last conv: g = dout * relu-mask feeds BN backward AND (as dres) the identity / downsample path
    conv, bn = chain[-1]
    d, dw, dg, db, dres = L.cba_bwd(recs[-1], bn, dout, need_dx=True, want_dres=True)
    gm.put(L.W(conv), dw); gm.put(L.W(bn), dg); gm.put(L.Bi(bn), db)
    for i in range(len(chain) - 2, -1, -1):
        conv, bn = chain[i]
        first = i == 0
        addend = dres if (first and ds_rec is None) else (extra if first else None)      # identity path summed in the dgrad epilogue
        d, dw, dg, db, _ = L.cba_bwd(recs[i], bn, d, need_dx=True, addend=addend,
                                     pack_for=prev_rec if (first and ds_rec is None) else None, dx32=first and in32)
        gm.put(L.W(conv), dw); gm.put(L.W(bn), dg); gm.put(L.Bi(bn), db)
    if ds_rec is not None:
        dconv, dbn = blk._ds
        if dconv.stride[0] > 1:
            _, dw, dg, db, _ = L.cba_bwd(ds_rec, dbn, dres, scatter_into=d)          # accumulate at strided pixels
        else:
            d, dw, dg, db, _ = L.cba_bwd(ds_rec, dbn, dres, need_dx=True, addend=d, dx32=in32)
        gm.put(L.W(dconv), dw); gm.put(L.W(dbn), dg); gm.put(L.Bi(dbn), db)
    return d


def _all_bns(mod):
    bns = mod.__dict__.get("_bn_list")           # the module tree is fixed after construction: walk it once
    if bns is None:
        bns = [m for m in mod.modules() if isinstance(m, nn.BatchNorm2d)]
        mod.__dict__["_bn_list"] = bns
    return bns


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        ctx.set_materialize_grads(False)      # an unused stage output (c2 in SegHiero) gets None, not a zero tensor to convert and add
        training = mod.training
        n, cin, hh, ww = x.shape
        if cin == 4:                                                   # already ingested (seghiero_amd.ingest): NHWC4, 4th channel 0
            x4 = x
        else:
            x4 = ops.new_act(n, 4, hh, ww, x.device)                   # NHWC, 3 -> 4 channels (4th = 0)
            xc = x if x.is_contiguous() else x.contiguous()
            ops._call("sh_nchw_to_nhwc", xc.data_ptr(), x4.data_ptr(), n, 3, hh, ww, 4, ops._st())
        w = mod.stem_conv.weight                                       # [64,3,7,7] -> OHWI with I padded to 4
        if not w.is_contiguous():
            raise SegHieroHipError("stem_conv.weight must be contiguous")
        # the OHWI copy with I padded to 4 is kept until the weight changes (an optimizer step, load_state_dict, .to())
        hit, wkey = mod.__dict__.get("_stem_wpad"), ops.weights_key(w)
        if hit is not None and hit[0] == wkey:
            wpad = hit[1]
        else:
            wpad = ops.new_act(w.shape[0], 4, 7, 7, w.device, zero=True)
            ops._call("sh_nchw_to_nhwc", w.data_ptr(), wpad.data_ptr(), w.shape[0], 3, 7, 7, 4, ops._st())
            mod.__dict__["_stem_wpad"] = (wkey, wpad)
        # act_dtype = torch.bfloat16 (training only): the trunk's raw conv outputs and block outputs are STORED as bf16 -- half the
        # bytes of every activation read and write; arithmetic, BatchNorm statistics and gradients stay fp32 (BASELINE configs[4])
        stored = mod.act_dtype if (training and L.FUSE_BN and ops.CONV_IMPL == "x6") else torch.float32
        # compute_dtype = torch.bfloat16 (with bf16 storage): ONE bf16 MFMA product per tile on operands rounded once to bf16 in the
        # loaders (csrc/conv_b16.hip) instead of the fp32-accurate six-product plan; stage outputs are then handed on as bf16 tensors
        compute = mod.compute_dtype if stored == torch.bfloat16 else torch.float32
        with ops.stored_as(stored), ops.compute_as(compute):
            s_out, s_rec = L.cba_fwd(x4, wpad, L.conv_geom(mod.stem_conv), mod.stem_bn, True, training, lazy=True)
            if isinstance(s_out, L.Lazy):          # stem_bn + stem_relu run in the pooling kernel's loader
                pooled, pool_idx = ops.maxpool_fwd(s_out.y, want_argmax=training, aff=s_out.coefs)
            else:
                pooled, pool_idx = ops.maxpool_fwd(s_out, want_argmax=training)
            h = pooled
            saved, outs = [], []
            for layer in (mod.layer1, mod.layer2, mod.layer3, mod.layer4):
                for blk in layer:
                    h, sv = _block_fwd(blk, h, training)
                    saved.append(sv)
                outs.append(h)
        if mod.__dict__.get("_relu_mask_sink") is not None:                 # parity tests only (ResNetBackbone.export_relu_masks)
            mod.__dict__["_relu_mask_sink"].extend(_relu_masks(s_rec, saved))
        outs = [o if o.dtype == torch.float32 else o.float() for o in outs]          # the head and aux head take fp32 stage outputs
        if training:
            L.bump_bn_counters(_all_bns(mod))
        if training:
            s_rec.out = None                                           # BN+ReLU backward recomputes its mask from y (no residual here)
        ctx.mod, ctx.saved, ctx.stem = mod, saved, (x4, wpad, s_rec, pool_idx, tuple(s_out.shape[2:]))
        ctx.training = training
        ctx.params = params
        ctx.compute = compute
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        if not ctx.training:
            raise SegHieroHipError("backward through eval-mode BatchNorm is not on the SegHiero hot path")
        with ops.compute_as(ctx.compute):
            return _BackboneFn._backward(ctx, *douts)

    @staticmethod
    def _backward(ctx, *douts):
        mod, saved = ctx.mod, ctx.saved
        gm = L.GradMap()
        layers = [mod.layer1, mod.layer2, mod.layer3, mod.layer4]
        blocks = [(li, blk) for li, layer in enumerate(layers) for blk in layer]
        last_of_layer = {}
        for idx, (li, _) in enumerate(blocks):
            last_of_layer[li] = idx
        d = None
        summed = set()                                                   # stages whose own output gradient is already inside d
        for idx in range(len(blocks) - 1, -1, -1):
            li, blk = blocks[idx]
            if last_of_layer[li] == idx and douts[li] is not None and li not in summed:       # this block's output is c_{li+1}
                g = L.grad_as_nhwc_padded(douts[li], douts[li].shape[1])
                if d is None:
                    d = g
                else:
                    if isinstance(d, L.GradPack):
                        raise SegHieroHipError("a packed gradient cannot take a stage gradient (STAGE_GRAD_IN_EPILOGUE covers this case)")
                    d = ops.f32(d)
                    ops._call("sh_axpy", d.data_ptr(), _dense(g).data_ptr(), 1.0, d.numel(), ops._st())
            if d is None:
                continue
            prev_rec = None
            if BNB_RESIDUAL and idx > 0 and blocks[idx - 1][0] == li and blk.downsample is None:
                prev_rec = saved[idx - 1][0][-1]                      # previous block of the same stage feeds this one directly
            extra = None
            if STAGE_GRAD_IN_EPILOGUE and idx > 0 and blocks[idx - 1][0] != li and blk.downsample is not None and douts[blocks[idx - 1][0]] is not None:
                # first block of a stage: the previous stage's own output gradient (c1 from the decoder, c3 from the aux head) rides in
                # this block's first dgrad epilogue instead of a separate read-modify-write pass over the stage output
                lp = blocks[idx - 1][0]
                extra = L.grad_as_nhwc_padded(douts[lp], douts[lp].shape[1])
                summed.add(lp)
            d = _block_bwd(blk, saved[idx], d, gm, prev_rec, extra, in32=idx == 0)
            if idx == 0 or blocks[idx - 1][0] != li:                      # first block of a stage: the stage's gradients are final
                gm.flush(L.params_of(layers[li]))
        if d is not None:
            x4, wpad, s_rec, pool_idx, (sh, sw) = ctx.stem
            dpool = ops.maxpool_bwd(pool_idx, ops.f32(d), sh, sw)
            _, dwp, dg, db, _ = L.cba_bwd(s_rec, mod.stem_bn, dpool, need_dx=False)
            dw = L.new_grad(mod.stem_conv.weight)
            ops.join_wgrad()                                               # dwp comes from the weight-gradient stream
            ops._call("sh_nhwc_to_nchw", dwp.data_ptr(), dw.data_ptr(), dw.shape[0], 3, 7, 7, 4, ops._st())
            gm.put(mod.stem_conv.weight, dw); gm.put(mod.stem_bn.weight, dg); gm.put(mod.stem_bn.bias, db)
            gm.flush([mod.stem_conv.weight, mod.stem_bn.weight, mod.stem_bn.bias])
        ctx.saved = ctx.stem = None
        return (None, None) + gm.ordered(ctx.params)


def _dense(t):
    return t if ops.pm(t)[1] == t.shape[1] else ops.dense_copy(t)


def _relu_masks(s_rec, saved):
    """The 0/1 masks of every ReLU of one training forward, in forward order (stem, then per block: one per conv), exactly as the
    kernels evaluate them: ``y * scale + shift > 0`` from the raw conv output and the BatchNorm coefficients where the activation is
    deferred to a loader, ``out > 0`` for a block output.  Parity tests pin torch's ReLUs to these (a pre-activation that two fp32
    evaluations round to different sides of 0 otherwise moves every upstream gradient by O(1e-4): DESIGN.md, parity section)."""
    def pre(rec):
        return (rec.y.float() * rec.coefs[2].view(1, -1, 1, 1) + rec.coefs[3].view(1, -1, 1, 1)) > 0
    masks = [pre(s_rec)]
    for recs, _ in saved:
        for k, rec in enumerate(recs):
            masks.append((rec.out > 0) if k == len(recs) - 1 else pre(rec))
    return masks


class ResNetBackbone(nn.Module):
    def __init__(self, depth: int = 101, pretrained: bool = True):
        super().__init__()
        if depth not in _SPECS:
            raise ValueError("`depth` must be 50 or 101 (this build also accepts 18, 34, 152)")
        if pretrained:
            warnings.warn("pretrained=True needs a network download (reference resnet.py:35,37); offline build keeps "
                          "the random initialisation -- load weights with load_state_dict()", stacklevel=2)
        kind, counts = _SPECS[depth]
        self.stem_conv = _conv(3, 64, 7, 2, 3)
        self.stem_bn = nn.BatchNorm2d(64)
        self.stem_relu = nn.ReLU(inplace=True)
        self.stem_pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        c = 64
        self.layer1, c = _stage(kind, c, 64, counts[0], 1)
        self.layer2, c = _stage(kind, c, 128, counts[1], 2)
        self.layer3, c = _stage(kind, c, 256, counts[2], 2)
        self.layer4, c = _stage(kind, c, 512, counts[3], 2)
        exp = 1 if kind == "basic" else 4
        self.out_channels = (64 * exp, 128 * exp, 256 * exp, 512 * exp)
        # storage type of the trunk's activations in training (not a reference argument): torch.float32, or torch.bfloat16 =
        # BASELINE configs[4]'s bf16 activation storage (see _BackboneFn.forward); set the attribute after construction
        self.act_dtype = torch.float32
        self.compute_dtype = torch.float32      # torch.bfloat16 (needs act_dtype = bfloat16): bf16 compute mode, see _BackboneFn.forward
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        L.to_native_layout(self)

    def export_relu_masks(self, sink):
        """Test hook: with a list, every training forward appends its ReLU masks (_relu_masks) to it; None switches it off."""
        self.__dict__["_relu_mask_sink"] = sink

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        L.to_native_layout(self)
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        L.to_native_layout(self)
        return out

    def forward(self, x: torch.Tensor):
        ingested = x.dim() == 4 and x.shape[1] == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
        if x.dim() != 4 or (x.shape[1] != 3 and not ingested):
            raise ValueError("expected input of shape [B, 3, H, W] (or the NHWC4 tensor made by seghiero_amd.ingest)")
        ops._require_gpu(x)
        return _BackboneFn.apply(self, x, *L.params_of(self))
