"""Drop-in loss modules over the fused HIP loss kernels.

Mirrors (same class names, constructor / forward signatures, attributes, error behaviour):

* ``CrossEntropyLoss``   <- reference ``models/loss/cross_entropy_loss.py:136-195`` (softmax branch; the reduction is
  the reference's all-pixel ``.mean()``, ``utils.py:6-55``)
* ``TreeTripletLoss``    <- ``models/loss/tree_triplet_loss.py:6-65``
* ``HieraTripletLoss``   <- ``models/loss/hiera_triplet_loss.py:110-211``

The kernels take the logits either at full resolution (what the reference's ``train.py`` hands over after
``F.interpolate``) or at the head's 1/4 resolution, in which case the bilinear resize of ``train.py:282-284`` is fused
into the loss and its backward (``seghiero_amd.train_step`` uses that path).
"""
import math

import torch
import torch.nn as nn

from . import ops
from ._lib import SegHieroHipError

IGNORE = 255


def _bitset(members):
    words = [0, 0, 0, 0]
    for v in members:
        words[v >> 6] |= 1 << (v & 63)
    return [w - (1 << 64) if w >= (1 << 63) else w for w in words]


def two_level_triplet_tables(hiera_map, hiera_index):
    """Per anchor class ii: positives = labels inside ii's bucket range except ii, negatives = every label outside
    the range, 255 included (``tree_triplet_loss.py:33-36``)."""
    masks = torch.zeros((256, 2, 4), dtype=torch.int64)
    ok = []
    for ii in range(min(len(hiera_map), 255)):
        rng = hiera_index[hiera_map[ii]]
        s, e = int(rng[0]), int(rng[-1])
        inside = [v for v in range(256) if s <= v < e]
        masks[ii, 0] = torch.tensor(_bitset([v for v in inside if v != ii]))
        masks[ii, 1] = torch.tensor(_bitset([v for v in range(256) if not (s <= v < e)]))
        ok.append(ii)
    return masks, torch.tensor(_bitset(ok), dtype=torch.int64)


def triplet_factor(step, total_steps):
    """Cosine ramp of ``hiera_triplet_loss.py:203-208``."""
    if step < total_steps:
        return 0.25 * (1 + math.cos((step - total_steps) / total_steps * math.pi))
    return 0.5


def _step_value(step):
    return int(step.item()) if torch.is_tensor(step) else int(step)


def _dense_nhwc(t):
    t = ops.to_nhwc(t)
    if ops.pm(t)[1] != t.shape[1]:
        t = ops.dense_copy(t)
    return t


# ----------------------------------------------------------------------------- cross entropy (all-pixel mean)
class _CEAllPixFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_score, label8):
        logits = ops.to_nhwc(cls_score)
        _, sums = ops.ce_fwd(logits, label8)
        npix = label8.numel()
        ctx.save_for_backward(logits, label8, sums)
        ctx.npix = npix
        return (sums[0] / npix).float()

    @staticmethod
    def backward(ctx, g):
        logits, label8, sums = ctx.saved_tensors
        s2 = sums.clone()
        s2[1] = float(ctx.npix)                       # divide by ALL pixels, not by the valid ones
        d = ops.ce_bwd(logits, label8, s2, g.reshape(1).float(), 1.0)
        return d, None


class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None, loss_weight=1.0):
        super().__init__()
        if use_sigmoid or use_mask or class_weight is not None or reduction != "mean":
            raise NotImplementedError("only the softmax / mean branch is on the SegHiero hot path")
        self.loss_weight = loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        return self.loss_weight * _CEAllPixFn.apply(cls_score, ops.labels_u8(label))


# ----------------------------------------------------------------------------- tree triplet
class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, label8, masks, anchor_ok, max_triplet):
        emb = _dense_nhwc(feats)
        out, ws = ops.triplet_fwd(emb, label8, masks, anchor_ok, max_triplet, 0.6)
        ctx.save_for_backward(emb, ws, out)
        ctx.mark_non_differentiable(ws)
        return out[0].clone(), out[1].clone()

    @staticmethod
    def backward(ctx, g, _gcount):
        emb, ws, out = ctx.saved_tensors
        return ops.triplet_bwd(emb, ws, out, g.reshape(1).float(), 1.0), None, None, None, None


class TreeTripletLoss(nn.Module):
    def __init__(self, num_classes, hiera_map, hiera_index, ignore_index=IGNORE):
        super().__init__()
        self.ignore_label = ignore_index
        self.num_classes = num_classes
        self.hiera_map = hiera_map
        self.hiera_index = hiera_index
        masks, ok = two_level_triplet_tables(hiera_map, hiera_index)
        self.register_buffer("_masks", masks, persistent=False)
        self.register_buffer("_anchor_ok", ok, persistent=False)

    def tables(self, device):
        if self._masks.device != device:
            self._masks = self._masks.to(device)
            self._anchor_ok = self._anchor_ok.to(device)
        return self._masks, self._anchor_ok

    def forward(self, feats, labels=None, max_triplet=200):
        """-> (loss or None, LongTensor([class_count])), like the reference (this standalone entry point reads the
        count back to decide on ``None``; the fused HieraTripletLoss path does not synchronise)."""
        masks, ok = self.tables(feats.device)
        loss, count = _TripletFn.apply(feats, ops.labels_u8(labels), masks, ok, max_triplet)
        n = int(count.item())
        cnt = torch.tensor([n], device=feats.device)
        return (None, cnt) if n == 0 else (loss, cnt)


# ----------------------------------------------------------------------------- fused 2-level loss
class _Hiera2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_score, embedding, label8, mod, step):
        logits = ops.to_nhwc(cls_score)
        emb = _dense_nhwc(embedding)
        nf, hidx = mod.num_classes, mod.hiera_index
        main, sums, _ = ops.hiera2_fwd(logits, label8, nf, hidx)
        masks, ok = mod.triplet_loss_fn.tables(logits.device)
        trip, ws = ops.triplet_fwd(emb, label8, masks, ok, 200, 0.6)
        ready = None
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # hiera_triplet_loss.py:193-198: the term counts only if EVERY rank produced triplets
            from . import ddp
            ready = trip[1:2].clone()
            ddp.all_reduce(ready, op=torch.distributed.ReduceOp.MIN)
        factor = triplet_factor(step, 80000)
        total = ops.combine_loss(main, trip, ready, factor, mod.loss_weight)
        ctx.save_for_backward(logits, emb, label8, sums, trip, ws)
        ctx.ready = ready
        ctx.cfg = (nf, hidx, factor, mod.loss_weight)
        mod.last_terms = (main, trip)
        return total.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, emb, label8, sums, trip, ws = ctx.saved_tensors
        nf, hidx, factor, lw = ctx.cfg
        g = g.reshape(1).float()
        dlogits = ops.hiera2_bwd(logits, label8, nf, hidx, sums, g, lw) if ctx.needs_input_grad[0] else None
        demb = None
        if ctx.needs_input_grad[1]:
            gt = g if ctx.ready is None else g * (ctx.ready > 0).float()
            demb = ops.triplet_bwd(emb, ws, trip, gt, factor * lw)
        return dlogits, demb, None, None, None


class HieraTripletLoss(nn.Module):
    def __init__(self, num_classes: int, hiera_map: list, hiera_index: list, ignore_index: int = IGNORE,
                 use_sigmoid: bool = False, loss_weight: float = 1.0):
        super().__init__()
        if ignore_index != IGNORE:
            raise SegHieroHipError("the kernels assume ignore_index == 255 (uint8 label maps)")
        self.num_classes = num_classes
        self.hiera_map = hiera_map
        self.hiera_index = hiera_index
        self.ignore_index = ignore_index
        self.ce = CrossEntropyLoss()
        self.triplet_loss_fn = TreeTripletLoss(num_classes=len(hiera_map), hiera_map=hiera_map,
                                               hiera_index=hiera_index, ignore_index=ignore_index)
        self.loss_weight = loss_weight
        self.last_terms = None

    def forward(self, step, embedding, cls_score_before, cls_score, label, weight=None, **kwargs):
        """``cls_score`` is [B, n_fine+n_coarse, h, w] with (h, w) either the label size (reference usage) or any
        lower resolution (the bilinear resize to the label grid is then fused).  ``cls_score_before``, ``weight``
        are accepted and ignored, as in the reference (``hiera_triplet_loss.py:163``)."""
        if cls_score.shape[1] != self.num_classes + len(self.hiera_index):
            raise ValueError("cls_score must have n_fine + n_coarse channels")
        return _Hiera2Fn.apply(cls_score, embedding, ops.labels_u8(label), self, _step_value(step))
