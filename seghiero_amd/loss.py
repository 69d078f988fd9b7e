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
        _, sums, _ = ops.ce_fwd(logits, label8)
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
        from . import ddp
        norm = ddp.current_counts()              # exact data-parallel mode: global normalisers (ddp.exact_counts), else None
        main, sums, _, gw = ops.hiera2_fwd(logits, label8, nf, hidx, want_grad=ctx.needs_input_grad[0], norm=norm)
        masks, ok = mod.triplet_loss_fn.tables(logits.device)
        trip, ws = ops.triplet_fwd(emb, label8, masks, ok, 200, 0.6)
        ready = None
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # hiera_triplet_loss.py:193-198: the term counts only if EVERY rank produced triplets
            ready = trip[1:2].clone()
            ddp.all_reduce_small(ready, op=torch.distributed.ReduceOp.MIN)
        factor = triplet_factor(step, 80000)
        if norm is not None:
            factor = factor / ddp.world_size()   # per-rank triplets (tree_triplet_loss.py:23-46), mean over the ranks; the rest sums
        total = ops.combine_loss(main, trip, ready, factor, mod.loss_weight)
        ctx.save_for_backward(logits, emb, label8, sums, trip, ws)
        ctx.grad_ws = gw            # per-pixel gradient left by the forward (None: the backward recomputes it)
        ctx.ready = ready
        ctx.cfg = (nf, hidx, factor, mod.loss_weight)
        mod.last_terms = (main, trip)
        return total.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, emb, label8, sums, trip, ws = ctx.saved_tensors
        nf, hidx, factor, lw = ctx.cfg
        g = g.reshape(1).float()
        dlogits = ops.hiera2_bwd(logits, label8, nf, hidx, sums, g, lw, grad_ws=ctx.grad_ws) if ctx.needs_input_grad[0] else None
        ctx.grad_ws = None
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


# ----------------------------------------------------------------------------- 3-level RMI loss
def three_level_triplet_tables(upper_ids, lower_ids):
    """Per anchor class ii (rmi_tree_triplet_loss.py:28-47): positives = the other ids of ii's hard-coded group,
    negatives = the other group; classes 0 and 255 are never anchors.  Deviation: a label outside both groups makes
    the reference raise ValueError from ``list.remove`` (:39); here such a class is simply not an anchor."""
    masks = torch.zeros((256, 2, 4), dtype=torch.int64)
    ok = []
    for group, other in ((upper_ids, lower_ids), (lower_ids, upper_ids)):
        for ii in group:
            if ii == 0 or ii == 255:
                continue
            masks[ii, 0] = torch.tensor(_bitset([v for v in group if v != ii]))
            masks[ii, 1] = torch.tensor(_bitset(list(other)))
            ok.append(ii)
    return masks, torch.tensor(_bitset(ok), dtype=torch.int64)


class RMITreeTripletLoss(nn.Module):
    """``TreeTripletLoss`` of reference ``models/loss/rmi_tree_triplet_loss.py:5-70`` (3-level variant)."""

    def __init__(self, num_classes, upper_ids, lower_ids, ignore_index=IGNORE, strict=False):
        """strict=True mirrors the reference's failure mode: a label present on the embedding grid that is in neither hard-coded
        group (nor 0 / 255) raises ``ValueError`` as ``list.remove`` does at ``rmi_tree_triplet_loss.py:39`` -- at the price of one
        host synchronisation per call.  Default (False): such a class is simply never an anchor and nothing synchronises."""
        super().__init__()
        self.ignore_label = ignore_index
        self.num_classes = num_classes
        self.upper_ids = upper_ids
        self.lower_ids = lower_ids
        self.strict = strict
        masks, ok = three_level_triplet_tables(upper_ids, lower_ids)
        self.register_buffer("_masks", masks, persistent=False)
        self.register_buffer("_anchor_ok", ok, persistent=False)

    tables = TreeTripletLoss.tables

    def check_labels(self, labels, h, w):
        """The reference's ValueError for a label outside both groups (``:34-39``): labels as its nearest-neighbour resize sees
        them (index = floor(i * H / h))."""
        H, W = labels.shape[-2:]
        iy = (torch.arange(h, device=labels.device) * H) // h
        ix = (torch.arange(w, device=labels.device) * W) // w
        present = torch.unique(labels[:, iy][:, :, ix]).tolist()
        known = set(self.upper_ids) | set(self.lower_ids) | {0, self.ignore_label}
        for v in present:
            if int(v) not in known:
                raise ValueError("list.remove(x): x not in list")      # the reference's own message (label %d in neither group)

    def forward(self, feats, labels=None, max_triplet=200):
        if self.strict:
            self.check_labels(labels, feats.shape[2], feats.shape[3])
        masks, ok = self.tables(feats.device)
        loss, count = _TripletFn.apply(feats, ops.labels_u8(labels), masks, ok, max_triplet)
        n = int(count.item())
        cnt = torch.tensor([n], device=feats.device)
        return (None, cnt) if n == 0 else (loss, cnt)


def prepare_targets_three_level(targets, fine_to_mid, fine_to_high):
    """``_prepare_targets_three_level`` of reference ``rmi_hiera_triplet_loss.py:21-63`` evaluated by the HIP loss kernel
    (the same code path ``RMIHieraTripletLoss.forward`` uses internally): -> (fine, mid, high) int64 maps, 255 = ignore."""
    l8 = ops.labels_u8(targets)
    f2m = [int(v) for v in fine_to_mid.tolist()]
    f2h = [int(v) for v in fine_to_high.tolist()]
    nf, nm, nh = len(f2m), max(f2m) + 1, max(f2h) + 1
    z = ops.new_act(l8.shape[0], nf + nm + nh, 1, 1, l8.device, zero=True)
    _, _, _, (mid, high) = ops.hiera3_fwd(z, l8, nf, nm, nh, f2m, f2h, want_probs=False, want_targets=True)
    return targets, mid.long(), high.long()


class _Hiera3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_score, embedding, label8, mod, step):
        logits = ops.to_nhwc(cls_score)
        emb = _dense_nhwc(embedding)
        nf, nm, nh = mod.n_fine, mod.n_mid, mod.n_high
        f2m, f2h = mod._f2m, mod._f2h
        lam, lw = mod.loss_weight_lambda, mod.loss_weight
        need_grad = ctx.needs_input_grad[0]
        ctx.grad_ws = (None, 0)                                    # (workspace, row stride): the forward's per-pixel gradient
        if need_grad:
            rest, sums, probs, ctx.grad_ws = ops.hiera3_fwd(logits, label8, nf, nm, nh, f2m, f2h, want_probs=True, want_grad=True)
        else:
            rest, sums, probs = ops.hiera3_fwd(logits, label8, nf, nm, nh, f2m, f2h, want_probs=True)
        rmi, dprob = ops.rmi_loss(probs, label8, nf, nm, nh, f2m, f2h, want_grad=need_grad)
        ctx.probs = probs if (need_grad and ctx.grad_ws[0] is not None) else None
        main = ops.scalar_axpy(rest, rmi, lam)
        masks, ok = mod.triplet_loss.tables(logits.device)
        trip, ws = ops.triplet_fwd(emb, label8, masks, ok, 200, 0.6)
        ready = None
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            from . import ddp
            ready = trip[1:2].clone()
            ddp.all_reduce_small(ready, op=torch.distributed.ReduceOp.MIN)
        factor = triplet_factor(step, 160000 if nf > 15 else 60000)
        total = ops.combine_loss(main, trip, ready, factor, lw)
        ctx.save_for_backward(logits, emb, label8, sums, trip, ws, dprob if dprob is not None else sums)
        ctx.has_dprob = dprob is not None
        ctx.ready = ready
        ctx.cfg = (nf, nm, nh, f2m, f2h, lam, lw, factor, logits.shape[0])
        mod.last_terms = (rest, rmi, trip)
        return total.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, emb, label8, sums, trip, ws, dprob = ctx.saved_tensors
        nf, nm, nh, f2m, f2h, lam, lw, factor, batch = ctx.cfg
        g = g.reshape(1).float()
        dlogits = None
        if ctx.needs_input_grad[0]:
            dlogits = ops.hiera3_bwd(logits, label8, nf, nm, nh, f2m, f2h, sums, dprob if ctx.has_dprob else None,
                                     lam / (9.0 * batch), g, lw, grad_ws=ctx.grad_ws, probs=ctx.probs)
            ctx.probs = None
        demb = None
        if ctx.needs_input_grad[1]:
            gt = g if ctx.ready is None else g * (ctx.ready > 0).float()
            demb = ops.triplet_bwd(emb, ws, trip, gt, factor * lw)
        return dlogits, demb, None, None, None


class RMIHieraTripletLoss(nn.Module):
    """Drop-in for reference ``models/loss/rmi_hiera_triplet_loss.py:180-546`` (same constructor, same forward).
    ``rmi_pool_way / rmi_pool_size / rmi_pool_stride`` are accepted, asserted equal and never applied -- exactly like
    the reference; ``rmi_radius`` must be 3 (the only value the kernels implement)."""

    def __init__(self, n_fine: int, n_mid: int, n_high: int, fine_to_mid: torch.Tensor, fine_to_high: torch.Tensor,
                 rmi_radius: int = 3, rmi_pool_way: int = 0, rmi_pool_size: int = 3, rmi_pool_stride: int = 3,
                 loss_weight_lambda: float = 0.5, loss_weight: float = 1.0, ignore_index: int = IGNORE):
        super().__init__()
        assert fine_to_mid.dtype == torch.long
        assert fine_to_high.dtype == torch.long
        assert fine_to_mid.numel() == n_fine
        assert fine_to_high.numel() == n_fine
        assert rmi_pool_size == rmi_pool_stride
        if rmi_radius != 3 or ignore_index != IGNORE:
            raise SegHieroHipError("the RMI kernels implement rmi_radius == 3 and ignore_index == 255")
        self.n_fine, self.n_mid, self.n_high = n_fine, n_mid, n_high
        self.fine_to_mid, self.fine_to_high = fine_to_mid.clone(), fine_to_high.clone()
        self._f2m, self._f2h = fine_to_mid.tolist(), fine_to_high.tolist()
        self.ignore_index = ignore_index
        self.rmi_radius, self.rmi_pool_way = rmi_radius, rmi_pool_way
        self.rmi_pool_size, self.rmi_pool_stride = rmi_pool_size, rmi_pool_stride
        if n_fine > 15:
            self.upper_ids = [1, 2, 3, 4, 5, 6, 7, 10, 11, 13, 14, 15]
            self.lower_ids = [8, 9, 12, 16, 17, 18, 19]
        else:
            self.upper_ids, self.lower_ids = [1, 2, 3, 4], [5, 6]
        self.loss_weight_lambda, self.loss_weight = loss_weight_lambda, loss_weight
        self.half_d = rmi_radius * rmi_radius
        self.d = 2 * self.half_d
        self.kernel_padding = rmi_pool_size // 2
        self.ce = CrossEntropyLoss()
        self.triplet_loss = RMITreeTripletLoss(num_classes=n_fine, upper_ids=self.upper_ids, lower_ids=self.lower_ids,
                                               ignore_index=ignore_index)
        self.last_terms = None

    def forward(self, step, embedding, cls_score_before, cls_score, label, weight=None, **kwargs):
        if cls_score.shape[1] != self.n_fine + self.n_mid + self.n_high:
            raise ValueError("cls_score must have n_fine + n_mid + n_high channels")
        if self.triplet_loss.strict:
            self.triplet_loss.check_labels(label, embedding.shape[2], embedding.shape[3])
        return _Hiera3Fn.apply(cls_score, embedding, ops.labels_u8(label), self, _step_value(step))
