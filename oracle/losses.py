"""Loss functions (oracle).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Written from the formulas in SURVEY.md Appendix A (A.4-A.7); every function names the
reference site it restates.  All of these are pinned by golden vectors G2, G4-G8
(``tests/test_oracle_golden.py``).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

IGNORE = 255


# ----------------------------------------------------------------------------- targets
def prepare_targets_two_level(targets, hiera_index):
    """Reference ``models/loss/hiera_triplet_loss.py:11-38``: coarse id = the bucket ``i``
    with ``start_i <= fine < end_i``; pixels in no bucket (255 included) stay 255."""
    coarse = torch.full_like(targets, IGNORE)
    for i, (s, e) in enumerate(hiera_index):
        coarse = torch.where((targets >= s) & (targets < e), torch.full_like(targets, i), coarse)
    return targets, coarse


def prepare_targets_three_level(targets, fine_to_mid, fine_to_high):
    """Reference ``models/loss/rmi_hiera_triplet_loss.py:21-63``: gather through the two maps
    where the fine label is not 255."""
    valid = targets != IGNORE
    safe = torch.where(valid, targets, torch.zeros_like(targets))
    mid = torch.where(valid, fine_to_mid.to(targets.device)[safe], torch.full_like(targets, IGNORE))
    high = torch.where(valid, fine_to_high.to(targets.device)[safe], torch.full_like(targets, IGNORE))
    return targets, mid, high


# ----------------------------------------------------------------------------- CE wrapper
def cross_entropy_allpix_mean(pred, label, ignore_index=IGNORE):
    """Reference ``models/loss/cross_entropy_loss.py:7-30`` + ``utils.py:6-55``:
    per-pixel CE with ``reduction='none'`` (ignored pixels -> 0) followed by a plain ``.mean()``
    over ALL B*H*W pixels."""
    return F.cross_entropy(pred, label, reduction="none", ignore_index=ignore_index).mean()


class CrossEntropyLoss(nn.Module):
    """Reference ``models/loss/cross_entropy_loss.py:136-195`` (only the softmax branch is ever
    used on the hot path)."""

    def __init__(self, loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, cls_score, label, **kwargs):
        return self.loss_weight * cross_entropy_allpix_mean(cls_score, label)


# ----------------------------------------------------------------------------- 2-level BCE
def _onehot(t, n):
    return F.one_hot(torch.where(t == IGNORE, torch.zeros_like(t), t), n).permute(0, 3, 1, 2).float()


def losses_hiera_two_level(predictions, targets_fine, targets_coarse, n_fine, hiera_index, eps=1e-8):
    """Reference ``models/loss/hiera_triplet_loss.py:41-107``."""
    n_coarse = len(hiera_index)
    p = torch.sigmoid(predictions.float())
    s, t = p[:, :n_fine], p[:, n_fine:n_fine + n_coarse]
    # which coarse bucket does fine channel k belong to (only channels inside a bucket get the min)
    mcla = s.clone()
    mcmb = torch.empty_like(t)
    for i, (a, b) in enumerate(hiera_index):
        mcla[:, a:b] = torch.minimum(s[:, a:b], t[:, i:i + 1])
        mcmb[:, i] = torch.maximum(s[:, a:b].amax(dim=1), t[:, i]) if b > a else t[:, i]
    oh_f = _onehot(targets_fine, n_fine)
    oh_c = _onehot(targets_coarse, n_coarse)
    vf = (targets_fine != IGNORE).unsqueeze(1).float()
    vc = (targets_coarse != IGNORE).unsqueeze(1).float()
    lf = ((-oh_f * torch.log(mcla + eps) - (1 - oh_f) * torch.log(1 - s + eps)) * vf).sum()
    lc = ((-oh_c * torch.log(t + eps) - (1 - oh_c) * torch.log(1 - mcmb + eps)) * vc).sum()
    lf = lf / (vf.sum().clamp_min(1.0) * n_fine)
    lc = lc / (vc.sum().clamp_min(1.0) * n_coarse)
    return 5.0 * (lf + lc)


# ----------------------------------------------------------------------------- triplet
def _nearest_labels(labels, h, w):
    """``F.interpolate(mode='nearest')`` of the (float-cast) label map, exactly as the reference
    does it (``tree_triplet_loss.py:17-20``): src = min(floorf(dst * float(in)/out), in-1)."""
    return F.interpolate(labels.unsqueeze(1).float(), (h, w), mode="nearest").squeeze(1).long()


def _triplet_core(feats, labels, class_sets, max_triplet):
    """Shared body: for each (anchor_mask, pos_mask, neg_mask) take the first m rows of each in
    raster order, hinge(d_ap - d_an + 0.6).mean(); average over the classes with m > 0."""
    total, count = 0.0, 0
    for anchor, pos, neg in class_sets:
        m = min(int(anchor.sum()), int(pos.sum()), int(neg.sum()), max_triplet)
        if m == 0:
            continue
        fa, fp, fn = feats[anchor][:m], feats[pos][:m], feats[neg][:m]
        d_ap = 1 - (fa * fp).sum(1)
        d_an = 1 - (fa * fn).sum(1)
        total = total + F.relu(d_ap - d_an + 0.6).mean()
        count += 1
    if count == 0:
        return None, torch.tensor([0])
    return total / count, torch.tensor([count])


class TreeTripletLoss(nn.Module):
    """2-level tree-triplet.  Reference ``models/loss/tree_triplet_loss.py:15-65`` (minus its
    hard ``.cuda()`` calls)."""

    def __init__(self, num_classes, hiera_map, hiera_index, ignore_index=IGNORE):
        super().__init__()
        self.num_classes, self.hiera_map, self.hiera_index = num_classes, hiera_map, hiera_index
        self.ignore_label = ignore_index

    def forward(self, feats, labels=None, max_triplet=200):
        lab = _nearest_labels(labels, feats.shape[2], feats.shape[3]).reshape(-1)
        f = feats.permute(0, 2, 3, 1).reshape(-1, feats.shape[1])
        sets = []
        for ii in torch.unique(lab).tolist():
            if ii == IGNORE:
                continue
            s, e = self.hiera_index[self.hiera_map[ii]][0], self.hiera_index[self.hiera_map[ii]][-1]
            anchor = lab == ii
            inside = (lab >= s) & (lab < e)
            sets.append((anchor, inside & ~anchor, ~inside))
        return _triplet_core(f, lab, sets, max_triplet)


class RMITreeTripletLoss(nn.Module):
    """3-level tree-triplet.  Reference ``models/loss/rmi_tree_triplet_loss.py:14-70``: classes
    255 and 0 are skipped; positives = the other ids of the anchor's hard-coded group, negatives
    = the other group; an id outside both groups raises ValueError (as ``list.remove`` does)."""

    def __init__(self, num_classes, upper_ids, lower_ids, ignore_index=IGNORE):
        super().__init__()
        self.num_classes, self.upper_ids, self.lower_ids = num_classes, upper_ids, lower_ids
        self.ignore_label = ignore_index

    def forward(self, feats, labels=None, max_triplet=200):
        lab = _nearest_labels(labels, feats.shape[2], feats.shape[3]).reshape(-1)
        f = feats.permute(0, 2, 3, 1).reshape(-1, feats.shape[1])
        sets = []
        for ii in torch.unique(lab).tolist():
            if ii == IGNORE or ii == 0:
                continue
            if ii in self.upper_ids:
                pos_ids, neg_ids = list(self.upper_ids), list(self.lower_ids)
            else:
                pos_ids, neg_ids = list(self.lower_ids), list(self.upper_ids)
            pos_ids.remove(ii)          # ValueError for ids outside both groups, like the reference
            isin = lambda ids: torch.isin(lab, torch.tensor(ids, dtype=lab.dtype))
            sets.append((lab == ii, isin(pos_ids), isin(neg_ids)))
        return _triplet_core(f, lab, sets, max_triplet)


def triplet_factor(step, total_steps):
    """Cosine ramp of reference ``hiera_triplet_loss.py:203-208`` / ``rmi_hiera_triplet_loss.py:537-542``."""
    if step < total_steps:
        return 0.25 * (1 + math.cos((step - total_steps) / total_steps * math.pi))
    return 0.5


# ----------------------------------------------------------------------------- 2-level loss
class HieraTripletLoss(nn.Module):
    """Reference ``models/loss/hiera_triplet_loss.py:110-211``.  ``cls_score_before``,
    ``weight`` and ``use_sigmoid`` are accepted and ignored, as in the reference."""

    def __init__(self, num_classes, hiera_map, hiera_index, ignore_index=IGNORE,
                 use_sigmoid=False, loss_weight=1.0):
        super().__init__()
        self.num_classes, self.hiera_map, self.hiera_index = num_classes, hiera_map, hiera_index
        self.ignore_index, self.loss_weight = ignore_index, loss_weight
        self.ce = CrossEntropyLoss()
        self.triplet_loss_fn = TreeTripletLoss(len(hiera_map), hiera_map, hiera_index, ignore_index)

    def forward(self, step, embedding, cls_score_before, cls_score, label, weight=None, **kwargs):
        nf, nc = self.num_classes, len(self.hiera_index)
        tf, tc = prepare_targets_two_level(label, self.hiera_index)
        loss = losses_hiera_two_level(cls_score, tf, tc, nf, self.hiera_index)
        loss = loss + self.ce(cls_score[:, :nf], tf) + self.ce(cls_score[:, nf:nf + nc], tc)
        trip, count = self.triplet_loss_fn(embedding, label)
        # hiera_triplet_loss.py:192-201: single process: ready <=> class_count > 0; under torch.distributed: every rank's count > 0
        # (the caller states that by passing `ready`, oracle/step.py:ddp_forward_loss)
        ready = kwargs.get("ready")
        if (int(count) > 0) if ready is None else ready:
            loss = loss + triplet_factor(int(step), 80000) * trip
        return loss * self.loss_weight


# ----------------------------------------------------------------------------- 3-level + RMI
_CLIP_MIN = 1e-6
_POS_ALPHA = 1e-3


def losses_hiera_three_level(probs, tf, tm, th, n_fine, n_mid, n_high, fine_to_mid, fine_to_high):
    """3-level BCE of reference ``rmi_hiera_triplet_loss.py:352-470``."""
    f2m, f2h = fine_to_mid.tolist(), fine_to_high.tolist()
    A, B_, C = probs[:, :n_fine], probs[:, n_fine:n_fine + n_mid], probs[:, n_fine + n_mid:n_fine + n_mid + n_high]
    mcmb = B_.clone()      # max(max_{f in m} A_f, B_m)
    mclb = B_.clone()      # min(B_m, min_{f in m} C_{high(f)})
    for m in range(n_mid):
        fs = [f for f in range(n_fine) if f2m[f] == m]
        if fs:
            mcmb[:, m] = torch.maximum(A[:, fs].amax(1), B_[:, m])
            mclb[:, m] = torch.minimum(C[:, sorted({f2h[f] for f in fs})].amin(1), B_[:, m])
    mcmc = C.clone()       # max(max_{m in j} mcmb_m, C_j)
    for j in range(n_high):
        ms = sorted({f2m[f] for f in range(n_fine) if f2h[f] == j})
        if ms:
            mcmc[:, j] = torch.maximum(mcmb[:, ms].amax(1), C[:, j])
    mcla = torch.minimum(A, B_[:, f2m])
    oh_f, oh_m, oh_h = _onehot(tf, n_fine), _onehot(tm, n_mid), _onehot(th, n_high)
    out = 0.0
    for oh, lo, hi, t, n in ((oh_f, mcla, A, tf, n_fine), (oh_m, mclb, mcmb, tm, n_mid), (oh_h, C, mcmc, th, n_high)):
        v = (t != IGNORE).unsqueeze(1).float()
        term = ((-oh * torch.log(lo + _CLIP_MIN) - (1 - oh) * torch.log(1 - hi + _CLIP_MIN)) * v).sum()
        out = out + term / (v.sum().clamp_min(1.0) * n)
    return 5.0 * out, (oh_f, oh_m, oh_h)


def rmi_lower_bound(onehot_all, probs_masked, radius=3):
    """RMI lower bound of reference ``rmi_hiera_triplet_loss.py:292-317, 479-517``.
    Returns (loss scalar f32, per-(b,c) rmi values f64)."""
    b, c, h, w = onehot_all.shape
    nh, nw = h - (radius - 1), w - (radius - 1)
    shifts = [(y, x) for y in range(radius) for x in range(radius)]
    la = torch.stack([onehot_all[:, :, y:y + nh, x:x + nw] for y, x in shifts], 2).reshape(b, c, radius * radius, -1).double()
    pr = torch.stack([probs_masked[:, :, y:y + nh, x:x + nw] for y, x in shifts], 2).reshape(b, c, radius * radius, -1).double()
    la = la.detach()
    eye = (torch.eye(radius * radius, dtype=torch.float32) * _POS_ALPHA).double()   # f32 1e-3 promoted, as the reference
    s_ll = la @ la.transpose(2, 3)
    s_pp = pr @ pr.transpose(2, 3)
    s_lp = la @ pr.transpose(2, 3)
    va = s_ll - s_lp @ torch.inverse(s_pp + eye) @ s_lp.transpose(2, 3)
    chol = torch.linalg.cholesky(va + eye)
    rmi = 0.5 * 2.0 * torch.log(torch.diagonal(chol, dim1=-2, dim2=-1) + 1e-8).sum(-1)   # [b, c] f64
    per_class = rmi.mean(0).float() / float(radius * radius)
    return per_class.sum(), rmi


class RMIHieraTripletLoss(nn.Module):
    """Reference ``models/loss/rmi_hiera_triplet_loss.py:180-546``.  ``rmi_pool_*`` are
    accepted, asserted equal and never applied."""

    def __init__(self, n_fine, n_mid, n_high, fine_to_mid, fine_to_high, rmi_radius=3, rmi_pool_way=0,
                 rmi_pool_size=3, rmi_pool_stride=3, loss_weight_lambda=0.5, loss_weight=1.0,
                 ignore_index=IGNORE):
        super().__init__()
        assert fine_to_mid.dtype == torch.long and fine_to_high.dtype == torch.long
        assert fine_to_mid.numel() == n_fine and fine_to_high.numel() == n_fine
        assert rmi_pool_size == rmi_pool_stride
        self.n_fine, self.n_mid, self.n_high = n_fine, n_mid, n_high
        self.fine_to_mid, self.fine_to_high = fine_to_mid.clone(), fine_to_high.clone()
        self.rmi_radius, self.loss_weight_lambda, self.loss_weight = rmi_radius, loss_weight_lambda, loss_weight
        self.ignore_index = ignore_index
        if n_fine > 15:
            self.upper_ids = [1, 2, 3, 4, 5, 6, 7, 10, 11, 13, 14, 15]
            self.lower_ids = [8, 9, 12, 16, 17, 18, 19]
        else:
            self.upper_ids, self.lower_ids = [1, 2, 3, 4], [5, 6]
        self.ce = CrossEntropyLoss()
        self.triplet_loss = RMITreeTripletLoss(n_fine, self.upper_ids, self.lower_ids, ignore_index)

    def forward(self, step, embedding, cls_score_before, cls_score, label, weight=None, **kwargs):
        nf, nm, nh = self.n_fine, self.n_mid, self.n_high
        tf, tm, th = prepare_targets_three_level(label, self.fine_to_mid, self.fine_to_high)
        probs = torch.sigmoid(cls_score.float())
        hiera, (oh_f, oh_m, oh_h) = losses_hiera_three_level(
            probs, tf, tm, th, nf, nm, nh, self.fine_to_mid, self.fine_to_high)
        onehot_all = torch.cat([oh_f, oh_m, oh_h], 1)
        valid_all = torch.cat([(t != IGNORE).unsqueeze(1).float().expand(-1, n, -1, -1)
                               for t, n in ((tf, nf), (tm, nm), (th, nh))], 1)
        rmi, _ = rmi_lower_bound(onehot_all, probs * valid_all + _CLIP_MIN, self.rmi_radius)
        loss = self.loss_weight_lambda * rmi + 0.5 * hiera
        loss = loss + self.ce(cls_score[:, :nf], tf) + self.ce(cls_score[:, nf:nf + nm], tm) \
            + self.ce(cls_score[:, nf + nm:nf + nm + nh], th)
        trip, count = self.triplet_loss(embedding, label)
        ready = kwargs.get("ready")                 # as HieraTripletLoss (rmi_hiera_triplet_loss.py:527-536)
        if (int(count) > 0) if ready is None else ready:
            loss = loss + triplet_factor(int(step), 160000 if nf > 15 else 60000) * trip
        return loss * self.loss_weight


# ----------------------------------------------------------------------------- metrics
def pixel_accuracy_counts(fine_logits, label, ignore_index=IGNORE):
    """Reference ``train.py:37-49, 381-385``: (#correct, #valid) over label != 255."""
    pred = fine_logits.argmax(1)
    valid = label != ignore_index
    return int(((pred == label) & valid).sum()), int(valid.sum())


def confusion_matrix(fine_logits, label, n_fine, ignore_index=IGNORE):
    """Build-defined (the reference has no IoU): rows = ground truth, cols = argmax."""
    pred = fine_logits.argmax(1)
    valid = label != ignore_index
    idx = label[valid] * n_fine + pred[valid]
    return torch.bincount(idx, minlength=n_fine * n_fine).reshape(n_fine, n_fine)


def miou_from_confusion(cm):
    tp = cm.diag().double()
    denom = cm.sum(0).double() + cm.sum(1).double() - tp
    present = cm.sum(1) > 0
    return float((tp[present] / denom[present].clamp_min(1)).mean()) if bool(present.any()) else 0.0
