#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE on CPU.

Runs only in the authoring container (needs /root/reference); the produced ``.npz`` files hold
arrays only -- inputs and the reference's outputs -- never reference source.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py

Harness shims (test harness only, none of this ships):
* ``torch.Tensor.cuda`` -> identity, because both reference TreeTripletLoss classes hard-code
  ``.cuda()`` (``models/loss/tree_triplet_loss.py:48,54,63,65``) and this box has no GPU;
* empty stub modules for ``torchvision`` / ``terminaltables`` / ``dataset.dataloader`` so that
  ``import train`` succeeds and its pure helper functions (``train.py:37-99``) can be called;
* ``torch.cholesky`` is still present in torch 2.10 (deprecated) -- used as is.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")
torch.Tensor.cuda = lambda self, *a, **k: self
for name in ("torchvision", "torchvision.models", "torchvision.transforms", "terminaltables",
             "dataset", "dataset.dataloader"):
    m = types.ModuleType(name)
    sys.modules.setdefault(name, m)
sys.modules["terminaltables"].AsciiTable = object
sys.modules["dataset.dataloader"].HieroDataloader = object
sys.modules["torchvision"].models = sys.modules["torchvision.models"]

import train as ref_train  # noqa: E402
from models.head.sep_aspp_contrast_head import DepthwiseSeparableASPPContrastHead  # noqa: E402
from models.loss import hiera_triplet_loss as ref_h2  # noqa: E402
from models.loss import rmi_hiera_triplet_loss as ref_h3  # noqa: E402
from models.loss.cross_entropy_loss import CrossEntropyLoss  # noqa: E402
from models.loss.tree_triplet_loss import TreeTripletLoss as RefTriplet2  # noqa: E402
from models.loss.rmi_tree_triplet_loss import TreeTripletLoss as RefTriplet3  # noqa: E402

torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrs)} arrays)")


def blocky_labels(g, b, h, w, n_fine, cell=8, p_ignore=0.05, border=2):
    """Blocky label map (so every bucket has anchors/positives/negatives) with 255 speckle + border."""
    hh, ww = -(-h // cell), -(-w // cell)
    small = torch.randint(0, n_fine, (b, hh, ww), generator=g)
    lab = small.repeat_interleave(cell, 1).repeat_interleave(cell, 2)[:, :h, :w].clone()
    lab[torch.rand(b, h, w, generator=g) < p_ignore] = 255
    if border:
        lab[:, :border] = 255
        lab[:, -border:] = 255
        lab[:, :, :border] = 255
        lab[:, :, -border:] = 255
    return lab.long()


# ------------------------------------------------------------------ G1 mapping helpers
def g1():
    maps = {
        "a": ([[0, 3], [4, 6], [7], [8]], 9),          # README 2-level example
        "b": ([[0, 1], [2, 3]], 4),                      # BASELINE config 1
        "c": ([[0], [1, 4], [5, 6]], 7),                 # config 4 fine->mid
    }
    out = {}
    for k, (cfg, nf) in maps.items():
        out[f"{k}_f2c"] = npy(ref_train.build_fine_to_coarse_map(cfg, nf))
        out[f"{k}_hidx"] = np.asarray(ref_train.build_hiera_index(cfg), dtype=np.int64)
    out["c_f2s"] = npy(ref_train.build_fine_to_super_map([[0], [1, 6]], 7))   # covers every fine id
    save("g1_maps", **out)


# ------------------------------------------------------------------ G2 target preparation
def g2():
    g = torch.Generator().manual_seed(2)
    out = {}
    for tag, (h, w) in {"even": (64, 64), "odd": (75, 51)}.items():
        lab9 = blocky_labels(g, 2, h, w, 9)
        _, coarse, _ = ref_h2._prepare_targets_two_level(lab9, [[0, 4], [4, 7], [7, 8], [8, 9]])
        lab7 = blocky_labels(g, 2, h, w, 7)
        f2m = torch.tensor([0, 1, 1, 1, 1, 2, 2])
        f2h = torch.tensor([0, 1, 1, 1, 1, 1, 1])
        _, mid, high = ref_h3._prepare_targets_three_level(lab7, f2m, f2h)
        out.update({f"{tag}_lab9": npy(lab9).astype(np.uint8), f"{tag}_coarse": npy(coarse).astype(np.uint8),
                    f"{tag}_lab7": npy(lab7).astype(np.uint8), f"{tag}_mid": npy(mid).astype(np.uint8),
                    f"{tag}_high": npy(high).astype(np.uint8)})
    # a hiera_index that leaves fine id 3 outside every bucket -> stays 255
    lab = blocky_labels(g, 1, 16, 16, 5, cell=4)
    _, coarse, _ = ref_h2._prepare_targets_two_level(lab, [[0, 3], [4, 5]])
    out.update({"gap_lab": npy(lab).astype(np.uint8), "gap_coarse": npy(coarse).astype(np.uint8)})
    save("g2_targets", **out)


# ------------------------------------------------------------------ G3 reduced-width head
HEAD_KW = dict(in_channels=64, c1_in_channels=16, c1_channels=8, aspp_channels=16,
               dilations=(1, 12, 24, 36), num_classes=6, proj_dim=8, proj_type="convmlp")


def g3():
    torch.manual_seed(3)
    head = DepthwiseSeparableASPPContrastHead(**HEAD_KW)
    # non-trivial BN affine so gamma/beta grads are exercised
    g = torch.Generator().manual_seed(33)
    with torch.no_grad():
        for m in head.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
    sd0 = {k: v.clone() for k, v in head.state_dict().items()}
    out = {"sd__" + k: npy(v) for k, v in sd0.items()}
    cases = {"A": ((32, 32), (4, 4)), "B": ((19, 13), (3, 2)), "C": ((24, 24), (16, 16))}
    for tag, (s1, s4) in cases.items():
        head.load_state_dict(sd0)
        head.train()
        c1 = torch.randn(2, 16, *s1, generator=g).requires_grad_(True)
        c4 = torch.randn(2, 64, *s4, generator=g).requires_grad_(True)
        logits, emb = head([c1, None, None, c4])
        gl = torch.randn(logits.shape, generator=g)
        ge = torch.randn(emb.shape, generator=g)
        head.zero_grad()
        (logits * gl).sum().add((emb * ge).sum()).backward()
        out.update({f"{tag}_c1": npy(c1), f"{tag}_c4": npy(c4), f"{tag}_logits": npy(logits),
                    f"{tag}_emb": npy(emb), f"{tag}_gl": npy(gl), f"{tag}_ge": npy(ge),
                    f"{tag}_dc1": npy(c1.grad), f"{tag}_dc4": npy(c4.grad)})
        for k, p in head.named_parameters():
            out[f"{tag}_grad__{k}"] = npy(p.grad)
        for k, v in head.state_dict().items():
            if "running" in k or k == "step" or "num_batches" in k:
                out[f"{tag}_after__{k}"] = npy(v)
        head.eval()
        with torch.no_grad():
            le, ee = head([c1, None, None, c4])
        out.update({f"{tag}_logits_eval": npy(le), f"{tag}_emb_eval": npy(ee)})
    save("g3_head", **out)


# ------------------------------------------------------------------ G4 2-level pieces
HIDX2 = [[0, 2], [2, 4]]
HMAP2 = [0, 0, 1, 1]


def g4():
    g = torch.Generator().manual_seed(4)
    out = {}
    for tag, (h, w) in {"even": (64, 64), "odd": (75, 51)}.items():
        lab = blocky_labels(g, 2, h, w, 4)
        z = (2.0 * torch.randn(2, 6, h, w, generator=g)).requires_grad_(True)
        tf, tc, _ = ref_h2._prepare_targets_two_level(lab, HIDX2)
        lh = ref_h2._losses_hiera_two_level(z, tf, tc, 4, HIDX2)
        lh.backward()
        ce = CrossEntropyLoss()
        z2 = z.detach().clone().requires_grad_(True)
        lce_f = ce(z2[:, :4], tf)
        lce_c = ce(z2[:, 4:6], tc)
        (lce_f + lce_c).backward()
        out.update({f"{tag}_lab": npy(lab).astype(np.uint8), f"{tag}_z": npy(z), f"{tag}_hiera": npy(lh),
                    f"{tag}_dz_hiera": npy(z.grad), f"{tag}_ce_f": npy(lce_f), f"{tag}_ce_c": npy(lce_c),
                    f"{tag}_dz_ce": npy(z2.grad)})
    # triplet: normal, all-255, singleton bucket (no positives for class 2 when bucket [2,3) is alone)
    trip = RefTriplet2(num_classes=4, hiera_map=HMAP2, hiera_index=HIDX2)
    lab = blocky_labels(g, 2, 64, 64, 4)
    emb = torch.nn.functional.normalize(torch.randn(2, 8, 8, 8, generator=g), dim=1).requires_grad_(True)
    val, cnt = trip(emb, lab)
    val.backward()
    out.update({"trip_lab": npy(lab).astype(np.uint8), "trip_emb": npy(emb), "trip_val": npy(val),
                "trip_cnt": npy(cnt), "trip_demb": npy(emb.grad)})
    val0, cnt0 = trip(emb.detach(), torch.full((2, 64, 64), 255, dtype=torch.long))
    assert val0 is None
    out["trip_void_cnt"] = npy(cnt0)
    trip_s = RefTriplet2(num_classes=3, hiera_map=[0, 0, 1], hiera_index=[[0, 2], [2, 3]])
    lab_s = blocky_labels(g, 2, 64, 64, 3)
    emb_s = torch.nn.functional.normalize(torch.randn(2, 8, 8, 8, generator=g), dim=1).requires_grad_(True)
    val_s, cnt_s = trip_s(emb_s, lab_s)
    val_s.backward()
    out.update({"trips_lab": npy(lab_s).astype(np.uint8), "trips_emb": npy(emb_s), "trips_val": npy(val_s),
                "trips_cnt": npy(cnt_s), "trips_demb": npy(emb_s.grad)})
    # more than 200 anchors per class: 2 x 32 x 32 grid, 4 classes
    lab_l = blocky_labels(g, 2, 128, 128, 4, cell=16, border=0)
    emb_l = torch.nn.functional.normalize(torch.randn(2, 8, 32, 32, generator=g), dim=1).requires_grad_(True)
    val_l, cnt_l = trip(emb_l, lab_l)
    val_l.backward()
    out.update({"tripl_lab": npy(lab_l).astype(np.uint8), "tripl_emb": npy(emb_l), "tripl_val": npy(val_l),
                "tripl_cnt": npy(cnt_l), "tripl_demb": npy(emb_l.grad)})
    save("g4_two_level_parts", **out)


# ------------------------------------------------------------------ G5 full 2-level loss
def g5():
    g = torch.Generator().manual_seed(5)
    out = {}
    loss_fn = ref_h2.HieraTripletLoss(num_classes=4, hiera_map=HMAP2, hiera_index=HIDX2, loss_weight=1.0)
    for tag, (h, w, eh, ew) in {"even": (64, 64, 8, 8), "odd": (75, 51, 5, 4)}.items():
        lab = blocky_labels(g, 2, h, w, 4)
        z0 = 2.0 * torch.randn(2, 6, h, w, generator=g)
        e0 = torch.nn.functional.normalize(torch.randn(2, 8, eh, ew, generator=g), dim=1)
        out.update({f"{tag}_lab": npy(lab).astype(np.uint8), f"{tag}_z": npy(z0), f"{tag}_emb": npy(e0)})
        for step in (0, 40000, 80000):
            z = z0.clone().requires_grad_(True)
            e = e0.clone().requires_grad_(True)
            before = torch.zeros(2, 4, h // 8, w // 8)
            val = loss_fn(torch.tensor([step]), e, before, z, lab)
            val.backward()
            out.update({f"{tag}_s{step}_loss": npy(val), f"{tag}_s{step}_dz": npy(z.grad),
                        f"{tag}_s{step}_demb": npy(e.grad)})
    save("g5_hiera_triplet_loss", **out)


def _rec(orig, seen, m):
    out = orig(m)
    seen.append(out.detach().clone())
    return out


# ------------------------------------------------------------------ G6/G7 3-level RMI loss + triplet
def g6():
    g = torch.Generator().manual_seed(6)
    f2m = torch.tensor([0, 1, 1, 1, 1, 2, 2])
    f2h = torch.tensor([0, 1, 1, 1, 1, 1, 1])
    out = {}
    for lam in (0.0, 0.5):
        loss_fn = ref_h3.RMIHieraTripletLoss(7, 3, 2, f2m, f2h, loss_weight_lambda=lam)
        # record rmi_now = 0.5 * log_det_by_cholesky(...) per (image, channel) in f64 (:509-513): harness-side wrapper of the
        # bound method, the reference code itself runs unchanged
        seen = []
        orig_logdet = loss_fn.log_det_by_cholesky
        loss_fn.log_det_by_cholesky = lambda m, _o=orig_logdet, _s=seen: _rec(_o, _s, m)
        for tag, (h, w, eh, ew) in {"even": (64, 64, 8, 8), "odd": (45, 37, 5, 4)}.items():
            gg = torch.Generator().manual_seed(60 + h)
            lab = blocky_labels(gg, 2, h, w, 7)
            z0 = 2.0 * torch.randn(2, 12, h, w, generator=gg)
            e0 = torch.nn.functional.normalize(torch.randn(2, 8, eh, ew, generator=gg), dim=1)
            out.update({f"{tag}_lab": npy(lab).astype(np.uint8), f"{tag}_z": npy(z0), f"{tag}_emb": npy(e0)})
            for step in (0, 30000):
                z = z0.clone().requires_grad_(True)
                e = e0.clone().requires_grad_(True)
                val = loss_fn(torch.tensor([step]), e, None, z, lab)
                val.backward()
                key = f"{tag}_lam{lam}_s{step}"
                out.update({f"{key}_loss": npy(val), f"{key}_dz": npy(z.grad), f"{key}_demb": npy(e.grad)})
                assert seen[-1].dtype == torch.float64 and tuple(seen[-1].shape) == (2, 12)
                out[f"{tag}_rmi_now"] = npy(0.5 * seen[-1])          # f64 [B, C]; independent of lambda and step
    save("g6_rmi_hiera_triplet_loss", **out)
    # G7: 3-level triplet alone
    trip = RefTriplet3(num_classes=7, upper_ids=[1, 2, 3, 4], lower_ids=[5, 6])
    lab = blocky_labels(g, 2, 64, 64, 7)
    emb = torch.nn.functional.normalize(torch.randn(2, 8, 8, 8, generator=g), dim=1).requires_grad_(True)
    val, cnt = trip(emb, lab)
    val.backward()
    save("g7_rmi_triplet", lab=npy(lab).astype(np.uint8), emb=npy(emb), val=npy(val), cnt=npy(cnt),
         demb=npy(emb.grad))


# ------------------------------------------------------------------ G8 pixel accuracy
def g8():
    g = torch.Generator().manual_seed(8)
    lab = blocky_labels(g, 2, 64, 64, 4)
    pred = torch.where(torch.rand(2, 64, 64, generator=g) < 0.7, lab.clamp(max=3),
                       torch.randint(0, 4, (2, 64, 64), generator=g))
    acc = ref_train.compute_pixel_accuracy(pred, lab)
    acc0 = ref_train.compute_pixel_accuracy(pred, torch.full_like(lab, 255))
    save("g8_pixel_accuracy", lab=npy(lab).astype(np.uint8), pred=npy(pred).astype(np.uint8),
         acc=np.float64(acc), acc_void=np.float64(acc0))


# ------------------------------------------------------------------ G9 PIL's antialiasing bilinear image resize
def g9():
    """``img.resize(size, Image.BILINEAR)`` of ``dataset/dataloader.py:50`` as Pillow itself computes it (third-party arithmetic of
    the reference; Pillow is importable in the authoring container): random RGB images, down-, up- and mixed scaling."""
    from PIL import Image
    import PIL
    rng = np.random.default_rng(9)
    out = {"pillow_version": np.array([int(v) for v in PIL.__version__.split(".")[:3]], dtype=np.int64)}
    for tag, (h, w), (wo, ho) in (("down", (97, 131), (50, 40)), ("up", (24, 20), (75, 60)), ("mixed", (40, 90), (64, 64)),
                                  ("big", (200, 300), (150, 150)), ("same_w", (33, 48), (48, 20))):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a[: h // 4, : w // 3] = 255            # saturated and flat regions too
        a[-h // 5:, :, 1] = 0
        ref = np.asarray(Image.fromarray(a, "RGB").resize((wo, ho), Image.BILINEAR))
        out[f"{tag}_in"] = a
        out[f"{tag}_out"] = ref.copy()
    save("g9_pil_bilinear", **out)


if __name__ == "__main__":
    for fn in (g1, g2, g3, g4, g5, g6, g8, g9):
        fn()
