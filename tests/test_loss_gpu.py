"""GPU parity of the fused loss kernels: against the golden vectors generated from the reference itself
(tests/golden, tools/make_goldens.py) and against the oracle on other shapes.  Index tensors bit-exact; loss values
rtol 1e-5 (the north-star tolerance is 1e-4 absolute on the loss); gradients rtol 1e-4."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HIDX2, HMAP2 = [[0, 2], [2, 4]], [0, 0, 1, 1]


@pytest.fixture(scope="module")
def sa():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import seghiero_amd
    from seghiero_amd import loss, ops
    return seghiero_amd, loss, ops


def T(a):
    return torch.from_numpy(a)


def lab(a):
    return T(a.astype(np.int64))


def close(a, b, rtol, atol, msg=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_coarse_targets_bit_exact_g2(sa, tag):
    _, loss, ops = sa
    g = load_golden("g2_targets")
    hidx = [[0, 4], [4, 7], [7, 8], [8, 9]]
    l8 = T(g[f"{tag}_lab9"]).to(DEV)
    n, h, w = l8.shape
    z = ops.new_act(n, 13, h, w, DEV, zero=True)
    _, _, coarse, _ = ops.hiera2_fwd(z, l8, 9, hidx, want_coarse=True)
    assert np.array_equal(coarse.cpu().numpy(), g[f"{tag}_coarse"])
    lg = T(g["gap_lab"]).to(DEV)
    zg = ops.new_act(1, 7, 16, 16, DEV, zero=True)
    _, _, cg, _ = ops.hiera2_fwd(zg, lg, 5, [[0, 3], [4, 5]], want_coarse=True)
    assert np.array_equal(cg.cpu().numpy(), g["gap_coarse"])


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_three_level_targets_bit_exact_g2(sa, tag):
    """a14: the mid / high target maps of `_prepare_targets_three_level` (rmi_hiera_triplet_loss.py:21-63) as the HIP
    kernel derives them, bit-exact against the reference's own output (G2)."""
    _, loss, ops = sa
    g = load_golden("g2_targets")
    f2m, f2h = [0, 1, 1, 1, 1, 2, 2], [0, 1, 1, 1, 1, 1, 1]
    l8 = T(g[f"{tag}_lab7"]).to(DEV)
    n, h, w = l8.shape
    z = ops.new_act(n, 12, h, w, DEV, zero=True)
    _, _, _, (mid, high) = ops.hiera3_fwd(z, l8, 7, 3, 2, f2m, f2h, want_probs=False, want_targets=True)
    assert mid.dtype == torch.uint8 and high.dtype == torch.uint8
    assert np.array_equal(mid.cpu().numpy(), g[f"{tag}_mid"])
    assert np.array_equal(high.cpu().numpy(), g[f"{tag}_high"])
    # the module-level helper (same name as the reference's) returns int64 maps like the reference does
    tf, tm, th = loss.prepare_targets_three_level(lab(g[f"{tag}_lab7"]).to(DEV), torch.tensor(f2m), torch.tensor(f2h))
    assert tm.dtype == torch.int64 and np.array_equal(tm.cpu().numpy(), g[f"{tag}_mid"].astype(np.int64))
    assert np.array_equal(th.cpu().numpy(), g[f"{tag}_high"].astype(np.int64))
    assert np.array_equal(tf.cpu().numpy(), g[f"{tag}_lab7"].astype(np.int64))


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_hiera_bce_and_ce_g4(sa, tag):
    _, loss, ops = sa
    g = load_golden("g4_two_level_parts")
    z = T(g[f"{tag}_z"]).to(DEV).requires_grad_(True)
    l8 = T(g[f"{tag}_lab"]).to(DEV)
    # the fused kernel returns hiera + ce_f + ce_c; check the parts through the sums it leaves behind
    zn = ops.to_nhwc(z.detach())
    total, sums, _, _ = ops.hiera2_fwd(zn, l8, 4, HIDX2)
    s = sums.cpu().numpy()
    hiera = 5.0 * (s[0] / (max(s[4], 1) * 4) + s[1] / (max(s[5], 1) * 2))
    close(hiera, g[f"{tag}_hiera"], 1e-5, 0)
    close(s[2] / s[6], g[f"{tag}_ce_f"], 1e-5, 0)
    close(s[3] / s[6], g[f"{tag}_ce_c"], 1e-5, 0)
    close(total[0], float(g[f"{tag}_hiera"]) + float(g[f"{tag}_ce_f"]) + float(g[f"{tag}_ce_c"]), 1e-5, 0)
    d = ops.hiera2_bwd(zn, l8, 4, HIDX2, sums, None, 1.0)
    close(d, g[f"{tag}_dz_hiera"] + g[f"{tag}_dz_ce"], 1e-4, 1e-9)
    # the stand-alone CE wrapper
    ce = loss.CrossEntropyLoss()
    lf = ce(z[:, :4], lab(g[f"{tag}_lab"]).to(DEV))
    close(lf, g[f"{tag}_ce_f"], 1e-5, 0)


@pytest.mark.parametrize("pre,nc,hmap,hidx", [("trip", 4, HMAP2, HIDX2), ("trips", 3, [0, 0, 1], [[0, 2], [2, 3]]),
                                              ("tripl", 4, HMAP2, HIDX2)])
def test_triplet_g4(sa, pre, nc, hmap, hidx):
    _, loss, ops = sa
    g = load_golden("g4_two_level_parts")
    trip = loss.TreeTripletLoss(nc, hmap, hidx)
    emb = T(g[f"{pre}_emb"]).to(DEV).requires_grad_(True)
    val, cnt = trip(emb, lab(g[f"{pre}_lab"]).to(DEV))
    assert np.array_equal(cnt.cpu().numpy(), g[f"{pre}_cnt"])          # class_count bit-exact
    val.backward()
    close(val, g[f"{pre}_val"], 1e-5, 0)
    close(emb.grad, g[f"{pre}_demb"], 1e-4, 1e-8)


def test_triplet_all_void(sa):
    _, loss, ops = sa
    g = load_golden("g4_two_level_parts")
    trip = loss.TreeTripletLoss(4, HMAP2, HIDX2)
    val, cnt = trip(T(g["trip_emb"]).to(DEV), torch.full((2, 64, 64), 255, dtype=torch.long, device=DEV))
    assert val is None and int(cnt) == 0 == int(g["trip_void_cnt"][0])


@pytest.mark.parametrize("tag", ["even", "odd"])
@pytest.mark.parametrize("step", [0, 40000, 80000])
def test_hiera_triplet_loss_g5(sa, tag, step):
    _, loss, ops = sa
    g = load_golden("g5_hiera_triplet_loss")
    fn = loss.HieraTripletLoss(4, HMAP2, HIDX2).to(DEV)
    z = T(g[f"{tag}_z"]).to(DEV).requires_grad_(True)
    e = T(g[f"{tag}_emb"]).to(DEV).requires_grad_(True)
    val = fn(torch.tensor([step]), e, None, z, lab(g[f"{tag}_lab"]).to(DEV))
    val.backward()
    close(val, g[f"{tag}_s{step}_loss"], 1e-5, 0)
    close(z.grad, g[f"{tag}_s{step}_dz"], 1e-4, 1e-9)
    close(e.grad, g[f"{tag}_s{step}_demb"], 1e-4, 1e-8)


def _blocky(g, b, h, w, nf, cell=16):
    small = torch.randint(0, nf, (b, -(-h // cell), -(-w // cell)), generator=g)
    l = small.repeat_interleave(cell, 1).repeat_interleave(cell, 2)[:, :h, :w].clone()
    l[torch.rand(b, h, w, generator=g) < 0.05] = 255
    l[:, :8] = 255
    return l


@pytest.mark.parametrize("lo,hi", [((32, 32), (128, 128)), ((19, 13), (75, 51)), ((8, 8), (128, 128))])
def test_fused_resize_loss_matches_oracle(sa, lo, hi):
    """Low-resolution logits + fused bilinear resize == oracle(F.interpolate(...)) incl. the gradient w.r.t. the
    low-resolution logits (the path train_step uses; reference train.py:282-306)."""
    _, loss, ops = sa
    from oracle import losses as ol
    g = torch.Generator().manual_seed(lo[0] * 7 + hi[1])
    hidx, hmap = [[0, 4], [4, 7], [7, 8], [8, 9]], [0, 0, 0, 0, 1, 1, 1, 2, 3]
    z = (1.5 * torch.randn(2, 13, *lo, generator=g))
    e = F.normalize(torch.randn(2, 16, 4, 4, generator=g), dim=1)
    label = _blocky(g, 2, *hi, 9)
    zr, er = z.clone().requires_grad_(True), e.clone().requires_grad_(True)
    ref = ol.HieraTripletLoss(9, hmap, hidx)(torch.tensor([30000]), er, None,
                                              F.interpolate(zr, size=hi, mode="bilinear", align_corners=False), label)
    ref.backward()
    zg, eg = z.to(DEV).requires_grad_(True), e.to(DEV).requires_grad_(True)
    val = loss.HieraTripletLoss(9, hmap, hidx).to(DEV)(30000, eg, None, zg, label.to(DEV))
    val.backward()
    close(val, ref, 1e-5, 0)
    close(zg.grad, zr.grad, 2e-4, 1e-8)
    close(eg.grad, er.grad, 1e-4, 1e-8)


@pytest.mark.parametrize("lo,hi", [((16, 16), (64, 64)), ((5, 4), (75, 51)), ((32, 32), (128, 128))])
def test_hiera2_forward_emits_gradient_for_gather_only_backward(sa, lo, hi):
    """sh_hiera2_loss_fwd with grad_out: same loss / sums bits as the plain forward, and the gather-only backward equals the
    two-pass backward (full-resolution gradient recomputed, then the same gather) bit for bit at unit upstream gradient; with an
    upstream gradient != 1 the two differ by one rounding (scale applied after instead of before the sum)."""
    _, loss, ops = sa
    g = torch.Generator().manual_seed(lo[0] + hi[1])
    hidx = [[0, 4], [4, 7], [7, 8], [8, 9]]
    zn = ops.to_nhwc((1.5 * torch.randn(2, 13, *lo, generator=g)).to(DEV), cpad=16)
    l8 = ops.labels_u8(_blocky(g, 2, *hi, 9).to(DEV))
    val, sums, _, none = ops.hiera2_fwd(zn, l8, 9, hidx)
    val2, sums2, _, gw = ops.hiera2_fwd(zn, l8, 9, hidx, want_grad=True)
    assert none is None and gw is not None
    assert torch.equal(val, val2) and torch.equal(sums, sums2)
    d = ops.hiera2_bwd(zn, l8, 9, hidx, sums, None, 1.0)
    assert torch.equal(ops.hiera2_bwd(zn, l8, 9, hidx, sums2, None, 1.0, grad_ws=gw), d)
    gs = torch.tensor([0.37], device=DEV)
    want = ops.hiera2_bwd(zn, l8, 9, hidx, sums, gs, 2.0)
    close(ops.hiera2_bwd(zn, l8, 9, hidx, sums2, gs, 2.0, grad_ws=gw), want, 1e-6, 1e-6 * float(want.abs().max()))


@pytest.mark.parametrize("lo,hi", [((8, 8), (128, 128)), ((5, 4), (75, 51)), ((16, 16), (16, 16))])
def test_aux_ce_fused_resize(sa, lo, hi):
    _, loss, ops = sa
    g = torch.Generator().manual_seed(hi[0])
    z = torch.randn(2, 9, *lo, generator=g)
    label = _blocky(g, 2, *hi, 9)
    zr = z.clone().requires_grad_(True)
    ref = F.cross_entropy(F.interpolate(zr, size=hi, mode="bilinear", align_corners=False), label, ignore_index=255)
    ref.backward()
    zn = ops.to_nhwc(z.to(DEV), cpad=12)
    l8 = ops.labels_u8(label.to(DEV))
    val, sums, _ = ops.ce_fwd(zn, l8)
    close(val[0], ref, 1e-5, 0)
    d = ops.ce_bwd(zn, l8, sums, None, 1.0)
    close(d, zr.grad, 2e-4, 1e-9)
    # forward that also leaves the per-pixel gradient: same loss bits; the gather-only backward equals the two-pass one bit for bit at
    # unit upstream gradient and to rounding (one extra multiply) otherwise
    val2, sums2, gw = ops.ce_fwd(zn, l8, want_grad=True)
    assert torch.equal(val2, val) and torch.equal(sums2, sums)
    if lo != hi:
        assert gw is not None
        assert torch.equal(ops.ce_bwd(zn, l8, sums2, None, 1.0, grad_ws=gw), d)
        gs = torch.tensor([0.4], device=DEV)
        close(ops.ce_bwd(zn, l8, sums2, gs, 1.0, grad_ws=gw), 0.4 * zr.grad, 2e-4, 1e-9)
    else:
        assert gw is None


def test_pixel_metrics(sa):
    _, loss, ops = sa
    from oracle import losses as ol
    g = torch.Generator().manual_seed(11)
    z = torch.randn(2, 13, 32, 32, generator=g)
    label = _blocky(g, 2, 128, 128, 9)
    full = F.interpolate(z, size=(128, 128), mode="bilinear", align_corners=False)
    c, v = ol.pixel_accuracy_counts(full[:, :9], label)
    cm = ol.confusion_matrix(full[:, :9], label, 9)
    counts = ops.pixel_metrics(ops.to_nhwc(z.to(DEV)), ops.labels_u8(label.to(DEV)), 9).cpu()
    assert int(counts[1]) == v
    # argmax over interpolated logits can flip on exact ties only; counts must agree exactly on this seeded input
    assert int(counts[0]) == c
    assert torch.equal(counts[2:].reshape(9, 9), cm)


# ---------------------------------------------------------------- 3-level RMI loss vs the reference goldens (G6, G7)
F2M, F2H = [0, 1, 1, 1, 1, 2, 2], [0, 1, 1, 1, 1, 1, 1]


@pytest.mark.parametrize("tag", ["even", "odd"])
@pytest.mark.parametrize("lam", [0.0, 0.5])
@pytest.mark.parametrize("step", [0, 30000])
def test_rmi_hiera_triplet_loss_g6(sa, tag, lam, step):
    """Loss rtol 1e-5; gradient w.r.t. the logits within 1e-5 of the tensor's max (measured: 2.4e-7, tests/diag/rmi_dz_err.py -- the
    reference differentiates through log(diag(chol)+1e-8) while the kernel uses the closed form; both in f64)."""
    _, loss, ops = sa
    g = load_golden("g6_rmi_hiera_triplet_loss")
    fn = loss.RMIHieraTripletLoss(7, 3, 2, torch.tensor(F2M), torch.tensor(F2H), loss_weight_lambda=lam).to(DEV)
    z = T(g[f"{tag}_z"]).to(DEV).requires_grad_(True)
    e = T(g[f"{tag}_emb"]).to(DEV).requires_grad_(True)
    val = fn(torch.tensor([step]), e, None, z, lab(g[f"{tag}_lab"]).to(DEV))
    val.backward()
    key = f"{tag}_lam{lam}_s{step}"
    close(val, g[f"{key}_loss"], 1e-5, 0)
    ref = g[f"{key}_dz"]
    np.testing.assert_allclose(z.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * float(np.abs(ref).max()))
    close(e.grad, g[f"{key}_demb"], 1e-4, 1e-8)


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_rmi_per_channel_values_g6(sa, tag):
    """The f64 core of the RMI term (Gram -> inverse -> Schur complement -> Cholesky log-det, :498-513): per-(image, channel)
    rmi_now read back from the kernels' workspace against the reference's own f64 values (G6 `*_rmi_now`)."""
    _, loss, ops = sa
    g = load_golden("g6_rmi_hiera_triplet_loss")
    z = ops.to_nhwc(T(g[f"{tag}_z"]).to(DEV))
    l8 = T(g[f"{tag}_lab"]).to(DEV)
    n, c, H, W = z.shape
    _, _, probs = ops.hiera3_fwd(z, l8, 7, 3, 2, F2M, F2H, want_probs=True)
    ops.rmi_loss(probs, l8, 7, 3, 2, F2M, F2H, want_grad=False)
    got = ops.rmi_values(n, c, H, W, z.device).cpu().numpy()
    ref = g[f"{tag}_rmi_now"]
    assert got.dtype == np.float64 and ref.dtype == np.float64
    # probs are f32 sigmoid values (1 ulp apart between libm implementations) feeding near-singular f64 Gram matrices
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=0)


def test_rmi_triplet_g7(sa):
    _, loss, ops = sa
    g = load_golden("g7_rmi_triplet")
    trip = loss.RMITreeTripletLoss(7, [1, 2, 3, 4], [5, 6])
    emb = T(g["emb"]).to(DEV).requires_grad_(True)
    val, cnt = trip(emb, lab(g["lab"]).to(DEV))
    assert np.array_equal(cnt.cpu().numpy(), g["cnt"])
    val.backward()
    close(val, g["val"], 1e-5, 0)
    close(emb.grad, g["demb"], 1e-4, 1e-8)


def test_rmi_triplet_strict_mirrors_reference_value_error(sa):
    """rmi_tree_triplet_loss.py:39: a label on the embedding grid that is in neither hard-coded group makes ``list.remove`` raise
    ValueError.  strict=True reproduces that (one host sync); the default skips such a class as an anchor without synchronising."""
    _, loss, ops = sa
    g = load_golden("g7_rmi_triplet")
    emb = T(g["emb"]).to(DEV)
    labels = lab(g["lab"]).to(DEV).clone()
    strict = loss.RMITreeTripletLoss(7, [1, 2, 3, 4], [5, 6], strict=True)
    val, cnt = strict(emb, labels)                                    # every label in a group: same result as the default
    assert np.array_equal(cnt.cpu().numpy(), g["cnt"])
    close(val, g["val"], 1e-5, 0)
    H, W = labels.shape[-2:]
    labels[0, (3 * H) // emb.shape[2], (2 * W) // emb.shape[3]] = 8   # a pixel the nearest-neighbour resize picks (row 3, column 2)
    with pytest.raises(ValueError, match="not in list"):
        strict(emb, labels)
    val2, cnt2 = loss.RMITreeTripletLoss(7, [1, 2, 3, 4], [5, 6])(emb, labels)      # default: class 8 is never an anchor
    assert int(cnt2) <= int(g["cnt"][0]) and (val2 is None or torch.isfinite(val2))
    labels[0, (3 * H) // emb.shape[2], (2 * W) // emb.shape[3]] = 2
    labels[0, 1, 1] = 8 if (H // emb.shape[2]) > 1 else 2           # a pixel the resize never reads: no error
    strict(emb, labels)


def test_rmi_loss_fused_resize_matches_oracle(sa):
    """Low-resolution logits + fused x4 resize (the train-step path) vs oracle(F.interpolate(...))."""
    _, loss, ops = sa
    from oracle import losses as ol
    g = torch.Generator().manual_seed(77)
    z = 1.5 * torch.randn(2, 12, 16, 16, generator=g)
    e = F.normalize(torch.randn(2, 16, 4, 4, generator=g), dim=1)
    label = _blocky(g, 2, 64, 64, 7, cell=8)
    zr, er = z.clone().requires_grad_(True), e.clone().requires_grad_(True)
    ref = ol.RMIHieraTripletLoss(7, 3, 2, torch.tensor(F2M), torch.tensor(F2H))(
        torch.tensor([20000]), er, None, F.interpolate(zr, size=(64, 64), mode="bilinear", align_corners=False), label)
    ref.backward()
    zg, eg = z.to(DEV).requires_grad_(True), e.to(DEV).requires_grad_(True)
    val = loss.RMIHieraTripletLoss(7, 3, 2, torch.tensor(F2M), torch.tensor(F2H)).to(DEV)(20000, eg, None, zg, label.to(DEV))
    val.backward()
    close(val, ref, 1e-5, 0)
    r = zr.grad.numpy()
    np.testing.assert_allclose(zg.grad.cpu().numpy(), r, rtol=1e-3, atol=1e-4 * float(np.abs(r).max()))
    close(eg.grad, er.grad, 1e-4, 1e-8)


@pytest.mark.parametrize("lam", [0.0, 0.5])
def test_three_level_backward_two_pass_equals_tile_kernel(sa, lam, monkeypatch):
    """sh_hiera3_loss_bwd with a workspace (full-resolution gradient once, then the resize-adjoint gather) against the LDS-tile
    kernel on the same inputs: same arithmetic, same summation order -- bit-identical."""
    _, loss, ops = sa
    g = torch.Generator().manual_seed(31)
    z = 1.5 * torch.randn(2, 12, 24, 20, generator=g)
    e = F.normalize(torch.randn(2, 16, 6, 5, generator=g), dim=1)
    label = _blocky(g, 2, 96, 80, 7, cell=8)
    grads = []
    monkeypatch.setattr(ops, "LOSS_FWD_GRAD", False)           # (the forward-emits-gradient form has its own test below)
    for two_pass in (True, False):
        monkeypatch.setattr(ops, "LOSS_BWD_TWO_PASS", two_pass)
        zg = z.to(DEV).requires_grad_(True)
        fn = loss.RMIHieraTripletLoss(7, 3, 2, torch.tensor(F2M), torch.tensor(F2H), loss_weight_lambda=lam).to(DEV)
        fn(20000, e.to(DEV), None, zg, label.to(DEV)).backward()
        grads.append(zg.grad.clone())
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("lam", [0.0, 0.5])
def test_three_level_forward_emits_gradient_equals_two_pass(sa, lam, monkeypatch):
    """r3: sh_hiera3_loss_fwd leaves the per-pixel gradient of the BCE / CE terms (unit upstream gradient), the backward adds the RMI term
    in one streaming pass (sigmoid recovered from the stored probabilities) and gathers with the scale -- against the two-pass backward
    that recomputes every pixel: same terms, the scale applied after instead of inside the sums (1e-5 of the largest entry), same loss."""
    _, loss, ops = sa
    g = torch.Generator().manual_seed(37)
    z = 1.5 * torch.randn(2, 12, 24, 20, generator=g)
    e = F.normalize(torch.randn(2, 16, 6, 5, generator=g), dim=1)
    label = _blocky(g, 2, 96, 80, 7, cell=8)
    label[0, :16, :24] = 255
    grads, vals = [], []
    for fwd_grad in (True, False):
        monkeypatch.setattr(ops, "LOSS_FWD_GRAD", fwd_grad)
        zg = z.to(DEV).requires_grad_(True)
        fn = loss.RMIHieraTripletLoss(7, 3, 2, torch.tensor(F2M), torch.tensor(F2H), loss_weight_lambda=lam).to(DEV)
        v = fn(20000, e.to(DEV), None, zg, label.to(DEV))
        (3.0 * v).backward()                                   # a non-unit upstream gradient: the scale is applied by the gather
        grads.append(zg.grad.clone()); vals.append(float(v))
    assert vals[0] == vals[1]
    a, b = grads[0].cpu().double(), grads[1].cpu().double()
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())


def test_rmi_loss_wide_image_matches_oracle(sa):
    """Several 64-column strips and 64-row chunks of Gram partials, and dprob blocks away from / at every border:
    full-resolution logits (identity resize) at 140 x 200 against the oracle."""
    _, loss, ops = sa
    from oracle import losses as ol
    g = torch.Generator().manual_seed(5)
    z = 1.5 * torch.randn(1, 12, 140, 200, generator=g)
    e = F.normalize(torch.randn(1, 16, 5, 7, generator=g), dim=1)
    label = _blocky(g, 1, 140, 200, 7, cell=10)
    zr, er = z.clone().requires_grad_(True), e.clone().requires_grad_(True)
    ref = ol.RMIHieraTripletLoss(7, 3, 2, torch.tensor(F2M), torch.tensor(F2H))(torch.tensor([20000]), er, None, zr, label)
    ref.backward()
    zg, eg = z.to(DEV).requires_grad_(True), e.to(DEV).requires_grad_(True)
    val = loss.RMIHieraTripletLoss(7, 3, 2, torch.tensor(F2M), torch.tensor(F2H)).to(DEV)(20000, eg, None, zg, label.to(DEV))
    val.backward()
    close(val, ref, 1e-5, 0)
    r = zr.grad.numpy()
    np.testing.assert_allclose(zg.grad.cpu().numpy(), r, rtol=1e-3, atol=1e-4 * float(np.abs(r).max()))
