"""Pin the oracle (CPU restatement) against the golden vectors produced by importing the
reference (tools/make_goldens.py).  Integer/index tensors bit-exact; floats to stated tolerances."""
import numpy as np
import pytest
import torch

from oracle import hierarchy, losses, nets

T = torch.from_numpy


def lab(a):
    return T(a.astype(np.int64))


# ---------------------------------------------------------------- G1
def test_g1_maps_bit_exact(golden):
    g = golden("g1_maps")
    cfgs = {"a": ([[0, 3], [4, 6], [7], [8]], 9), "b": ([[0, 1], [2, 3]], 4), "c": ([[0], [1, 4], [5, 6]], 7)}
    for k, (cfg, nf) in cfgs.items():
        assert np.array_equal(hierarchy.build_fine_to_coarse_map(cfg, nf).numpy(), g[f"{k}_f2c"])
        assert np.array_equal(np.asarray(hierarchy.build_hiera_index(cfg)), g[f"{k}_hidx"])
    assert np.array_equal(hierarchy.build_fine_to_super_map([[0], [1, 6]], 7).numpy(), g["c_f2s"])


def test_maps_raise_on_uncovered_ids():
    with pytest.raises(ValueError):
        hierarchy.build_fine_to_super_map([[0, 1], [2, 3]], 9)   # example-config.yaml:10 shape


# ---------------------------------------------------------------- G2
def test_g2_targets_bit_exact(golden):
    g = golden("g2_targets")
    f2m, f2h = torch.tensor([0, 1, 1, 1, 1, 2, 2]), torch.tensor([0, 1, 1, 1, 1, 1, 1])
    for tag in ("even", "odd"):
        _, c = losses.prepare_targets_two_level(lab(g[f"{tag}_lab9"]), [[0, 4], [4, 7], [7, 8], [8, 9]])
        assert np.array_equal(c.numpy(), g[f"{tag}_coarse"].astype(np.int64))
        _, m, h = losses.prepare_targets_three_level(lab(g[f"{tag}_lab7"]), f2m, f2h)
        assert np.array_equal(m.numpy(), g[f"{tag}_mid"].astype(np.int64))
        assert np.array_equal(h.numpy(), g[f"{tag}_high"].astype(np.int64))
    _, c = losses.prepare_targets_two_level(lab(g["gap_lab"]), [[0, 3], [4, 5]])
    assert np.array_equal(c.numpy(), g["gap_coarse"].astype(np.int64))


# ---------------------------------------------------------------- G3
HEAD_KW = dict(in_channels=64, c1_in_channels=16, c1_channels=8, aspp_channels=16,
               dilations=(1, 12, 24, 36), num_classes=6, proj_dim=8, proj_type="convmlp")


def _head_from_golden(g):
    head = nets.DepthwiseSeparableASPPContrastHead(**HEAD_KW)
    sd = {k[4:]: T(v) for k, v in g.items() if k.startswith("sd__")}
    assert set(sd) == set(head.state_dict())            # state_dict keys identical to the reference
    for k, v in head.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    head.load_state_dict(sd)
    return head, sd


@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_g3_head_fwd_bwd(golden, tag):
    g = golden("g3_head")
    head, sd = _head_from_golden(g)
    head.train()
    c1 = T(g[f"{tag}_c1"]).requires_grad_(True)
    c4 = T(g[f"{tag}_c4"]).requires_grad_(True)
    logits, emb = head([c1, None, None, c4])
    np.testing.assert_allclose(logits.detach().numpy(), g[f"{tag}_logits"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(emb.detach().numpy(), g[f"{tag}_emb"], rtol=1e-5, atol=1e-6)
    ((logits * T(g[f"{tag}_gl"])).sum() + (emb * T(g[f"{tag}_ge"])).sum()).backward()
    np.testing.assert_allclose(c1.grad.numpy(), g[f"{tag}_dc1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c4.grad.numpy(), g[f"{tag}_dc4"], rtol=1e-4, atol=1e-5)
    for k, p in head.named_parameters():
        ref = g[f"{tag}_grad__{k}"]
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    for k, v in head.state_dict().items():
        if "running" in k or k == "step" or "num_batches" in k:
            np.testing.assert_allclose(v.numpy(), g[f"{tag}_after__{k}"], rtol=1e-5, atol=1e-6, err_msg=k)
    head.eval()
    with torch.no_grad():
        le, ee = head([c1, None, None, c4])
    np.testing.assert_allclose(le.numpy(), g[f"{tag}_logits_eval"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ee.numpy(), g[f"{tag}_emb_eval"], rtol=1e-5, atol=1e-6)


def test_head_default_init_matches_reference_seed_for_seed(golden):
    """Same RNG draw order as the reference constructor (incl. the discarded dense ASPP convs)."""
    g = golden("g3_head")
    torch.manual_seed(3)
    head = nets.DepthwiseSeparableASPPContrastHead(**HEAD_KW)
    for k, v in head.state_dict().items():
        if k.endswith("weight") and v.dim() == 4 or k == "cls_seg.bias":
            np.testing.assert_array_equal(v.numpy(), g["sd__" + k], err_msg=k)


def test_head_rejects_unknown_proj_type():
    with pytest.raises(ValueError):
        nets.DepthwiseSeparableASPPContrastHead(**{**HEAD_KW, "proj_type": "mlp"})


# ---------------------------------------------------------------- backbone (parity unpinned)
@pytest.mark.parametrize("depth,params,chans", [(18, 11176512, (64, 128, 256, 512)),
                                                (50, 23508032, (256, 512, 1024, 2048)),
                                                (101, 42500160, (256, 512, 1024, 2048))])
def test_backbone_param_counts_and_shapes(depth, params, chans):
    bb = nets.ResNetBackbone(depth, pretrained=False)
    assert sum(p.numel() for p in bb.parameters()) == params
    bb.eval()
    with torch.no_grad():
        outs = bb(torch.randn(1, 3, 64, 64))
    assert [o.shape[1] for o in outs] == list(chans)
    assert [o.shape[2] for o in outs] == [16, 8, 4, 2]
    keys = set(bb.state_dict())
    assert {"stem_conv.weight", "stem_bn.running_var", "layer1.0.conv1.weight", "layer4.0.downsample.1.weight"} <= keys


def test_backbone_rejects_bad_depth():
    with pytest.raises(ValueError):
        nets.ResNetBackbone(77)


# ---------------------------------------------------------------- G4
HIDX2, HMAP2 = [[0, 2], [2, 4]], [0, 0, 1, 1]


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_g4_two_level_bce_and_ce(golden, tag):
    g = golden("g4_two_level_parts")
    label = lab(g[f"{tag}_lab"])
    z = T(g[f"{tag}_z"]).requires_grad_(True)
    tf, tc = losses.prepare_targets_two_level(label, HIDX2)
    lh = losses.losses_hiera_two_level(z, tf, tc, 4, HIDX2)
    lh.backward()
    np.testing.assert_allclose(float(lh), float(g[f"{tag}_hiera"]), rtol=2e-6)
    np.testing.assert_allclose(z.grad.numpy(), g[f"{tag}_dz_hiera"], rtol=1e-4, atol=1e-9)
    z2 = T(g[f"{tag}_z"]).requires_grad_(True)
    ce = losses.CrossEntropyLoss()
    lf, lc = ce(z2[:, :4], tf), ce(z2[:, 4:6], tc)
    (lf + lc).backward()
    np.testing.assert_allclose(float(lf), float(g[f"{tag}_ce_f"]), rtol=2e-6)
    np.testing.assert_allclose(float(lc), float(g[f"{tag}_ce_c"]), rtol=2e-6)
    np.testing.assert_allclose(z2.grad.numpy(), g[f"{tag}_dz_ce"], rtol=1e-5, atol=1e-10)


@pytest.mark.parametrize("pre,nc,hmap,hidx", [("trip", 4, HMAP2, HIDX2), ("trips", 3, [0, 0, 1], [[0, 2], [2, 3]]),
                                              ("tripl", 4, HMAP2, HIDX2)])
def test_g4_triplet(golden, pre, nc, hmap, hidx):
    g = golden("g4_two_level_parts")
    trip = losses.TreeTripletLoss(nc, hmap, hidx)
    emb = T(g[f"{pre}_emb"]).requires_grad_(True)
    val, cnt = trip(emb, lab(g[f"{pre}_lab"]))
    assert np.array_equal(cnt.numpy(), g[f"{pre}_cnt"])           # class_count bit-exact
    val.backward()
    np.testing.assert_allclose(float(val), float(g[f"{pre}_val"]), rtol=2e-6)
    np.testing.assert_allclose(emb.grad.numpy(), g[f"{pre}_demb"], rtol=1e-5, atol=1e-8)


def test_g4_triplet_all_void(golden):
    g = golden("g4_two_level_parts")
    trip = losses.TreeTripletLoss(4, HMAP2, HIDX2)
    val, cnt = trip(T(g["trip_emb"]), torch.full((2, 64, 64), 255, dtype=torch.long))
    assert val is None and np.array_equal(cnt.numpy(), g["trip_void_cnt"])


# ---------------------------------------------------------------- G5
@pytest.mark.parametrize("tag", ["even", "odd"])
@pytest.mark.parametrize("step", [0, 40000, 80000])
def test_g5_hiera_triplet_loss(golden, tag, step):
    g = golden("g5_hiera_triplet_loss")
    fn = losses.HieraTripletLoss(4, HMAP2, HIDX2)
    z = T(g[f"{tag}_z"]).requires_grad_(True)
    e = T(g[f"{tag}_emb"]).requires_grad_(True)
    val = fn(torch.tensor([step]), e, None, z, lab(g[f"{tag}_lab"]))
    val.backward()
    np.testing.assert_allclose(float(val), float(g[f"{tag}_s{step}_loss"]), rtol=2e-6)
    np.testing.assert_allclose(z.grad.numpy(), g[f"{tag}_s{step}_dz"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(e.grad.numpy(), g[f"{tag}_s{step}_demb"], rtol=1e-5, atol=1e-8)


# ---------------------------------------------------------------- G6 / G7
@pytest.mark.parametrize("tag", ["even", "odd"])
@pytest.mark.parametrize("lam", [0.0, 0.5])
@pytest.mark.parametrize("step", [0, 30000])
def test_g6_rmi_hiera_triplet_loss(golden, tag, lam, step):
    g = golden("g6_rmi_hiera_triplet_loss")
    f2m, f2h = torch.tensor([0, 1, 1, 1, 1, 2, 2]), torch.tensor([0, 1, 1, 1, 1, 1, 1])
    fn = losses.RMIHieraTripletLoss(7, 3, 2, f2m, f2h, loss_weight_lambda=lam)
    z = T(g[f"{tag}_z"]).requires_grad_(True)
    e = T(g[f"{tag}_emb"]).requires_grad_(True)
    val = fn(torch.tensor([step]), e, None, z, lab(g[f"{tag}_lab"]))
    val.backward()
    key = f"{tag}_lam{lam}_s{step}"
    np.testing.assert_allclose(float(val), float(g[f"{key}_loss"]), rtol=5e-6)
    np.testing.assert_allclose(z.grad.numpy(), g[f"{key}_dz"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(e.grad.numpy(), g[f"{key}_demb"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_g6_rmi_now_per_channel_f64(golden, tag):
    """The oracle's f64 per-(image, channel) rmi_now against the reference's own values (rmi_hiera_triplet_loss.py:513)."""
    g = golden("g6_rmi_hiera_triplet_loss")
    f2m, f2h = torch.tensor([0, 1, 1, 1, 1, 2, 2]), torch.tensor([0, 1, 1, 1, 1, 1, 1])
    label = lab(g[f"{tag}_lab"])
    tf, tm, th = losses.prepare_targets_three_level(label, f2m, f2h)
    probs = torch.sigmoid(T(g[f"{tag}_z"]))
    _, (oh_f, oh_m, oh_h) = losses.losses_hiera_three_level(probs, tf, tm, th, 7, 3, 2, f2m, f2h)
    valid = torch.cat([(t != 255).unsqueeze(1).float().expand(-1, n, -1, -1) for t, n in ((tf, 7), (tm, 3), (th, 2))], 1)
    _, rmi = losses.rmi_lower_bound(torch.cat([oh_f, oh_m, oh_h], 1), probs * valid + 1e-6, 3)
    assert rmi.dtype == torch.float64
    np.testing.assert_allclose(rmi.numpy(), g[f"{tag}_rmi_now"], rtol=1e-9, atol=0)


def test_g7_rmi_triplet(golden):
    g = golden("g7_rmi_triplet")
    trip = losses.RMITreeTripletLoss(7, [1, 2, 3, 4], [5, 6])
    emb = T(g["emb"]).requires_grad_(True)
    val, cnt = trip(emb, lab(g["lab"]))
    assert np.array_equal(cnt.numpy(), g["cnt"])
    val.backward()
    np.testing.assert_allclose(float(val), float(g["val"]), rtol=2e-6)
    np.testing.assert_allclose(emb.grad.numpy(), g["demb"], rtol=1e-5, atol=1e-8)


def test_rmi_triplet_raises_outside_groups():
    trip = losses.RMITreeTripletLoss(9, [1, 2, 3, 4], [5, 6])
    with pytest.raises(ValueError):
        trip(torch.randn(1, 4, 2, 2), torch.full((1, 4, 4), 7, dtype=torch.long))


# ---------------------------------------------------------------- G8
def test_g8_pixel_accuracy(golden):
    g = golden("g8_pixel_accuracy")
    pred, label = lab(g["pred"]), lab(g["lab"])
    logits = torch.nn.functional.one_hot(pred, 4).permute(0, 3, 1, 2).float()
    c, v = losses.pixel_accuracy_counts(logits, label)
    assert c / max(v, 1) == float(g["acc"])
    c0, v0 = losses.pixel_accuracy_counts(logits, torch.full_like(label, 255))
    assert (c0, v0) == (0, 0) and float(g["acc_void"]) == 0.0


# ---------------------------------------------------------------- G9: PIL's antialiasing bilinear image resize
@pytest.mark.parametrize("tag", ["down", "up", "mixed", "big", "same_w"])
def test_g9_pil_bilinear_resize_oracle_and_host_tables(golden, tag):
    """The numpy restatement of Pillow's resampler is bit-exact against Pillow's own output, and the product's HOST coefficient
    tables (sh_resize_bilinear_coeffs: double arithmetic in the C library, no GPU involved) equal the restatement's."""
    import ctypes
    from oracle import resize
    from seghiero_amd._lib import LIB
    g = golden("g9_pil_bilinear")
    a, ref = g[f"{tag}_in"], g[f"{tag}_out"]
    ho, wo = ref.shape[:2]
    assert np.array_equal(resize.pil_bilinear_resize_u8(a, (wo, ho)), ref)
    fn = LIB.raw("sh_resize_bilinear_coeffs")
    for n_in, n_out in ((a.shape[1], wo), (a.shape[0], ho)):
        bounds, kk = resize.coeffs(n_in, n_out)
        ks = fn(n_in, n_out, None, None, 0)
        assert ks == kk.shape[1]
        cb, ck = (ctypes.c_int * (2 * n_out))(), (ctypes.c_int * (ks * n_out))()
        assert fn(n_in, n_out, cb, ck, ks * n_out) == ks
        assert np.array_equal(np.array(list(cb), np.int32).reshape(n_out, 2), bounds)
        assert np.array_equal(np.array(list(ck), np.int32).reshape(n_out, ks), kk)
