"""GPU parity of the drop-in modules and of the whole training step.

* head: against golden G3 = outputs of the REFERENCE head itself (tests/golden/g3_head.npz);
* backbone / full step: against the oracle (torchvision trunk is parity-unpinned, see oracle/__init__.py).
Tolerances are stated inline; the north-star bar is 1e-4 on the loss."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HEAD_KW = dict(in_channels=64, c1_in_channels=16, c1_channels=8, aspp_channels=16,
               dilations=(1, 12, 24, 36), num_classes=6, proj_dim=8, proj_type="convmlp")


@pytest.fixture(scope="module")
def sa():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import seghiero_amd
    return seghiero_amd


def T(a):
    return torch.from_numpy(a)


def close(a, b, rtol, atol, msg=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_head_matches_reference_golden(sa, tag):
    from seghiero_amd.head import DepthwiseSeparableASPPContrastHead
    g = load_golden("g3_head")
    head = DepthwiseSeparableASPPContrastHead(**HEAD_KW)
    sd = {k[4:]: T(v) for k, v in g.items() if k.startswith("sd__")}
    assert set(sd) == set(head.state_dict())
    head.load_state_dict(sd)
    head.to(DEV).train()
    c1 = T(g[f"{tag}_c1"]).to(DEV).requires_grad_(True)
    c4 = T(g[f"{tag}_c4"]).to(DEV).requires_grad_(True)
    logits, emb = head([c1, None, None, c4])
    assert tuple(logits.shape) == g[f"{tag}_logits"].shape and tuple(emb.shape) == g[f"{tag}_emb"].shape
    close(logits, g[f"{tag}_logits"], 1e-4, 1e-4, "logits")
    close(emb, g[f"{tag}_emb"], 1e-4, 1e-5, "embedding")
    ((logits * T(g[f"{tag}_gl"]).to(DEV)).sum() + (emb * T(g[f"{tag}_ge"]).to(DEV)).sum()).backward()
    assert relerr(c1.grad, T(g[f"{tag}_dc1"])) < 1e-3
    assert relerr(c4.grad, T(g[f"{tag}_dc4"])) < 1e-3
    for k, p in head.named_parameters():
        ref = T(g[f"{tag}_grad__{k}"])
        assert p.grad is not None, k
        err = float((p.grad.cpu().double() - ref.double()).abs().max()) / max(float(ref.abs().max()), 1e-3)
        assert err < 2e-3, (k, err)
    for k, v in head.state_dict().items():
        if "running" in k or k == "step" or "num_batches" in k:
            close(v, g[f"{tag}_after__{k}"], 1e-4, 1e-5, k)
    head.eval()
    with torch.no_grad():
        le, ee = head([c1, None, None, c4])
    close(le, g[f"{tag}_logits_eval"], 1e-4, 1e-4, "eval logits")
    close(ee, g[f"{tag}_emb_eval"], 1e-4, 1e-5, "eval embedding")


def _sync_modules(dst, src):
    dst.load_state_dict({k: v.clone() for k, v in src.state_dict().items()})


@pytest.mark.parametrize("depth,size", [(18, 64), (50, 64)])
def test_backbone_matches_oracle(sa, depth, size):
    from oracle import nets
    from seghiero_amd.backbone import ResNetBackbone
    torch.manual_seed(depth)
    ref = nets.ResNetBackbone(depth, pretrained=False).train()
    mine = ResNetBackbone(depth, pretrained=False)
    _sync_modules(mine, ref)
    mine.to(DEV).train()
    x = torch.randn(2, 3, size, size + 32)
    outs_r = ref(x)
    gs = [torch.randn(o.shape) for o in outs_r]
    sum((o * g).sum() for o, g in zip(outs_r, gs)).backward()
    outs_m = mine(x.to(DEV))
    sum((o * g.to(DEV)).sum() for o, g in zip(outs_m, gs)).backward()
    for i, (a, b) in enumerate(zip(outs_m, outs_r)):
        assert a.shape == b.shape
        assert relerr(a, b) < 2e-5, (i, relerr(a, b))
    gm = dict(mine.named_parameters())
    worst = 0.0
    for k, p in ref.named_parameters():
        e = relerr(gm[k].grad, p.grad)
        worst = max(worst, e)
        assert e < 5e-3, (k, e)
    for k, v in ref.state_dict().items():
        close(mine.state_dict()[k], v, 1e-4, 1e-5, k)


def test_backbone_rejects_bad_input(sa):
    from seghiero_amd.backbone import ResNetBackbone
    with pytest.raises(ValueError):
        ResNetBackbone(77, pretrained=False)
    with pytest.warns(UserWarning):
        ResNetBackbone(18, pretrained=True)


def test_train_steps_match_oracle_config1(sa):
    """BASELINE config 1 shape family: ResNet-18, 4 fine / 2 coarse, B=2 (128x128 crops to keep the CPU side short):
    3 SGD steps, loss within 1e-4 of the oracle each step, parameters afterwards within 1e-4 relative."""
    from oracle.step import OracleTrainer
    from seghiero_amd.synthetic import make_batch
    from seghiero_amd.train_step import SegHieroTrainer
    torch.manual_seed(0)
    kw = dict(depth=18, n_fine=4, coarse_to_fine_map=[[0, 1], [2, 3]], lr=0.01)
    ref = OracleTrainer(**kw)
    mine = SegHieroTrainer(device=DEV, **kw)
    mine.load_state_dicts(ref.state_dicts())
    ref.train(); mine.train()
    for step in range(3):
        img, lab = make_batch(2, 128, 4, seed=step)
        lr_ = ref.train_step(img, lab, epoch=step)
        lm = mine.train_step(img.to(DEV), lab.to(DEV), epoch=step)
        assert abs(float(lm) - float(lr_)) < 1e-4 * max(1.0, abs(float(lr_))), (step, float(lm), float(lr_))
    for name, m in ref.modules().items():
        mm = mine.modules()[name]
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                e = relerr(mm.state_dict()[k], v)
                assert e < 2e-4, (name, k, e)
            else:
                assert torch.equal(mm.state_dict()[k].cpu(), v), (name, k)
    # validation step: loss + pixel-accuracy counts (train.py:341-393)
    ref.eval(); mine.eval()
    img, lab = make_batch(2, 128, 4, seed=9)
    lr_, correct, valid, cm = ref.eval_step(img, lab, 0)
    lm, counts = mine.eval_step(img.to(DEV), lab.to(DEV), 0)
    assert abs(float(lm) - float(lr_)) < 1e-4 * max(1.0, abs(float(lr_)))
    counts = counts.cpu()
    assert int(counts[1]) == valid
    assert abs(int(counts[0]) - correct) <= max(2, valid // 10000)     # argmax may flip on near-ties only
