"""Data-parallel path on the GPU.  (1) Cross-GPU BatchNorm (SyncBN): two ranks, each with half of a batch and SYNC_BN on,
must reproduce the single-rank full-batch forward and (summed over ranks) the parameter gradients of backbone + contrast head.  Both ranks share cuda:0 and exchange over gloo (staged through the host by
seghiero_amd.ddp), which is the rehearsal path for RCCL on a one-GPU box.  (2) The whole DDP training step -- GradSync with
buckets handed over from inside the backward nodes -- on two ranks fed the same batch must reproduce single-rank training
bit for bit (x + x, then the 1/2 folded into SGD, is exact)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HEAD_KW = dict(in_channels=512, c1_in_channels=64, c1_channels=16, aspp_channels=32,
               dilations=(1, 2, 3, 4), num_classes=6, proj_dim=16, proj_type="convmlp")
B, S = 4, 96


def _run(lo, hi, sync, depth=18, B=B, S=S):
    """forward/backward of the ResNet-`depth` trunk + head on images [lo, hi) of the fixed seeded batch"""
    from seghiero_amd import ops
    from seghiero_amd.backbone import ResNetBackbone
    from seghiero_amd.head import DepthwiseSeparableASPPContrastHead
    ops.SYNC_BN = sync
    torch.manual_seed(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    bb = ResNetBackbone(depth=depth).to(dev).train()
    kw = dict(HEAD_KW) if depth < 50 else dict(HEAD_KW, in_channels=2048, c1_in_channels=256)
    head = DepthwiseSeparableASPPContrastHead(**kw).to(dev).train()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 3, S, S, generator=g)
    x = x[lo:hi].to(dev)                                         # the trunk never differentiates the image (stem dgrad skipped)
    logits, emb = head(bb(x))
    gl = torch.randn(B, *logits.shape[1:], generator=g)          # fixed per-image output weights of the full batch
    ge = torch.randn(B, *emb.shape[1:], generator=g)
    ((logits * gl[lo:hi].to(dev)).sum() + (emb * ge[lo:hi].to(dev)).sum()).backward()
    torch.cuda.synchronize()
    named = list(bb.named_parameters()) + list(head.named_parameters())
    return dict(logits=logits.detach().cpu().numpy(), emb=emb.detach().cpu().numpy(),
                grads={k: p.grad.detach().cpu().numpy().copy() for k, p in named if p.grad is not None},
                rm=bb.layer4[1].bn2.running_mean.cpu().numpy(), rv=head.sep_bottleneck[1].bn_pw.running_var.cpu().numpy(),
                rm1=bb.layer1[0].bn3.running_mean.cpu().numpy() if depth >= 50 else None)


def _run_blocks(lo, hi, sync):
    """Bottleneck blocks of ResNet-50 ALONE (layer1.1: identity shortcut; layer2.0: stride-2 downsample branch), forward and
    hand-scheduled backward on images [lo, hi) of a fixed batch: the paths a whole random ResNet-50 is too ill-conditioned to
    pin -- BatchNorm + ReLU in the conv loaders, dgrad-epilogue statistics, the deferred BatchNorm-backward apply (conv3, width >= 16)
    with its coefficients from the all-reduced sums."""
    from seghiero_amd import layers as L, ops
    from seghiero_amd.backbone import ResNetBackbone, _block_bwd, _block_fwd
    ops.SYNC_BN = sync
    torch.manual_seed(3)
    dev = torch.device("cuda", torch.cuda.current_device())
    bb = ResNetBackbone(depth=50, pretrained=False).to(dev).train()
    for m in bb.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.normal_(m.bias, 0.0, 0.2)
    g = torch.Generator().manual_seed(11)
    out = {}
    for name, blk, cin, hw in (("l1.1", bb.layer1[1], 256, 32), ("l2.0", bb.layer2[0], 256, 32)):
        x = torch.randn(B, cin, hw, hw, generator=g)
        y, saved = _block_fwd(blk, ops.to_nhwc(x[lo:hi].to(dev)), True)
        dout = torch.randn(B, *y.shape[1:], generator=g)
        gm = L.GradMap()
        dx = _block_bwd(blk, saved, L.grad_as_nhwc_padded(dout[lo:hi].to(dev), y.shape[1]), gm)
        ops.join_wgrad()
        torch.cuda.synchronize()
        out[name] = dict(y=y.cpu().numpy(), dx=dx.cpu().numpy(),
                         grads={k: gm.g[id(p)].reshape(p.shape).cpu().numpy().copy() for k, p in blk.named_parameters()},
                         rm=blk.bn3.running_mean.cpu().numpy())
    return out


def _block_worker(rank, world, port, q, cuts):
    _env(rank, world, port, "gloo")
    import torch.distributed as dist
    from seghiero_amd import ddp
    ddp.init_from_env(backend="gloo")
    try:
        out = _run_blocks(cuts[rank], cuts[rank + 1], sync=True)
    except Exception as e:
        import traceback
        out = "rank %d failed: %s\n%s" % (rank, e, traceback.format_exc())
    q.put((rank, out))
    if not isinstance(out, str):
        dist.barrier()
    dist.destroy_process_group()


def _backends():
    """gloo with both ranks on cuda:0 (host-staged rehearsal, runs on the one-GPU box) and, where the node has two GPUs,
    RCCL ("nccl") with one device per rank."""
    return ["gloo"] + (["nccl"] if torch.cuda.device_count() >= 2 else [])


def _env(rank, world, port, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank) if backend == "nccl" else "0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        torch.cuda.set_device(rank)


def _worker(rank, world, port, q, backend="gloo", cuts=None, depth=18, batch=B, size=S):
    _env(rank, world, port, backend)
    import torch.distributed as dist
    from seghiero_amd import ddp
    ddp.init_from_env(backend=backend)
    cuts = cuts or [r * (batch // world) for r in range(world + 1)]
    try:
        out = _run(cuts[rank], cuts[rank + 1], sync=True, depth=depth, B=batch, S=size)
    except Exception as e:                                       # report instead of leaving the parent to time out
        import traceback
        out = "rank %d failed: %s\n%s" % (rank, e, traceback.format_exc())
    q.put((rank, out))
    if not isinstance(out, str):
        dist.barrier()
    dist.destroy_process_group()


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("backend,cuts,depth", [("gloo", None, 18), ("gloo", [0, 3, 4], 18), ("nccl", None, 18)])
def test_syncbn_two_ranks_equals_full_batch(backend, cuts, depth):
    """cuts = [0, 3, 4]: rank 0 holds three images and rank 1 one -- the pixel counts travel with the sums, so uneven
    shards give the full-batch statistics too.  (A randomly initialised ResNet-50 amplifies a 1e-6 relative perturbation of its
    input to 5 % of its parameter gradients at any batch / image size -- tests/diag/sens50.py -- so the Bottleneck paths are
    checked block by block: test_syncbn_bottleneck_blocks_two_ranks.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    if backend not in _backends():
        pytest.skip("two ranks over RCCL need two GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150 + (3 if cuts else 0) + (7 if depth >= 50 else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend, cuts, depth)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    for r in res.values():
        assert not isinstance(r, str), r
    full = _run(0, B, sync=False, depth=depth)
    for k in ("logits", "emb"):
        got = np.concatenate([res[0][k], res[1][k]], 0)
        assert _rel(got, full[k]) < 2e-4, k          # fp32 sums in a different order; conv itself is the same arithmetic
    # running statistics use the global batch on every rank
    for k in ("rm", "rv") + (("rm1",) if depth >= 50 else ()):
        np.testing.assert_allclose(res[0][k], res[1][k], rtol=0, atol=0)
        np.testing.assert_allclose(res[0][k], full[k], rtol=1e-4, atol=1e-6)
    assert set(res[0]["grads"]) == set(full["grads"])
    worst = max(_rel(res[0]["grads"][k] + res[1]["grads"][k], full["grads"][k]) for k in full["grads"])
    tot = _rel(np.concatenate([(res[0]["grads"][k] + res[1]["grads"][k]).ravel() for k in full["grads"]]),
               np.concatenate([full["grads"][k].ravel() for k in full["grads"]]))
    assert tot < 5e-4 and worst < 5e-3, (tot, worst)


TR_KW = dict(depth=18, n_fine=4, coarse_to_fine_map=[[0, 1], [2, 3]], lr=0.01,
             head_kw=dict(c1_channels=16, aspp_channels=32, dilations=(1, 2, 3, 4), proj_dim=16))


def _train(n_steps, sync, syncbn=False):
    from seghiero_amd import ddp, ops
    from seghiero_amd.synthetic import make_batch
    from seghiero_amd.train_step import SegHieroTrainer
    torch.manual_seed(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.SYNC_BN = syncbn
    b16 = os.environ.get("SEGHIERO_TEST_B16") == "1"              # (inherited by the spawned ranks)
    tr = SegHieroTrainer(device=dev, compute_dtype=torch.bfloat16 if b16 else torch.float32, **TR_KW)
    if sync:
        ddp.broadcast_module_state(list(tr.modules().values()))
        tr.grad_sync = ddp.GradSync(tr.params, bucket_mb=4.0)
    img, lab = make_batch(2, 64, 4, seed=5, device=dev)
    losses = [float(tr.train_step(img, lab, 0)) for _ in range(n_steps)]
    torch.cuda.synchronize()
    launched = len(tr.grad_sync.buckets) if sync else 0
    return losses, [p.detach().cpu().numpy().copy() for p in tr.params], launched


def _ddp_worker(rank, world, port, q, backend="gloo"):
    _env(rank, world, port, backend)
    import torch.distributed as dist
    from seghiero_amd import ddp
    ddp.init_from_env(backend=backend)
    try:
        out = _train(2, sync=True)
    except Exception as e:
        import traceback
        out = "rank %d failed: %s\n%s" % (rank, e, traceback.format_exc())
    q.put((rank, out))
    if not isinstance(out, str):
        dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,mode", [("gloo", "f32"), ("gloo", "b16"), ("nccl", "f32")])
def test_ddp_two_ranks_same_batch_equals_single_rank(backend, mode, monkeypatch):
    """... also in bf16 compute mode (r3): the exchange works on the fp32 weight gradients whatever the compute mode."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    if backend not in _backends():
        pytest.skip("two ranks over RCCL need two GPUs (RCCL refuses two ranks on one device)")
    monkeypatch.setenv("SEGHIERO_TEST_B16", "1" if mode == "b16" else "0")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 150
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    for r in res.values():
        assert not isinstance(r, str), r
    losses, params, _ = _train(2, sync=False)
    for rank in (0, 1):
        l2, p2, nb = res[rank]
        assert nb >= 3                                  # several buckets => some left during backward
        assert l2 == losses, (rank, l2, losses)
        for a, b in zip(p2, params):
            np.testing.assert_array_equal(a, b)


# ---------------------------------------------------------------- the RCCL code path itself, on the one-GPU box
def _rccl_world1_worker(port, q):
    """One rank, backend "nccl" (= RCCL): with ddp.FORCE_COLLECTIVES every collective of the N > 1 step is issued for real --
    bucketed gradient all-reduce with async_op on the side stream straight from the arena views, the f64 SyncBN
    all-reduces (sums + count) and the class_count MIN-reduce on the small-message group, the parameter broadcast."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        import torch.distributed as dist
        from seghiero_amd import ddp
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        ddp.SMALL_GROUP = True                                          # the second communicator, serialised behind the buckets
        ddp.init_small_group()
        ddp.FORCE_COLLECTIVES = True
        assert dist.get_backend() == "nccl" and dist.get_backend(ddp.small_group()) == "nccl"
        t = torch.tensor([3.0, 5.0, 7.0], device="cuda:0", dtype=torch.float64)
        ddp.all_reduce_small(t)                                         # f64 SUM through RCCL
        r = torch.tensor([2.0], device="cuda:0")
        ddp.all_reduce_small(r[0:1], op=dist.ReduceOp.MIN)               # MIN on a 1-element slice
        assert t.tolist() == [3.0, 5.0, 7.0] and r.item() == 2.0
        out = {"plain": _train(2, sync=True), "syncbn": _train(2, sync=True, syncbn=True)}
        torch.cuda.synchronize()
        ddp.shutdown()
    except Exception as e:
        import traceback
        out = "rccl worker failed: %s\n%s" % (e, traceback.format_exc())
    q.put(out)


def test_rccl_backend_world1_runs_every_collective_of_the_step():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(29250 + os.getpid() % 150, q))
    p.start()
    res = q.get(timeout=300)
    p.join(60)
    assert not isinstance(res, str), res
    losses, params, _ = _train(2, sync=False)
    l2, p2, nb = res["plain"]
    assert nb >= 3
    assert l2 == losses                                  # an all-reduce over one rank is the identity: bit-equal training
    for a, b in zip(p2, params):
        np.testing.assert_array_equal(a, b)
    # SyncBN finalizes from f64 global sums instead of the local centred partials: same statistics up to fp32 rounding
    l3, p3, _ = res["syncbn"]
    np.testing.assert_allclose(l3, losses, rtol=2e-5)
    tot = _rel(np.concatenate([a.ravel() for a in p3]), np.concatenate([b.ravel() for b in params]))
    assert tot < 1e-4, tot


@pytest.mark.parametrize("cuts", [[0, 2, 4], [0, 1, 4]])
def test_syncbn_bottleneck_blocks_two_ranks(cuts):
    """SyncBN through the fused Bottleneck paths (see _run_blocks): two ranks with 2 + 2 or 1 + 3 images reproduce the single-rank
    full batch -- outputs, input gradients (per-image rows), summed parameter gradients, running statistics."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 150 + cuts[1]
    procs = [ctx.Process(target=_block_worker, args=(r, 2, port, q, cuts)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    for r in res.values():
        assert not isinstance(r, str), r
    full = _run_blocks(0, B, sync=False)
    for name in full:
        for k in ("y", "dx"):
            got = np.concatenate([res[0][name][k], res[1][name][k]], 0)
            assert _rel(got, full[name][k]) < 2e-5, (name, k, _rel(got, full[name][k]))
        np.testing.assert_allclose(res[0][name]["rm"], res[1][name]["rm"], rtol=0, atol=0)
        np.testing.assert_allclose(res[0][name]["rm"], full[name]["rm"], rtol=1e-4, atol=1e-6)
        for k, want in full[name]["grads"].items():
            e = _rel(res[0][name]["grads"][k] + res[1][name]["grads"][k], want)
            assert e < 2e-4, (name, k, e)


# ---------------------------------------------------------------- the whole trainer step, different shards per rank, SyncBN on
SH_B, SH_S, SH_EPOCH = 4, 256, 40000          # epoch 40000: the triplet term is live (cosine factor 0.25)


def _sharded_step(cuts, rank, world, size=None, batch=None, syncbn=True):
    size, batch = size or SH_S, batch or SH_B
    from seghiero_amd import ddp, ops
    from seghiero_amd.synthetic import make_batch
    from seghiero_amd.train_step import SegHieroTrainer
    torch.manual_seed(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.SYNC_BN = syncbn
    tr = SegHieroTrainer(device=dev, **TR_KW)
    init = {k: {n: v.detach().cpu().numpy().copy() for n, v in m.state_dict().items()} for k, m in tr.modules().items()}      # (numpy: pickled by value)
    if world > 1:
        ddp.broadcast_module_state(list(tr.modules().values()))
        tr.grad_sync = ddp.GradSync(tr.params, bucket_mb=4.0)
    tr.train()
    img, lab = make_batch(batch, size, 4, seed=5)
    a, b = cuts[rank], cuts[rank + 1]
    loss = float(tr.train_step(img[a:b].to(dev), lab[a:b].to(dev), SH_EPOCH))
    torch.cuda.synchronize()
    named = [(f"{mk}.{k}", p) for mk, m in tr.modules().items() for k, p in m.named_parameters()]
    grads = {k: (p.grad.detach().cpu().numpy() / world).copy() for k, p in named}          # 1/world is folded into the SGD kernel
    rstat = {f"{mk}.{k}": v.cpu().numpy().copy() for mk, m in tr.modules().items() for k, v in m.state_dict().items() if "running" in k}
    return dict(loss=loss, grads=grads, init=init if rank == 0 else None, rstat=rstat)


def _sharded_worker(rank, world, port, q, cuts, size=None, batch=None, syncbn=True):
    _env(rank, world, port, "gloo")
    import torch.distributed as dist
    from seghiero_amd import ddp
    ddp.init_from_env(backend="gloo")
    try:
        out = _sharded_step(cuts, rank, world, size, batch, syncbn)
    except Exception as e:
        import traceback
        out = "rank %d failed: %s\n%s" % (rank, e, traceback.format_exc())
    q.put((rank, out))
    if not isinstance(out, str):
        dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cuts", [[0, 2, 4], [0, 1, 4]])
def test_ddp_sharded_step_with_syncbn_matches_oracle_ddp_statement(cuts):
    """Two ranks, DIFFERENT shards (2 + 2 and 1 + 3 images of one seeded batch), SyncBN on, the whole trainer step (trunk, head, aux
    head, 2-level loss with the live triplet term, backward with gradients handed over from inside the backward nodes, all-reduce)
    against the oracle's statement of the data-parallel step (oracle/step.py:ddp_forward_loss: full-batch BatchNorm through trunk and
    heads on CPU, per-shard losses with per-shard normalisers -- hiera_triplet_loss.py:41-107 num_valid, utils.py:20-21 all-pixel
    mean --, per-shard triplets with the all-ranks `ready` rule of :193-198, mean over ranks; train.py:260-320):

    * every rank's loss within 1e-4 ABSOLUTE of the oracle's loss for that shard;
    * averaged gradients against an fp64 run of the same statement, with the fp32 oracle's own distance from it as the yardstick:
      the head + aux-head gradient vector and the whole vector (trunk included) within 4x (+1e-5) in relative L2; per head / aux tensor
      the MEDIAN over output channels of the per-channel error within 6x the fp32 oracle's median (+2e-4 of the tensor's scale; c1_bottleneck,
      fed by the ill-conditioned random trunk, measures 4.8x).  The
      median is the robust form of "every element": a semantic error (a wrong normaliser, a wrong BatchNorm sum) moves every channel
      of a tensor, a ReLU whose pre-activation two fp32 evaluations round to different sides of 0 moves ONE channel by percents
      (tests/diag/ddp_sharded.py: at most a handful of channels per tensor, on either side); those are counted and bounded at 5 % of
      a tensor's channels;
    * running statistics equal on both ranks and equal to the full-batch oracle's."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import copy
    import torch.multiprocessing as mp
    from oracle.step import OracleTrainer
    from seghiero_amd.synthetic import make_batch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29050 + os.getpid() % 150 + cuts[1]
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, cuts)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    for r in res.values():
        assert not isinstance(r, str), r
    img, lab = make_batch(SH_B, SH_S, 4, seed=5)
    init = {k: {n: torch.from_numpy(v) for n, v in sd.items()} for k, sd in res[0]["init"].items()}
    ref = OracleTrainer(**TR_KW)
    for k, m in ref.modules().items():
        m.load_state_dict(init[k])
    ref64 = OracleTrainer(**TR_KW)
    for k, m in ref64.modules().items():
        m.load_state_dict(init[k])
        m.double()
    ref.train(); ref64.train()
    mean32, tot32, _, _ = ref.ddp_forward_loss(img, lab, SH_EPOCH, cuts)
    mean32.backward()
    mean64, _, _, _ = ref64.ddp_forward_loss(img.double(), lab, SH_EPOCH, cuts)
    mean64.backward()
    for r in (0, 1):
        assert abs(res[r]["loss"] - float(tot32[r])) < 1e-4, (r, res[r]["loss"], float(tot32[r]))
    names = [(f"{mk}.{k}", p) for mk, m in ref.modules().items() for k, p in m.named_parameters()]
    g64 = {f"{mk}.{k}": p.grad.numpy() for mk, m in ref64.modules().items() for k, p in m.named_parameters()}
    for k, _ in names:
        np.testing.assert_array_equal(res[0]["grads"][k], res[1]["grads"][k])             # the all-reduce left both ranks with the same sums
    g32 = {k: p.grad.numpy() for k, p in names}
    cat = lambda d, ks: np.concatenate([np.asarray(d[k], np.float64).ravel() for k in ks])
    allk = [k for k, _ in names]
    heads = [k for k in allk if not k.startswith("backbone.")]
    for ks in (heads, allk):
        e_m, e_r = _rel(cat(res[0]["grads"], ks), cat(g64, ks)), _rel(cat(g32, ks), cat(g64, ks))
        assert e_m < 4 * e_r + 1e-5, (len(ks), e_m, e_r)
    bad = []
    for k in heads:
        t = g64[k]
        scale = max(float(np.abs(t).max()), 1e-3)
        per = lambda a: np.abs(a - t).reshape(t.shape[0], -1).max(1) / scale
        em, er = per(res[0]["grads"][k]), per(g32[k])
        bound = 6 * float(np.median(er)) + 2e-4
        flipped = int((em > 10 * bound).sum())
        if not (float(np.median(em)) < bound and flipped <= max(1, t.shape[0] // 20)):
            bad.append((k, float(np.median(em)), float(np.median(er)), flipped, t.shape[0]))
    assert not bad, bad
    want = {f"{mk}.{k}": v.numpy() for mk, m in ref.modules().items() for k, v in m.state_dict().items() if "running" in k}
    for k, v in want.items():
        np.testing.assert_array_equal(res[0]["rstat"][k], res[1]["rstat"][k])
        np.testing.assert_allclose(res[0]["rstat"][k], v, rtol=2e-4, atol=2e-6, err_msg=k)


# ---------------------------------------------------------------- exact data-parallel mode: sharded == full batch
def _exact_step(cuts, rank, world):
    from seghiero_amd import ddp, ops
    from seghiero_amd.synthetic import make_batch
    from seghiero_amd.train_step import SegHieroTrainer
    torch.manual_seed(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.SYNC_BN = world > 1
    ddp.EXACT = world > 1
    tr = SegHieroTrainer(device=dev, **TR_KW)
    if world > 1:
        ddp.broadcast_module_state(list(tr.modules().values()))
        tr.grad_sync = ddp.GradSync(tr.params, bucket_mb=4.0)
    tr.train()
    img, lab = make_batch(4, 128, 4, seed=5)
    a, b = cuts[rank], cuts[rank + 1]
    loss = float(tr.train_step(img[a:b].to(dev), lab[a:b].to(dev), 0))          # epoch 0: the (per-rank by design) triplet term has weight 0
    torch.cuda.synchronize()
    ddp.EXACT = False
    named = [(f"{mk}.{k}", p) for mk, m in tr.modules().items() for k, p in m.named_parameters()]
    return dict(loss=loss, grads={k: p.grad.detach().cpu().numpy().copy() for k, p in named},
                after={k: p.detach().cpu().numpy().copy() for k, p in named})


def _exact_worker(rank, world, port, q, cuts):
    _env(rank, world, port, "gloo")
    import torch.distributed as dist
    from seghiero_amd import ddp
    ddp.init_from_env(backend="gloo")
    try:
        out = _exact_step(cuts, rank, world)
    except Exception as e:
        import traceback
        out = "rank %d failed: %s\n%s" % (rank, e, traceback.format_exc())
    q.put((rank, out))
    if not isinstance(out, str):
        dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cuts", [[0, 2, 4], [0, 1, 4]])
def test_ddp_exact_normalisers_sharded_equals_full_batch(cuts):
    """Exact data-parallel mode (ddp.EXACT): the ranks all-reduce the 24-byte vector of loss normalisers (sh_label_counts:
    valid fine / valid coarse / all pixels -- hiera_triplet_loss.py:41-107 num_valid, utils.py:20-21, nn.CrossEntropyLoss) and divide
    their LOCAL numerators by the GLOBAL denominators; with SyncBN on, two ranks holding 2 + 2 or 1 + 3 images then reproduce the
    single-process step on the whole batch: the per-rank losses SUM to the full-batch loss (1e-5 relative), the summed gradients and
    the parameters after the SGD step equal the single-process ones (the gradient vector 5e-4, any one tensor 5e-3: fp32 sums in
    another order, as test_syncbn_two_ranks_equals_full_batch)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 28900 + os.getpid() % 150 + cuts[1]
    procs = [ctx.Process(target=_exact_worker, args=(r, 2, port, q, cuts)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    for r in res.values():
        assert not isinstance(r, str), r
    full = _exact_step([0, 4], 0, 1)
    total = res[0]["loss"] + res[1]["loss"]
    assert abs(total - full["loss"]) < 1e-5 * abs(full["loss"]), (res[0]["loss"], res[1]["loss"], full["loss"])
    ks = list(full["grads"])
    for k in ks:
        np.testing.assert_array_equal(res[0]["grads"][k], res[1]["grads"][k])          # summed by the all-reduce, not averaged
    tot = _rel(np.concatenate([res[0]["grads"][k].ravel() for k in ks]), np.concatenate([full["grads"][k].ravel() for k in ks]))
    worst = max((_rel(res[0]["grads"][k], full["grads"][k]), k) for k in ks)
    assert tot < 5e-4 and worst[0] < 5e-3, (tot, worst)
    upd = _rel(np.concatenate([res[0]["after"][k].ravel() for k in ks]), np.concatenate([full["after"][k].ravel() for k in ks]))
    assert upd < 1e-6, upd


def test_c_abi_rccl_communicator_world1_runs_every_entry_point():
    """sh_comm_* (the RCCL communicator for a host without PyTorch): at world size 1 every entry point runs for real on the GPU --
    id, init, in-order and side-stream all-reduces of each dtype / op with the event hand-off, wait, broadcast, destroy -- and leaves the
    data what a one-rank reduction must leave.  (More than one rank needs more than one GPU: unmeasured, like the torch.distributed path.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from seghiero_amd.comm import Comm
    from seghiero_amd._lib import SegHieroHipError
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    comm = Comm(Comm.unique_id(), 1, 0)
    try:
        g = torch.Generator().manual_seed(3)
        for dt in (torch.float32, torch.float64, torch.int64):
            for op in ("sum", "min", "max"):
                ref = (torch.randn(100003, generator=g) * 100).to(dt)
                t = ref.to(dev)
                comm.all_reduce(t, op)
                assert torch.equal(t.cpu(), ref), (dt, op)
        # side stream: the reduction must see what the compute stream produced before it, and the consumer must see the result
        bucket = torch.zeros(1 << 22, device=dev)
        for k in range(3):
            bucket.add_(1.0)                              # producer kernel on the compute stream
            comm.all_reduce_async(bucket)
            comm.wait()
            bucket.mul_(2.0)                              # consumer kernel on the compute stream
        torch.cuda.synchronize()
        assert float(bucket.min()) == float(bucket.max()) == 14.0        # ((0 + 1) * 2 + 1) * 2 + 1) * 2
        w = torch.arange(1000, device=dev, dtype=torch.float32)
        comm.broadcast(w, 0)
        assert torch.equal(w.cpu(), torch.arange(1000, dtype=torch.float32))
        with pytest.raises(SegHieroHipError):
            comm.all_reduce(torch.zeros(4, device=dev, dtype=torch.float16))
    finally:
        comm.close()
