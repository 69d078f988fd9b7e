"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares, the host-side
helpers are bit-exact against the reference goldens, module surfaces (names, state_dict keys, error behaviour) match,
and the compute entry points fail loudly instead of falling back when there is no GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import ctypes

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from seghiero_amd._lib import LIBPATH
    if not os.path.exists(LIBPATH):
        subprocess.check_call(["bash", os.path.join(ROOT, "seghiero_amd", "csrc", "build.sh")])
    return LIBPATH


def test_abi_exports_every_declared_symbol(built):
    from seghiero_amd._lib import HEADER, LIB
    declared = set(re.findall(r"\b(sh_\w+)\s*\(", re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)))
    assert len(declared) >= 45
    assert declared == set(LIB.protos), declared ^ set(LIB.protos)
    assert set(LIB.exported_symbols()) == declared          # dlsym succeeded for each of them
    assert LIB.raw("sh_abi_version")() == 1                  # host-only calls work without a GPU
    assert LIB.raw("sh_conv_tile_rows")() == 64
    assert LIB.raw("sh_conv_wgrad_workspace")(16, 128, 128, 64, 64, 3, 3, 1, 1, 1) > 0
    assert LIB.raw("sh_conv_wgrad_workspace")(16, 128, 128, 3, 64, 3, 3, 1, 1, 1) == -1     # Cin % 4 != 0 rejected


def test_abi_rejects_bad_arguments_without_launching(built):
    from seghiero_amd._lib import LIB
    assert LIB.raw("sh_conv_fprop")(None, 4, None, None, None, 4, None, 1, 8, 8, 4, 4, 1, 1, 1, 0, 1, None) == -1
    assert LIB.raw("sh_sgd_step")(0, None, None, None, None, 0.1, 0.9, 0.0, 1, 1.0, None) == -1
    assert LIB.raw("sh_fill")(None, 0.0, 10, None) == -1
    # round-2 entry points: NULL operands are rejected before anything is launched (no GPU needed to find out)
    assert LIB.raw("sh_conv_dgrad_x6_lin")(None, 64, None, 64, None, None, None, 0, None, 64, None, 0, None, None, None, None, 0, None,
                                           1, 8, 8, 64, 64, None, 0, 0, None) == -1
    assert LIB.raw("sh_conv_wgrad_x6_lin")(None, 64, None, None, None, 64, None, 64, None, None, None, 1, 8, 8, 64, 64, 1, 1, 1, 0, 1, 0, None) == -1
    assert LIB.raw("sh_dwconv_dgrad")(None, 64, None, 0, None, None, None, 64, 1, 8, 8, 64, 1, 0, 0, None) == -1
    assert LIB.raw("sh_bn_bwd_reduce")(None, 64, None, 0, None, 64, None, None, None, None, None, 64, 64, 0, None, 0, 0, None) == -1
    assert LIB.raw("sh_bilinear_bwd")(None, 64, None, 64, 1, 4, 4, 8, 8, 64, None, 0, None) == -1
    assert LIB.raw("sh_hiera2_loss_fwd")(None, 16, None, None, 9, 4, None, None, None, None, 1, 8, 8, 32, 32, None, 0, 0, None, None) == -1
    assert LIB.raw("sh_bn_fold_partials")(None, 4096, 64, 262144.0, 64, 64, None, None) == -1
    assert LIB.raw("sh_bilinear_bwd_workspace")(0, 4, 4, 64) == -1 and LIB.raw("sh_bilinear_bwd_workspace")(2, 4, 4, 64) == 2 * 2 * 16 * 64 * 4
    assert LIB.raw("sh_resize_bilinear_coeffs")(0, 8, None, None, 0) < 0
    # round-3 entry points and flags (argument checks precede every launch: dummy non-NULL pointers are never dereferenced on the host)
    P = 4096
    dg = LIB.raw("sh_conv_dgrad_b16")
    def dgrad(kh, stride, pad, flags, addend=None, n=1):
        return dg(P, 64, None, 0, None, P, addend, 64 if addend else 0, P, 64, None, 0, None, 0, None, None, None, None, 0, None,
                  n, 8, 8, 64, 64, kh, kh, stride, pad, 1, None, 0, flags, None)
    assert dgrad(1, 2, 0, 128) == -1                      # unknown act_flags bit
    assert dgrad(3, 2, 1, 64) == -1                       # scatter-add is the 1x1 strided conv's form
    assert dgrad(1, 2, 0, 64, addend=P) == -1             # ... and takes no hooks
    assert dgrad(3, 2, 1, 0, addend=P) == -3              # stride-2 KxK with an addend: SH_EUNSUPPORTED (the fp32-accurate entry point takes it)
    assert dgrad(3, 3, 1, 0) == -3                        # stride 3: no parity plan
    h3 = LIB.raw("sh_hiera3_loss_fwd")
    f2m = (ctypes.c_int * 7)(0, 1, 1, 1, 1, 2, 2); f2h = (ctypes.c_int * 7)(0, 1, 1, 1, 1, 1, 1)
    assert h3(P, 12, P, f2m, f2h, 7, 3, 2, P, P, P, None, None, None, 1, 8, 8, 32, 32, P, 0, 16, None) == -1      # grad_out too small
    assert h3(P, 12, P, f2m, f2h, 7, 3, 2, P, P, P, None, None, None, 1, 8, 8, 32, 32, P, 1 << 20, 12, None) == -1  # ldg must be 16 or 32
    assert LIB.raw("sh_comm_unique_id")(None) == -1 and LIB.raw("sh_comm_init")(None, 2, 0, None) == -1
    assert LIB.raw("sh_comm_all_reduce")(None, P, 4, 0, 0, None) == -1 and LIB.raw("sh_comm_wait")(None, None) == -1
    h3b = LIB.raw("sh_hiera3_loss_bwd")
    assert h3b(P, 12, P, f2m, f2h, 7, 3, 2, P, P, 0.1, None, 1.0, P, 16, 1, 8, 8, 32, 32, P, 1 << 20, 1, None, None) == -1   # dprob without probs


def test_header_is_plain_c_and_links_from_c(built, tmp_path):
    """include/seghiero_hip.h is the boundary a non-Python host binds: it must compile as C99 and as C++, and a C program that
    takes the address of entry points must link against the shared library (no C++ name mangling, no torch types)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "seghiero_hip.h")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr])
    subprocess.check_call(["g++", "-fsyntax-only", "-x", "c++", hdr])
    src = tmp_path / "host.c"
    src.write_text('#include "seghiero_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { void* f[] = {(void*)sh_conv_fprop_x6, (void*)sh_conv_dgrad_x6_lin, (void*)sh_hiera2_loss_fwd,\n'
                   '                              (void*)sh_bn_finalize, (void*)sh_sgd_step};\n'
                   '  printf("%d\\n", (int)(sizeof f / sizeof f[0]) + (sh_stats_tile_rows() > 0)); return 0; }\n')
    lib = os.path.join(root, "seghiero_amd", "libseghiero_hip.so")
    exe = tmp_path / "host"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.dirname(hdr), str(src), lib, "-o", str(exe), "-Wl,-rpath," + os.path.dirname(lib),
                           "-Wl,--allow-shlib-undefined"])


def test_hierarchy_helpers_bit_exact_vs_reference_golden(golden):
    import seghiero_amd as sa
    g = golden("g1_maps")
    cfgs = {"a": ([[0, 3], [4, 6], [7], [8]], 9), "b": ([[0, 1], [2, 3]], 4), "c": ([[0], [1, 4], [5, 6]], 7)}
    for k, (cfg, nf) in cfgs.items():
        assert np.array_equal(sa.build_fine_to_coarse_map(cfg, nf).numpy(), g[f"{k}_f2c"])
        assert sa.build_fine_to_coarse_map(cfg, nf).dtype == torch.long
        assert np.array_equal(np.asarray(sa.build_hiera_index(cfg)), g[f"{k}_hidx"])
    assert np.array_equal(sa.build_fine_to_super_map([[0], [1, 6]], 7).numpy(), g["c_f2s"])
    with pytest.raises(ValueError):
        sa.build_fine_to_super_map([[0, 1], [2, 3]], 9)


def test_module_surface_matches_reference(golden):
    from seghiero_amd.backbone import ResNetBackbone
    from seghiero_amd.head import AuxHead, DepthwiseSeparableASPPContrastHead
    from oracle import nets
    g = golden("g3_head")
    kw = dict(in_channels=64, c1_in_channels=16, c1_channels=8, aspp_channels=16, dilations=(1, 12, 24, 36),
              num_classes=6, proj_dim=8, proj_type="convmlp")
    head = DepthwiseSeparableASPPContrastHead(**kw)
    ref_keys = {k[4:]: v.shape for k, v in g.items() if k.startswith("sd__")}     # keys/shapes of the REFERENCE head
    assert {k: tuple(v.shape) for k, v in head.state_dict().items()} == {k: tuple(s) for k, s in ref_keys.items()}
    full = DepthwiseSeparableASPPContrastHead(2048, 256, 48, 512, (1, 12, 24, 36), 13)
    assert len(full.state_dict()) == 94 and sum(p.numel() for p in full.parameters()) == 11925104 + 513 * 13
    with pytest.raises(ValueError):
        DepthwiseSeparableASPPContrastHead(**{**kw, "proj_type": "mlp"})
    torch.manual_seed(3)                      # seed-for-seed default init == the reference's (incl. discarded draws)
    h2 = DepthwiseSeparableASPPContrastHead(**kw)
    for k, v in h2.state_dict().items():
        if v.dim() == 4 or k == "cls_seg.bias":
            np.testing.assert_array_equal(v.numpy(), g["sd__" + k], err_msg=k)
    for depth, n in ((18, 11176512), (50, 23508032), (101, 42500160)):
        with pytest.warns(UserWarning):
            bb = ResNetBackbone(depth)            # pretrained=True default: warns, never fetches
        assert sum(p.numel() for p in bb.parameters()) == n
        ob = nets.ResNetBackbone(depth, pretrained=False)
        assert {k: tuple(v.shape) for k, v in bb.state_dict().items()} == {k: tuple(v.shape) for k, v in ob.state_dict().items()}
    with pytest.raises(ValueError):
        ResNetBackbone(77, pretrained=False)
    assert list(AuxHead(1024, 9).state_dict()) == list(nets.make_aux_head(1024, 9).state_dict())


def test_state_dict_roundtrip_keeps_native_layout():
    from seghiero_amd.backbone import ResNetBackbone
    from oracle import nets
    ob = nets.ResNetBackbone(18, pretrained=False)
    bb = ResNetBackbone(18, pretrained=False)
    bb.load_state_dict(ob.state_dict())
    w = bb.layer1[0].conv1.weight
    assert torch.equal(w, ob.layer1[0].conv1.weight)                       # same logical values
    assert w.is_contiguous(memory_format=torch.channels_last)             # OHWI memory for the kernels
    assert bb.stem_conv.weight.is_contiguous()


def test_no_cpu_fallback():
    """The product must raise on CPU tensors -- there is no eager / oracle fallback behind it."""
    import seghiero_amd as sa
    from seghiero_amd.backbone import ResNetBackbone
    from seghiero_amd.loss import HieraTripletLoss
    bb = ResNetBackbone(18, pretrained=False)
    with pytest.raises(sa.SegHieroHipError, match="MI355X"):
        bb(torch.randn(1, 3, 32, 32))
    loss = HieraTripletLoss(4, [0, 0, 1, 1], [[0, 2], [2, 4]])
    with pytest.raises(sa.SegHieroHipError):
        loss(0, torch.randn(1, 8, 2, 2), None, torch.randn(1, 6, 8, 8), torch.zeros(1, 8, 8, dtype=torch.long))
    src = "".join(open(os.path.join(ROOT, "seghiero_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "seghiero_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_triplet_tables_and_factor():
    from seghiero_amd.loss import triplet_factor, two_level_triplet_tables
    masks, ok = two_level_triplet_tables([0, 0, 1, 1], [[0, 2], [2, 4]])
    def members(words):
        return {v for v in range(256) if (int(words[v >> 6]) >> (v & 63)) & 1}
    assert members(masks[0, 0]) == {1} and members(masks[3, 0]) == {2}
    assert members(masks[0, 1]) == set(range(2, 256))                      # negatives include 255 (reference :36)
    assert members(ok) == {0, 1, 2, 3}
    assert triplet_factor(0, 80000) == 0.0 and triplet_factor(80000, 80000) == 0.5
    assert abs(triplet_factor(40000, 80000) - 0.25) < 1e-12


def _ddp_worker(rank, world, port, q, small=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from seghiero_amd import ddp
    ddp.SMALL_GROUP = small
    ddp.init_from_env(backend="gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(s)) for s in [(7,), (64, 16, 3, 3), (13, 32, 1, 1), (300,)]]
    params[1].data = params[1].data.contiguous(memory_format=torch.channels_last)
    sync = ddp.GradSync(params, bucket_mb=0.01)
    assert len(sync.buckets) >= 2
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.empty_like(p).copy_(torch.randn(p.shape, generator=g)) for p in params]
    # protocol of a training step: begin, the backward nodes hand over finished gradients early (last parameters first),
    # autograd then stores p.grad, reduce() sends the rest and repoints p.grad at the reduced arena
    sync.begin()
    # a backward node asks for the arena slice of a parameter and writes the gradient in place (layers.new_grad): the
    # hand-over then stages nothing for it
    view = ddp.grad_buffer(params[3])
    assert view is not None and view.shape == params[3].shape and view.data_ptr() == sync.flat.data_ptr() + 4 * sync.views[id(params[3])][0]
    assert (view.data_ptr() - sync.flat.data_ptr()) % 256 == 0 and view.data_ptr() % 16 == 0 and all(o % ddp.GradSync.ARENA_ALIGN == 0 for o, _ in sync.views.values())     # 16-byte stores of the wgrad kernels
    assert ddp.grad_buffer(params[3]) is None        # a slice is handed out once per step (a second gradient must not overwrite the first)
    view.copy_(grads[3])
    ddp.early_flush([(params[3], view)])
    assert ddp.grad_buffer(params[3]) is None                                   # handed over: no second writer
    v1 = ddp.grad_buffer(params[1])
    assert v1.stride() == params[1].stride()                                    # channels_last weight: same strides as the parameter
    v1.copy_(grads[1])
    ddp.early_flush([(params[2], grads[2]), (params[1], v1)])
    assert len(sync._launched) >= 1                 # at least one bucket went out before the "backward" ended
    for p, gr in zip(params, grads):
        p.grad = gr
    scale = sync.reduce(params)
    assert ddp._ACTIVE is None
    # latency-class collectives (SyncBN sums + count in f64, triplet class_count MIN): on the default group, or -- behind the
    # SEGHIERO_SMALL_GROUP switch -- on a process group of their own
    if small:
        assert ddp.small_group() is not None and ddp.small_group() is not dist.group.WORLD
    else:
        assert ddp.small_group() is None
    sq = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 5.0 + rank], dtype=torch.float64)      # [sum, sum^2, local count]
    ddp.all_reduce_small(sq)
    assert sq.tolist() == [3.0, 30.0, 11.0]
    ready = torch.tensor([float(rank)])
    ddp.all_reduce_small(ready, op=dist.ReduceOp.MIN)
    assert ready.item() == 0.0
    m = torch.nn.Linear(3, 2)
    with torch.no_grad():
        m.weight.fill_(float(rank))
    ddp.broadcast_module_state([m])
    q.put((rank, scale, [p.grad.detach().numpy().copy() for p in params], float(m.weight.detach().sum())))
    ddp.shutdown()


@pytest.mark.parametrize("small", [False, True])
def test_gradsync_two_ranks_gloo(small):
    """N>1 path on CPU: bucketed all-reduce over gloo == sum of the per-rank gradients; broadcast from rank 0."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200 + (211 if small else 0)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, small)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    expect = []
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        expect.append([torch.randn(s, generator=g) for s in [(7,), (64, 16, 3, 3), (13, 32, 1, 1), (300,)]])
    for rank, scale, grads, wsum in res:
        assert scale == 0.5 and wsum == 0.0
        for gi, g in enumerate(grads):
            np.testing.assert_allclose(g, (expect[0][gi] + expect[1][gi]).numpy(), rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------- YAML config surface (train.py:104-105, 137-143, 182-246)
def _example_cfg(three_level, sup_map=None):
    """A dict shaped like the reference's example-config.yaml (same keys; 9 fine / 4 coarse / 2 super names)."""
    classes = {"coarse_to_fine_map": [[0, 3], [4, 6], [7], [8]],
               "coarse_names": {0: "Flower", 1: "Tree", 2: "Grass", 3: "Mushroom"},
               "fine_names": {i: f"f{i}" for i in range(9)}}
    if three_level:
        classes["super_coarse_to_coarse_map"] = sup_map
        classes["super_coarse_names"] = {0: "Plant", 1: "Fungus"}
    return {"dataset": {"root": "/nowhere"}, "classes": classes, "model": {"pretrained_model": "resnet-18"},
            "training": {"epochs": 50, "batch_size": 8, "lr": 0.001, "device": "cpu", "fine_weight": 0.7, "coarse_weight": 1.0,
                         "super_weight": 1.0, "num_workers": 1, "gpus": [0]},
            "transform": {"resize": [150, 150], "hflip_prob": 0.5},
            "output": {"checkpoint_dir": "./ck", "project_name": "fun"}}


def test_trainer_from_yaml_config_two_and_three_level(tmp_path):
    import yaml
    from seghiero_amd.loss import HieraTripletLoss, RMIHieraTripletLoss
    from seghiero_amd.train_step import SegHieroTrainer, load_config
    # 2-level: no super_coarse_names -> HieraTripletLoss(num_classes=n_fine, hiera_map, hiera_index, loss_weight=fine_weight)
    cfg = _example_cfg(False)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    assert load_config(path) == cfg
    tr = SegHieroTrainer.from_config(str(path))
    assert isinstance(tr.hiera_loss_fn, HieraTripletLoss)
    assert tr.hiera_loss_fn.hiera_map == [0, 0, 0, 0, 1, 1, 1, 2, 3]
    assert tr.hiera_loss_fn.hiera_index == [[0, 4], [4, 7], [7, 8], [8, 9]]
    assert tr.hiera_loss_fn.loss_weight == 0.7 and tr.optimizer.param_groups[0]["lr"] == 0.001
    assert tr.aspp_head.cls_seg.out_channels == 13 and tr.aux_head[0].out_channels == 9
    assert tr.backbone.out_channels == (64, 128, 256, 512)                      # resnet-18 from model.pretrained_model
    assert tr.checkpoint_path(3) == os.path.join("./ck", "fun_epoch_3_best.pth")
    # 3-level: presence of super_coarse_names (train.py:139); a super map that covers every fine id
    cfg3 = _example_cfg(True, [[0, 6], [7, 8]])
    cfg3["training"]["rmi_pool_size"] = cfg3["training"]["rmi_pool_stride"] = 5
    tr3 = SegHieroTrainer.from_config(cfg3)
    fn = tr3.hiera_loss_fn
    assert isinstance(fn, RMIHieraTripletLoss) and (fn.n_fine, fn.n_mid, fn.n_high) == (9, 4, 2)
    assert fn.fine_to_high.tolist() == [0] * 7 + [1, 1] and fn.fine_to_mid.tolist() == [0, 0, 0, 0, 1, 1, 1, 2, 3]
    assert fn.loss_weight_lambda == 0.7 and fn.loss_weight == 1.0 and fn.rmi_pool_size == 5   # train.py:226-233
    assert tr3.aspp_head.cls_seg.out_channels == 15
    # default depth is the reference's hard-coded ResNet-101 when the YAML names no resnet-<N> (not built here: 42 M params)
    # the example file's own super map [[0,2],[3]] leaves fine ids 4..8 uncovered: documented ValueError
    with pytest.raises(ValueError, match="not covered"):
        SegHieroTrainer.from_config(_example_cfg(True, [[0, 2], [3]]))
    bad = _example_cfg(False)
    bad["classes"]["coarse_to_fine_map"] = [[0, 3], [4, 8]]
    with pytest.raises(ValueError):
        SegHieroTrainer.from_config(bad)
