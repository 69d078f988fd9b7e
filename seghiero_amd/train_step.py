"""The training / validation step of reference ``train.py:260-320`` and ``:327-393`` on the HIP path.

``SegHieroTrainer`` owns what ``train.py:155-246`` builds (backbone, DS-ASPP head, aux head, loss, SGD) and runs one
step exactly as the reference loop does -- forward, main loss on the logits resized to the label grid, aux head on C3
with its x16 resize + ``nn.CrossEntropyLoss(ignore_index=255)``, ``loss = main + 0.4 * aux``, backward, SGD -- with
these fusions (results unchanged):

* the two full-resolution logit tensors of ``train.py:282-284, 310-312`` are never materialised: the bilinear resize is
  evaluated inside the loss kernels and inside their gather-form backward;
* ``logit_before_full`` (``train.py:277-279``) is not computed: neither loss reads it;
* ``loss.item()`` (``train.py:319``) is not called: the step returns device scalars, the caller decides when to sync.

``step`` passed to the loss is the EPOCH, as in the reference (``train.py:287``).
"""
import os
import re

import torch
import torch.nn as nn

from . import ops
from .backbone import ResNetBackbone
from .head import AuxHead, DepthwiseSeparableASPPContrastHead
from .hierarchy import build_fine_to_coarse_map, build_fine_to_super_map, build_hiera_index
from .loss import HieraTripletLoss, RMIHieraTripletLoss
from .sgd import FusedSGD


class _AuxCEFn(torch.autograd.Function):
    """x16 bilinear resize + nn.CrossEntropyLoss(ignore_index=255) (mean over valid pixels), train.py:309-313."""

    @staticmethod
    def forward(ctx, logits, label8):
        lg = ops.to_nhwc(logits)
        from . import ddp
        loss, sums, gw = ops.ce_fwd(lg, label8, want_grad=ctx.needs_input_grad[0], norm=ddp.current_counts())
        ctx.save_for_backward(lg, label8, sums)
        ctx.grad_ws = gw
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lg, label8, sums = ctx.saved_tensors
        d = ops.ce_bwd(lg, label8, sums, g.reshape(1).float(), 1.0, grad_ws=ctx.grad_ws)
        ctx.grad_ws = None
        return d, None


def aux_ce_loss(aux_logits, label):
    return _AuxCEFn.apply(aux_logits, ops.labels_u8(label))


def load_config(path):
    """``yaml.safe_load`` of a reference-format config file (``train.py:104-105``)."""
    import yaml
    with open(path, "r") as f:
        return yaml.safe_load(f)


class SegHieroTrainer:
    @classmethod
    def from_config(cls, cfg, depth=None, device=None, grad_sync=None, head_kw=None):
        """Build the trainer from the reference's YAML config surface -- the dict ``yaml.safe_load`` returns, or a path.

        Consumes exactly the keys ``train.py`` reads for the model / loss / optimizer (``:137-143, 182, 199, 203-233, 243``):
        ``classes.{fine_names, coarse_names, super_coarse_names, coarse_to_fine_map, super_coarse_to_coarse_map}``,
        ``training.{lr, fine_weight, rmi_radius, rmi_pool_way, rmi_pool_size, rmi_pool_stride, device}``; 2- vs 3-level
        by the presence of ``classes.super_coarse_names`` (``:139``).  ``training.{batch_size, epochs}`` and ``output.*``
        are kept on the instance (``cfg``) for the caller's loop and ``checkpoint_path``.
        The reference hard-codes ResNet-101 (``:155``) and ignores ``model.pretrained_model``; here ``depth`` (argument)
        wins, else a ``resnet-<N>`` value of ``model.pretrained_model`` is honoured, else 101 -- the example config's
        ``resnet-101`` gives the reference's model."""
        if isinstance(cfg, (str, os.PathLike)):
            cfg = load_config(cfg)
        classes, training = cfg["classes"], cfg["training"]
        n_fine, n_coarse = len(classes["fine_names"]), len(classes["coarse_names"])
        has_super = "super_coarse_names" in classes
        if len(classes["coarse_to_fine_map"]) != n_coarse:
            raise ValueError("classes.coarse_to_fine_map must have one entry per coarse name")
        if has_super and len(classes["super_coarse_to_coarse_map"]) != len(classes["super_coarse_names"]):
            raise ValueError("classes.super_coarse_to_coarse_map must have one entry per super-coarse name")
        if depth is None:
            m = re.fullmatch(r"resnet-?(\d+)", str(cfg.get("model", {}).get("pretrained_model", "")).strip().lower())
            depth = int(m.group(1)) if m else 101
        if device is None:
            device = training.get("device", "cuda")
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        rmi = {k: training.get(k, d) for k, d in (("rmi_radius", 3), ("rmi_pool_way", 0), ("rmi_pool_size", 3),
                                                  ("rmi_pool_stride", 3))}
        tr = cls(depth=depth, n_fine=n_fine, coarse_to_fine_map=classes["coarse_to_fine_map"], lr=training["lr"],
                 fine_weight=training.get("fine_weight", 1.0), device=device, head_kw=head_kw, grad_sync=grad_sync,
                 super_coarse_to_coarse_map=classes["super_coarse_to_coarse_map"] if has_super else None, **rmi)
        tr.cfg = cfg
        return tr

    def checkpoint_path(self, epoch):
        """``<output.checkpoint_dir>/<output.project_name>_epoch_<epoch>_best.pth`` (``train.py:430-433``)."""
        if self.cfg is None:
            raise ValueError("checkpoint_path() needs the YAML config: build the trainer with SegHieroTrainer.from_config(cfg)")
        out = self.cfg["output"]
        return os.path.join(out["checkpoint_dir"], f"{out['project_name']}_epoch_{epoch}_best.pth")

    def __init__(self, depth=50, n_fine=9, coarse_to_fine_map=((0, 3), (4, 6), (7,), (8,)), lr=0.01, fine_weight=1.0,
                 device="cuda:0", head_kw=None, grad_sync=None, super_coarse_to_coarse_map=None, rmi_radius=3,
                 rmi_pool_way=0, rmi_pool_size=3, rmi_pool_stride=3, act_dtype=torch.float32, compute_dtype=torch.float32):
        """``super_coarse_to_coarse_map`` given -> 3-level model + RMIHieraTripletLoss (train.py:202-233), else the
        2-level HieraTripletLoss (train.py:176-200).  ``act_dtype=torch.bfloat16``: the trunk stores its activations as bf16
        (BASELINE configs[4]; ResNetBackbone.act_dtype)."""
        cfg_map = [list(x) for x in coarse_to_fine_map]
        sup_map = None if super_coarse_to_coarse_map is None else [list(x) for x in super_coarse_to_coarse_map]
        self.n_fine, self.n_coarse = n_fine, len(cfg_map)
        self.n_super = 0 if sup_map is None else len(sup_map)
        self.device = torch.device(device)
        self.backbone = ResNetBackbone(depth=depth, pretrained=False)
        # compute_dtype = torch.bfloat16 (BASELINE configs[4] "bf16"): bf16 compute mode -- trunk and decoder store bf16 activations and
        # their convolutions issue ONE bf16 MFMA product per tile on operands rounded once to bf16 (fp32 accumulate, fp32 BatchNorm
        # statistics, fp32 weights / weight gradients / SGD); the default (float32) is the fp32-accurate six-product plan
        self.compute_dtype = compute_dtype
        if compute_dtype == torch.bfloat16:
            act_dtype = torch.bfloat16
        self.backbone.act_dtype = act_dtype
        self.backbone.compute_dtype = compute_dtype
        ch = self.backbone.out_channels
        kw = dict(in_channels=ch[3], c1_in_channels=ch[0], c1_channels=48, aspp_channels=512,
                  dilations=(1, 12, 24, 36), num_classes=n_fine + self.n_coarse + self.n_super, proj_dim=256,
                  proj_type="convmlp")
        kw.update(head_kw or {})
        self.aspp_head = DepthwiseSeparableASPPContrastHead(**kw)
        # (the head's decoder can store bf16 too -- aspp_head.act_dtype -- but its depthwise kernels then move 128-byte segments and run
        # slower than they save: measured 49.0 vs 48.3 ms per step at the configs[4] shape, 8.9 vs 10.3 GiB; left to the caller)
        if compute_dtype == torch.bfloat16:
            self.aspp_head.act_dtype = torch.bfloat16
            self.aspp_head.compute_dtype = torch.bfloat16
        self.aux_head = AuxHead(ch[2], n_fine)
        if sup_map is None:
            self.hiera_loss_fn = HieraTripletLoss(num_classes=n_fine, hiera_map=build_fine_to_coarse_map(cfg_map, n_fine).tolist(),
                                                  hiera_index=build_hiera_index(cfg_map), loss_weight=fine_weight)
        else:   # train.py:226-233: loss_weight_lambda = training.fine_weight, loss_weight = 1.0
            self.hiera_loss_fn = RMIHieraTripletLoss(n_fine, self.n_coarse, self.n_super, build_fine_to_coarse_map(cfg_map, n_fine),
                                                     build_fine_to_super_map(sup_map, n_fine), rmi_radius=rmi_radius,
                                                     rmi_pool_way=rmi_pool_way, rmi_pool_size=rmi_pool_size,
                                                     rmi_pool_stride=rmi_pool_stride,
                                                     loss_weight_lambda=fine_weight, loss_weight=1.0)
        for m in (self.backbone, self.aspp_head, self.aux_head, self.hiera_loss_fn):
            m.to(self.device)
        self.params = list(self.backbone.parameters()) + list(self.aspp_head.parameters()) + list(self.aux_head.parameters())
        # dense conv weights whose backward runs a dgrad (everything but the depthwise convs and the stem)
        self._wt_cache = {}
        self._wb_cache = {}
        self._dgrad_weights = [m.weight for mod in (self.backbone, self.aspp_head, self.aux_head) for m in mod.modules()
                               if isinstance(m, nn.Conv2d) and m.groups == 1 and m is not self.backbone.stem_conv]
        self.optimizer = FusedSGD(self.params, lr=lr, momentum=0.9, weight_decay=1e-4)
        self.grad_sync = grad_sync
        self.cfg = None
        self._step_scope = {}                    # buffers that live from the forward to the backward of one step (ops.STEP_SCOPE)

    def modules(self):
        return {"backbone": self.backbone, "aspp_head": self.aspp_head, "aux_head": self.aux_head}

    def load_state_dicts(self, sd):
        for k, m in self.modules().items():
            m.load_state_dict(sd[k])

    def save_checkpoint(self, path, epoch, config=None):
        """Write the dict ``train.py:421-428`` writes (same keys; state_dicts in the reference's NCHW shapes, optimizer state
        in ``torch.optim.SGD``'s format), so reference tooling -- ``infer.py:277-279``, a resumed ``train.py`` -- can load it."""
        ckpt = {"epoch": int(epoch),
                "backbone_state_dict": self.backbone.state_dict(),
                "aspp_head_state_dict": self.aspp_head.state_dict(),
                "aux_head_state_dict": self.aux_head.state_dict(),
                "optimizer_state_dict": self.optimizer.state_dict(),
                "config": config if config is not None else {}}
        torch.save(ckpt, path)

    def load_checkpoint(self, path, load_optimizer=True):
        """Load a checkpoint written by the reference (``train.py:419-435``) or by ``save_checkpoint``.  Only tensors and plain
        containers are read (``weights_only=True``); returns the stored epoch."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        self.backbone.load_state_dict(ckpt["backbone_state_dict"])
        self.aspp_head.load_state_dict(ckpt["aspp_head_state_dict"])
        self.aux_head.load_state_dict(ckpt["aux_head_state_dict"])
        if load_optimizer and ckpt.get("optimizer_state_dict") is not None:
            self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
            self.optimizer.adopt_layouts()
        return int(ckpt.get("epoch", 0))

    def train(self):
        for m in self.modules().values():
            m.train()

    def eval(self):
        for m in self.modules().values():
            m.eval()

    def forward_loss(self, img, fine_mask, epoch):
        c1, c2, c3, c4 = self.backbone(img)
        main_logits, embedding = self.aspp_head([c1, c2, c3, c4])
        main_loss = self.hiera_loss_fn(epoch, embedding, None, main_logits, fine_mask)
        aux_loss = aux_ce_loss(self.aux_head(c3), fine_mask)
        return main_loss + 0.4 * aux_loss, main_loss, aux_loss, main_logits

    def train_step(self, img, fine_mask, epoch=0):
        """One iteration of train.py:260-320.  Returns the (device) loss scalar; nothing is synchronised."""
        self.optimizer.zero_grad(set_to_none=True)
        if ops.CONV_IMPL == "x6":
            ops.prepare_dgrad_weights(self._dgrad_weights, self._wt_cache)   # all dgrad operands in 2 launches; valid until SGD
            if self.compute_dtype == torch.bfloat16:
                ops.prepare_bf16_weights(self._dgrad_weights, self._wb_cache)    # bf16 operand copies of the fp32 master weights
        ops.STEP_SCOPE = self._step_scope            # the loss forward's per-pixel gradient buffers are reused from step to step
        try:
            if self.grad_sync is not None and self.n_super == 0:
                from . import ddp
                ddp.exact_counts(ops.labels_u8(fine_mask), self.n_fine, self.hiera_loss_fn.hiera_index)      # (no-op unless ddp.EXACT)
            loss, _, _, _ = self.forward_loss(img, fine_mask, epoch)
            if self.grad_sync is not None:
                self.grad_sync.begin()                   # buckets are exchanged as the backward nodes finish them
            loss.backward()
        finally:
            ops.release_dgrad_weights()
            ops.STEP_SCOPE = None
            if self.grad_sync is not None:
                from . import ddp
                ddp.end_step()
        gscale = 1.0
        if self.grad_sync is not None:
            gscale = self.grad_sync.reduce(self.params)
        self.optimizer.step(grad_scale=gscale)
        return loss.detach()

    @torch.no_grad()
    def eval_step(self, img, fine_mask, epoch=0, counts=None):
        """One iteration of train.py:341-393: loss + fine pixel-accuracy counts (+ confusion matrix for mIoU)."""
        loss, _, _, logits = self.forward_loss(img, fine_mask, epoch)
        counts = ops.pixel_metrics(ops.to_nhwc(logits), ops.labels_u8(fine_mask), self.n_fine, counts)
        return loss, counts
