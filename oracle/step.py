"""One training / validation step of the reference loop on CPU (oracle).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates reference ``train.py:155-246`` (model / loss / optimizer construction) and ``train.py:260-320`` (the step):
backbone -> head -> ``F.interpolate`` x0.5 (dead value, kept) and to the label size -> ``HieraTripletLoss`` with
``step = epoch`` -> aux head on C3, x16 ``F.interpolate``, ``nn.CrossEntropyLoss(ignore_index=255)`` ->
``loss = main + 0.4 * aux`` -> backward -> ``SGD(momentum=0.9, weight_decay=1e-4)``.
This is also what ``bench.py`` times as the CPU baseline (kind "port").
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hierarchy, losses, nets


class OracleTrainer:
    def __init__(self, depth=50, n_fine=9, coarse_to_fine_map=((0, 3), (4, 6), (7,), (8,)), lr=0.01, fine_weight=1.0,
                 head_kw=None, super_coarse_to_coarse_map=None, rmi_radius=3):
        cfg_map = [list(x) for x in coarse_to_fine_map]
        sup_map = None if super_coarse_to_coarse_map is None else [list(x) for x in super_coarse_to_coarse_map]
        self.n_fine, self.n_coarse = n_fine, len(cfg_map)
        self.n_super = 0 if sup_map is None else len(sup_map)
        self.backbone = nets.ResNetBackbone(depth, pretrained=False)
        ch = self.backbone.out_channels
        kw = dict(in_channels=ch[3], c1_in_channels=ch[0], c1_channels=48, aspp_channels=512,
                  dilations=(1, 12, 24, 36), num_classes=n_fine + self.n_coarse + self.n_super, proj_dim=256,
                  proj_type="convmlp")
        kw.update(head_kw or {})
        self.aspp_head = nets.DepthwiseSeparableASPPContrastHead(**kw)
        self.aux_head = nets.make_aux_head(ch[2], n_fine)
        if sup_map is None:
            self.hiera_loss_fn = losses.HieraTripletLoss(n_fine, hierarchy.build_fine_to_coarse_map(cfg_map, n_fine).tolist(),
                                                         hierarchy.build_hiera_index(cfg_map), loss_weight=fine_weight)
        else:
            self.hiera_loss_fn = losses.RMIHieraTripletLoss(n_fine, self.n_coarse, self.n_super,
                                                            hierarchy.build_fine_to_coarse_map(cfg_map, n_fine),
                                                            hierarchy.build_fine_to_super_map(sup_map, n_fine), rmi_radius=rmi_radius,
                                                            loss_weight_lambda=fine_weight, loss_weight=1.0)
        self.aux_criterion = nn.CrossEntropyLoss(ignore_index=255)
        self.params = list(self.backbone.parameters()) + list(self.aspp_head.parameters()) + list(self.aux_head.parameters())
        self.optimizer = torch.optim.SGD(self.params, lr=lr, momentum=0.9, weight_decay=1e-4)

    def modules(self):
        return {"backbone": self.backbone, "aspp_head": self.aspp_head, "aux_head": self.aux_head}

    def state_dicts(self):
        return {k: {n: v.detach().clone() for n, v in m.state_dict().items()} for k, m in self.modules().items()}

    def train(self):
        for m in self.modules().values():
            m.train()

    def eval(self):
        for m in self.modules().values():
            m.eval()

    def forward_loss(self, img, fine_mask, epoch):
        c1, c2, c3, c4 = self.backbone(img)
        main_logits, embedding = self.aspp_head([c1, c2, c3, c4])
        H, W = fine_mask.shape[-2:]
        before = F.interpolate(main_logits, scale_factor=0.5, mode="bilinear", align_corners=False)
        after = F.interpolate(main_logits, size=(H, W), mode="bilinear", align_corners=False)
        main_loss = self.hiera_loss_fn(torch.tensor([epoch]), embedding, before[:, :self.n_fine], after, fine_mask)
        aux = F.interpolate(self.aux_head(c3), size=(H, W), mode="bilinear", align_corners=False)
        aux_loss = self.aux_criterion(aux, fine_mask)
        return main_loss + 0.4 * aux_loss, main_loss, aux_loss, after

    def ddp_forward_loss(self, img, fine_mask, epoch, cuts):
        """Statement of the DATA-PARALLEL step with SyncBN (the build's multi-GPU semantics; the reference has no DDP
        launcher -- its only distributed call is the class_count all_gather of hiera_triplet_loss.py:193-198):

        * rank r holds images [cuts[r], cuts[r+1]); BatchNorm statistics are those of the WHOLE batch on every rank (SyncBN), so the
          trunk, head and aux head are one forward over the concatenated batch;
        * every rank evaluates the loss on ITS shard with ITS OWN normalisers -- num_valid of hiera_triplet_loss.py:41-107, the
          all-pixel mean of models/loss/utils.py:20-21, nn.CrossEntropyLoss's valid-pixel mean, the batch mean of
          rmi_hiera_triplet_loss.py:515 -- and its own triplet selection (tree_triplet_loss.py:23-46 over the shard);
        * the triplet term counts only if EVERY rank found triplets (hiera_triplet_loss.py:193-201);
        * gradient averaging over the ranks = the gradient of the MEAN of the per-rank losses.

        -> (mean loss, [per-rank total], [per-rank main], [per-rank aux])."""
        c1, c2, c3, c4 = self.backbone(img)
        main_logits, embedding = self.aspp_head([c1, c2, c3, c4])
        H, W = fine_mask.shape[-2:]
        before = F.interpolate(main_logits, scale_factor=0.5, mode="bilinear", align_corners=False)
        after = F.interpolate(main_logits, size=(H, W), mode="bilinear", align_corners=False)
        aux = F.interpolate(self.aux_head(c3), size=(H, W), mode="bilinear", align_corners=False)
        trip_fn = getattr(self.hiera_loss_fn, "triplet_loss_fn", None) or self.hiera_loss_fn.triplet_loss
        shards = list(zip(cuts[:-1], cuts[1:]))
        with torch.no_grad():
            ready = all(int(trip_fn(embedding[a:b], fine_mask[a:b])[1]) > 0 for a, b in shards)
        mains, auxs = [], []
        for a, b in shards:
            mains.append(self.hiera_loss_fn(torch.tensor([epoch]), embedding[a:b], before[a:b, :self.n_fine], after[a:b], fine_mask[a:b],
                                            ready=ready))
            auxs.append(self.aux_criterion(aux[a:b], fine_mask[a:b]))
        totals = [m + 0.4 * x for m, x in zip(mains, auxs)]
        return sum(totals) / len(totals), totals, mains, auxs

    def train_step(self, img, fine_mask, epoch=0):
        self.optimizer.zero_grad()
        loss, _, _, _ = self.forward_loss(img, fine_mask, epoch)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    @torch.no_grad()
    def eval_step(self, img, fine_mask, epoch=0):
        loss, _, _, after = self.forward_loss(img, fine_mask, epoch)
        correct, valid = losses.pixel_accuracy_counts(after[:, :self.n_fine], fine_mask)
        cm = losses.confusion_matrix(after[:, :self.n_fine], fine_mask, self.n_fine)
        return loss, correct, valid, cm
