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
