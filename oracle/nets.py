"""Backbone + DS-ASPP contrast head (oracle).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Pure ``torch.nn`` CPU restatement with the reference's attribute names, so state_dicts are
interchangeable with the reference and with ``seghiero_amd``:

* ``ResNetBackbone``  <- reference ``models/backbone/resnet.py:26-75`` which wraps
  ``torchvision.models.resnet{50,101}`` (third-party, un-pinned in ``requirements.txt:3``, not
  installed here).  The trunk below is the public torchvision architecture (7x7/2 stem, 3x3/2
  max-pool, Bottleneck v1.5 with the stride on the 3x3, BasicBlock for 18/34, kaiming-normal
  fan_out init, BN gamma=1 beta=0).  **Parity unpinned** for this part (no reference fixture can
  exist); checked only through parameter counts and key shapes.
* ``ProjectionHead``, ``DepthwiseSeparableConv``, ``DepthwiseSeparableASPPModule``,
  ``DepthwiseSeparableASPPContrastHead``  <- reference
  ``models/head/sep_aspp_contrast_head.py:6-30, 33-62, 65-132, 135-254``; pinned by golden G3.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_RESNET_SPECS = {
    18: ("basic", (2, 2, 2, 2)),
    34: ("basic", (3, 4, 6, 3)),
    50: ("bottleneck", (3, 4, 6, 3)),
    101: ("bottleneck", (3, 4, 23, 3)),
    152: ("bottleneck", (3, 8, 36, 3)),
}


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, width, stride, downsample):
        super().__init__()
        self.conv1 = _conv(cin, width, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(width, width, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(width)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride, downsample):
        super().__init__()
        self.conv1 = _conv(cin, width, 1)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _conv(width, width, 3, stride, 1)     # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = _conv(width, width * 4, 1)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


def _make_stage(block, cin, width, n, stride):
    cout = width * block.expansion
    ds = None
    if stride != 1 or cin != cout:
        ds = nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))
    blocks = [block(cin, width, stride, ds)]
    blocks += [block(cout, width, 1, None) for _ in range(n - 1)]
    return nn.Sequential(*blocks), cout


class ResNetBackbone(nn.Module):
    """``ResNetBackbone(depth, pretrained)`` -> ``(c1, c2, c3, c4)`` at strides 4/8/16/32.

    The reference accepts only 50/101 (``resnet.py:34-39``); 18/34/152 are a superset needed
    by BASELINE config 1.  ``pretrained=True`` is a network fetch in the reference
    (``resnet.py:35,37``) -- unavailable offline, so it is accepted and ignored (random init).
    """

    def __init__(self, depth: int = 101, pretrained: bool = True):
        super().__init__()
        if depth not in _RESNET_SPECS:
            raise ValueError("`depth` must be one of 18, 34, 50, 101, 152")
        kind, counts = _RESNET_SPECS[depth]
        block = BasicBlock if kind == "basic" else Bottleneck
        self.stem_conv = _conv(3, 64, 7, 2, 3)
        self.stem_bn = nn.BatchNorm2d(64)
        self.stem_relu = nn.ReLU(inplace=True)
        self.stem_pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        c = 64
        self.layer1, c = _make_stage(block, c, 64, counts[0], 1)
        self.layer2, c = _make_stage(block, c, 128, counts[1], 2)
        self.layer3, c = _make_stage(block, c, 256, counts[2], 2)
        self.layer4, c = _make_stage(block, c, 512, counts[3], 2)
        self.out_channels = (64 * block.expansion, 128 * block.expansion,
                             256 * block.expansion, 512 * block.expansion)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.stem_pool(self.stem_relu(self.stem_bn(self.stem_conv(x))))
        c1 = self.layer1(x)
        c2 = self.layer2(c1)
        c3 = self.layer3(c2)
        c4 = self.layer4(c3)
        return c1, c2, c3, c4


class ProjectionHead(nn.Module):
    def __init__(self, dim_in, proj_dim=256, proj="convmlp"):
        super().__init__()
        if proj == "linear":
            self.proj = _conv(dim_in, proj_dim, 1)
        elif proj == "convmlp":
            self.proj = nn.Sequential(_conv(dim_in, dim_in, 1), nn.BatchNorm2d(dim_in),
                                      nn.ReLU(inplace=True), _conv(dim_in, proj_dim, 1))
        else:
            raise ValueError(f"Unknown proj type: {proj}")

    def forward(self, x):
        return F.normalize(self.proj(x), p=2, dim=1)


class DepthwiseSeparableConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, dilation=1, padding=1, bias=False):
        super().__init__()
        self.depthwise = nn.Conv2d(in_channels, in_channels, kernel_size, padding=padding,
                                   dilation=dilation, groups=in_channels, bias=bias)
        self.bn_dw = nn.BatchNorm2d(in_channels)
        self.act_dw = nn.ReLU(inplace=True)
        self.pointwise = nn.Conv2d(in_channels, out_channels, 1, bias=bias)
        self.bn_pw = nn.BatchNorm2d(out_channels)
        self.act_pw = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.act_dw(self.bn_dw(self.depthwise(x)))
        return self.act_pw(self.bn_pw(self.pointwise(x)))


def _cbr(cin, cout):
    return nn.Sequential(_conv(cin, cout, 1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DepthwiseSeparableASPPModule(nn.Module):
    """Image-pool branch + 1x1 branch + DS dilated branches, concatenated in the order
    [imgpool, br0, br1, ...] (reference ``sep_aspp_contrast_head.py:100-114``).

    The reference first builds dense dilated 3x3 branches and then replaces them
    (``:84-90, 125-131``), which consumes RNG draws; the same throw-away construction is
    done here so that seed-for-seed default initialisation matches.
    """

    def __init__(self, dilations, in_channels, channels):
        super().__init__()
        self.dilations = dilations
        self.branches = nn.ModuleList([_cbr(in_channels, channels)])
        for d in dilations[1:]:
            nn.Conv2d(in_channels, channels, 3, padding=d, dilation=d, bias=False)  # discarded draw
            self.branches.append(None)
        self.image_pool = nn.AdaptiveAvgPool2d(1)
        self.image_pool_conv = _cbr(in_channels, channels)
        for i, d in enumerate(dilations[1:], start=1):
            self.branches[i] = nn.Sequential(
                DepthwiseSeparableConv(in_channels, channels, 3, dilation=d, padding=d))

    def forward(self, x):
        h, w = x.shape[2:]
        pooled = self.image_pool_conv(self.image_pool(x))
        outs = [F.interpolate(pooled, size=(h, w), mode="bilinear", align_corners=False)]
        outs += [br(x) for br in self.branches]
        return torch.cat(outs, dim=1)


class DepthwiseSeparableASPPContrastHead(nn.Module):
    def __init__(self, in_channels, c1_in_channels, c1_channels, aspp_channels, dilations,
                 num_classes, proj_dim=256, proj_type="convmlp"):
        super().__init__()
        self.proj_head = ProjectionHead(in_channels, proj_dim, proj_type)
        self.register_buffer("step", torch.zeros(1, dtype=torch.long))
        self.aspp = DepthwiseSeparableASPPModule(dilations, in_channels, aspp_channels)
        self.bottleneck = _cbr(aspp_channels * (len(dilations) + 1), aspp_channels)
        if c1_in_channels > 0:
            self.c1_bottleneck = _cbr(c1_in_channels, c1_channels)
        else:
            self.c1_bottleneck = None
            c1_channels = 0
        self.sep_bottleneck = nn.Sequential(
            DepthwiseSeparableConv(aspp_channels + c1_channels, aspp_channels, 3, padding=1),
            DepthwiseSeparableConv(aspp_channels, aspp_channels, 3, padding=1))
        self.cls_seg = nn.Conv2d(aspp_channels, num_classes, 1)
        self.align_corners = False

    def forward(self, inputs):
        self.step += 1
        c4 = inputs[-1]
        embedding = self.proj_head(c4)
        x = self.bottleneck(self.aspp(c4))
        if self.c1_bottleneck is not None:
            c1 = self.c1_bottleneck(inputs[0])
            x = F.interpolate(x, size=c1.shape[2:], mode="bilinear", align_corners=False)
            x = torch.cat([x, c1], dim=1)
        x = self.sep_bottleneck(x)
        return self.cls_seg(x), embedding


def make_aux_head(c3_channels, n_fine):
    """Aux head of reference ``train.py:169-173``: Conv1x1(no bias)+BN+ReLU on C3."""
    return nn.Sequential(_conv(c3_channels, n_fine, 1), nn.BatchNorm2d(n_fine), nn.ReLU(inplace=True))
