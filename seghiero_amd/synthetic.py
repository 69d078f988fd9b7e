"""Synthetic inputs of the benchmark configurations: seeded CPU generator, ``randn`` images (already
"normalised"), blocky label maps (``randint`` at 1/16 resolution blown up x16, so every coarse bucket has anchors,
positives and negatives), 5 % of the pixels and an 8-pixel border set to 255."""
import torch

CONFIGS = {
    # name: depth, fine names, coarse_to_fine_map, image size, batch
    "C1": dict(depth=18, n_fine=4, coarse_to_fine_map=[[0, 1], [2, 3]], size=256, batch=2, images=8),
    "C2": dict(depth=50, n_fine=9, coarse_to_fine_map=[[0, 3], [4, 6], [7], [8]], size=512, batch=16, images=16),
}


def make_batch(batch, size, n_fine, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    h = w = size
    img = torch.randn(batch, 3, h, w, generator=g)
    small = torch.randint(0, n_fine, (batch, -(-h // 16), -(-w // 16)), generator=g)
    lab = small.repeat_interleave(16, 1).repeat_interleave(16, 2)[:, :h, :w].clone()
    lab[torch.rand(batch, h, w, generator=g) < 0.05] = 255
    lab[:, :8] = 255
    lab[:, -8:] = 255
    lab[:, :, :8] = 255
    lab[:, :, -8:] = 255
    return img.to(device), lab.long().to(device)
