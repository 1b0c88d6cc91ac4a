"""Synthetic CT / MR volumes with labels (there is no dataset in this image: SURVEY.md 8(d) "Synthetic inputs").

A volume is smooth noise in [0, 1] (what ScaleIntensityd yields, reference data/multi_modal.py:46) plus `n_classes - 1` ellipsoids, one per
foreground class, each brighter than the background by a class-dependent step; "MR" volumes (modality 1) go through a different
intensity transfer curve.  Deterministic in (seed, shape, modality)."""
import torch


def synthetic_volume(shape, seed, modality=0, n_classes=6):
    """-> image fp32 [1, 1, D, H, W] in [0, 1], label int64 [1, 1, D, H, W] in [0, n_classes)"""
    g = torch.Generator().manual_seed(seed)
    D, H, W = shape
    coarse = tuple(max(2, s // 8) for s in shape)
    v = torch.nn.functional.interpolate(torch.rand(1, 1, *coarse, generator=g), size=shape, mode="trilinear", align_corners=False)[0, 0] * 0.35
    zz, yy, xx = torch.meshgrid(torch.linspace(-1, 1, D), torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    label = torch.zeros(shape, dtype=torch.int64)
    for c in range(1, n_classes):
        ctr = torch.rand(3, generator=g) * 1.2 - 0.6
        rad = torch.rand(3, generator=g) * 0.25 + 0.2
        inside = (((zz - ctr[0]) / rad[0]) ** 2 + ((yy - ctr[1]) / rad[1]) ** 2 + ((xx - ctr[2]) / rad[2]) ** 2) < 1
        label[inside] = c
        v = torch.where(inside, torch.full_like(v, 0.15 + 0.12 * c) + 0.2 * v, v)
    v = (v - v.min()) / (v.max() - v.min())
    if modality == 1:
        v = v ** 0.6
    return v[None, None].contiguous(), label[None, None].contiguous()
