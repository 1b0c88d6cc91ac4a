"""GPU-resident training augmentation: the random tail of the reference's MONAI chain (data/multi_modal.py:50-65) on volumes that stay
in HBM (the deterministic head -- load, orient, resample, ScaleIntensity, SpatialPad -- is what the reference's CacheDataset caches).

    RandCropByPosNegLabeld(pos=1, neg=1, num_samples=patches_training_sample, image_threshold=0)
    RandFlipd x 3 (axes 0, 1, 2, prob randFlipd_prob each), RandRotate90d(prob, max_k=3, axes (0, 1))
    RandScaleIntensityd(factors=0.1, prob), RandShiftIntensityd(offsets=0.1, prob)

The host draws the per-patch parameters (a torch.Generator: reproducible, rank-seedable) and ONE gather kernel per batch
(csrc/training.hip::augment_kernel) writes image and label patches - no intermediate crops, no per-transform passes.  MONAI's own random
stream cannot be reproduced (its RNG consumption order is an implementation detail of a dependency that is not vendored): the
DISTRIBUTIONS are restated, parity unpinned."""
import ctypes as C

import torch

from ..hip import lib as L
from ..hip import ops


class ResidentVolume:
    """one cached sample: image fp32 [C, D, H, W] and label [D, H, W] on the device + its foreground / background voxel lists
    (MONAI map_binary_to_indices: fg = label > 0, bg = label == 0 & image > image_threshold)"""

    def __init__(self, image, label, modality=0, image_threshold=0.0):
        if image.dim() != 4 or label.shape != image.shape[1:]:
            raise ValueError("image [C, D, H, W] and label [D, H, W] expected")
        self.image = image.float().contiguous()
        self.label = label.contiguous()
        self.modality = int(modality)
        flat = self.label.reshape(-1)
        self.fg = torch.nonzero(flat > 0).reshape(-1)
        self.bg = torch.nonzero((flat == 0) & (self.image.amax(0).reshape(-1) > image_threshold)).reshape(-1)


class GpuAugmenter:
    def __init__(self, roi, patches_training_sample=1, randFlipd_prob=0.2, randRotate90d_prob=0.2, randScaleIntensityd_prob=0.1,
                 randShiftIntensityd_prob=0.1, pos=1.0, neg=1.0, seed=0):
        self.roi = (roi,) * 3 if isinstance(roi, int) else tuple(roi)
        self.n = int(patches_training_sample)
        if not 1 <= self.n <= L.AUG_MAX_SAMPLES:
            raise ValueError(f"patches_training_sample must be in 1..{L.AUG_MAX_SAMPLES}")
        self.p_flip, self.p_rot, self.p_scale, self.p_shift = randFlipd_prob, randRotate90d_prob, randScaleIntensityd_prob, randShiftIntensityd_prob
        self.pos_ratio = pos / (pos + neg)
        self.gen = torch.Generator().manual_seed(seed)

    def _u(self):
        return float(torch.rand((), generator=self.gen))

    def draw(self, vol: ResidentVolume):
        """per-patch parameters (host side): list of dicts origin / flip / rot_k / scale / shift (+ centre: the voxel the crop was drawn around,
        before the crop was moved inside the volume - what the pos / neg balance is about)"""
        D, H, W = vol.label.shape
        out = []
        for _ in range(self.n):
            use_fg = (self._u() < self.pos_ratio and vol.fg.numel() > 0) or vol.bg.numel() == 0
            idx = vol.fg if (use_fg and vol.fg.numel() > 0) else vol.bg
            if idx.numel() == 0:
                centre = (D // 2, H // 2, W // 2)
            else:
                flat = int(idx[int(torch.randint(0, idx.numel(), (), generator=self.gen))])
                centre = (flat // (H * W), (flat // W) % H, flat % W)
            # MONAI correct_crop_centers: the crop stays inside the volume
            origin = [min(max(c - r // 2, 0), s - r) for c, r, s in zip(centre, self.roi, (D, H, W))]
            flip = [int(self._u() < self.p_flip) for _ in range(3)]
            rot_k = int(torch.randint(1, 4, (), generator=self.gen)) if self._u() < self.p_rot else 0
            scale = (self._u() * 0.2 - 0.1) if self._u() < self.p_scale else 0.0
            shift = (self._u() * 0.2 - 0.1) if self._u() < self.p_shift else 0.0
            out.append(dict(origin=origin, flip=flip, rot_k=rot_k, scale=scale, shift=shift, centre=centre))
        return out

    def __call__(self, vol: ResidentVolume, params=None):
        """-> batch dict like the reference's collated loader output: image [n, C, r, r, r] fp32, label [n, 1, r, r, r], modality [n]"""
        params = params if params is not None else self.draw(vol)
        if any(s < r for s, r in zip(vol.label.shape, self.roi)):
            raise ValueError("volume smaller than the roi: pad it first (SpatialPadd is part of the cached, deterministic head)")
        n, Cc = len(params), vol.image.shape[0]
        img = torch.empty((n, Cc) + self.roi, dtype=torch.float32, device=vol.image.device)
        lab = torch.empty((n, 1) + self.roi, dtype=vol.label.dtype, device=vol.image.device)
        samples = (L.AugSample * n)()
        for i, p in enumerate(params):
            samples[i] = L.AugSample((C.c_int * 3)(*p["origin"]), (C.c_int * 3)(*p["flip"]), p["rot_k"], p["scale"], p["shift"])
        D, H, W = vol.label.shape
        q = L.Augment(C.sizeof(L.Augment), vol.image.data_ptr(), vol.label.data_ptr(), vol.label.element_size(), Cc, D, H, W, self.roi[0], self.roi[1],
                      self.roi[2], n, img.data_ptr(), lab.data_ptr(), C.cast(samples, C.c_void_p))
        ops._call("miseg_augment_crop", q)
        return {"image": img, "label": lab, "modality": torch.full((n,), vol.modality, dtype=torch.int64)}
