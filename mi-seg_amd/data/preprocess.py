"""The cached, deterministic head of the reference's transform chain on the device (data/multi_modal.py:37-49, what its CacheDataset
caches): LoadImaged + EnsureChannelFirstd + Orientationd (data/nifti.py, host) -> Spacingd (image trilinear, label nearest) ->
ScaleIntensityd -> SpatialPadd.  MONAI's own resampling grid and rounding rules are not restated (parity unpinned):
the output size is round(size * pixdim_in / pixdim_out) per axis with voxel centres aligned."""
import ctypes as C

import numpy as np
import torch

from ..hip import lib as L
from ..hip import ops
from .augment import ResidentVolume
from .decathlon import modality_id
from .nifti import read_nifti, reorient_to_ras


def resample(vol, out_size, mode="trilinear"):
    """vol [C, D, H, W] on the device -> [C, *out_size]; mode 'trilinear' (fp32) | 'nearest' (any 1 / 4 / 8-byte dtype)"""
    if vol.dim() != 4 or not vol.is_cuda:
        raise ValueError("resample: device tensor [C, D, H, W] expected")
    vol = vol.contiguous()
    if mode == "trilinear":
        vol = vol.float()
    elif mode != "nearest":
        raise ValueError(f"resample: mode '{mode}'")
    out = torch.empty((vol.shape[0],) + tuple(int(s) for s in out_size), dtype=vol.dtype, device=vol.device)
    p = L.Resample3d(C.sizeof(L.Resample3d), vol.data_ptr(), out.data_ptr(), vol.shape[0], vol.shape[1], vol.shape[2], vol.shape[3], out.shape[1],
                     out.shape[2], out.shape[3], 0 if mode == "trilinear" else 1, vol.element_size())
    ops._call("miseg_resample3d", p)
    return out


def spacing(vol, pixdim_in, pixdim_out, mode):
    size = [max(1, int(round(s * pi / po))) for s, pi, po in zip(vol.shape[1:], pixdim_in, pixdim_out)]
    return resample(vol, size, mode)


def scale_intensity(img):
    """ScaleIntensityd defaults: (x - min) / (max - min) -> [0, 1] (a constant image maps to 0)"""
    lo, hi = img.amin(), img.amax()
    return (img - lo) / (hi - lo).clamp_min(1e-30)


def spatial_pad(vol, roi, value=0):
    """SpatialPadd(method="symmetric"): pad each axis up to the roi, half before / half after"""
    pads = []
    for s, r in zip(reversed(vol.shape[1:]), reversed(roi)):
        t = max(r - s, 0)
        pads += [t // 2, t - t // 2]
    return torch.nn.functional.pad(vol, pads, value=value) if any(pads) else vol


def load_resident_volume(item, pixdim=(1.0, 1.0, 1.0), roi=(96, 96, 96), device="cuda"):
    """one data-list item ({'image', 'label', 'modality'}: data/decathlon.py) -> data/augment.py::ResidentVolume, ready for the GPU
    augmenter / the sliding-window inferer"""
    img, aff = read_nifti(item["image"])
    lab, laff = read_nifti(item["label"])
    img, aff = reorient_to_ras(img, aff)
    lab, _ = reorient_to_ras(lab, laff)
    vox = np.sqrt((aff[:3, :3] ** 2).sum(0))
    image = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))[None].to(device)
    label = torch.from_numpy(np.ascontiguousarray(lab).astype(np.int64))[None].to(device)
    image = spatial_pad(scale_intensity(spacing(image, vox, pixdim, "trilinear")), roi)
    label = spatial_pad(spacing(label, vox, pixdim, "nearest"), roi)
    return ResidentVolume(image, label[0], modality=modality_id(item.get("modality", 0)))
