"""Sliding-window inference with MONAI 1.1.0's geometry (reference lightning_monai.py:86-93,187):
interval = int(roi * (1 - overlap)) per axis, num = ceil((size - roi) / interval) + 1, start_i = min(i * interval, size - roi),
constant importance map, out = sum(pred) / count, symmetric zero pad when the image is smaller than the roi.

Unlike the reference (sw_batch_size must stay 1 with instance_cond because the one-element `modalities` is passed unchanged to every
window batch), windows of one volume are batched with the volume's modality broadcast.  Stitching is done on the device."""
import math

import torch
import torch.nn.functional as F


def _starts(size, roi, overlap):
    if size <= roi:
        return [0]
    interval = int(roi * (1 - overlap))
    interval = interval if interval > 0 else 1
    num = int(math.ceil((size - roi) / interval)) + 1
    return [min(i * interval, size - roi) for i in range(num)]


def window_grid(image_size, roi_size, overlap):
    sd, sh, sw = (_starts(s, r, overlap) for s, r in zip(image_size, roi_size))
    return [(d, h, w) for d in sd for h in sh for w in sw]


@torch.no_grad()
def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.5, modalities=None, device=None, **kwargs):
    """inputs [B, C, D, H, W]; returns [B, out, D, H, W] (same stitching arithmetic as MONAI's mode="constant")."""
    roi = (roi_size,) * 3 if isinstance(roi_size, int) else tuple(roi_size)
    B = inputs.shape[0]
    orig = tuple(inputs.shape[2:])
    pads = [max(r - s, 0) for r, s in zip(roi, orig)]
    if any(pads):
        pp = []
        for p in reversed(pads):
            pp += [p // 2, p - p // 2]
        inputs = F.pad(inputs, pp)
    size = tuple(inputs.shape[2:])
    grid = window_grid(size, roi, overlap)
    out = cnt = None
    mods = None
    if modalities is not None:
        mods = [int(m) for m in (modalities.reshape(-1).tolist() if isinstance(modalities, torch.Tensor) else modalities)]
    for b in range(B):
        for i in range(0, len(grid), sw_batch_size):
            chunk = grid[i:i + sw_batch_size]
            win = torch.cat([inputs[b:b + 1, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]] for (d, h, w) in chunk], 0)
            pred = predictor(win, [mods[b]] * len(chunk), **kwargs) if mods is not None else predictor(win, **kwargs)
            if out is None:
                dev = device or pred.device
                out = torch.zeros((B, pred.shape[1]) + size, dtype=torch.float32, device=dev)
                cnt = torch.zeros((1, 1) + size, dtype=torch.float32, device=dev)
            for j, (d, h, w) in enumerate(chunk):
                out[b, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += pred[j].to(out.device, torch.float32)
                if b == 0:
                    cnt[0, 0, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += 1.0
    out = out / cnt
    if any(pads):
        sl = [slice(None), slice(None)] + [slice(p // 2, p // 2 + s) for p, s in zip(pads, orig)]
        out = out[tuple(sl)]
    return out
