"""Sliding-window inference with MONAI 1.1.0's geometry (reference lightning_monai.py:86-93,187):
interval = int(roi * (1 - overlap)) per axis, num = ceil((size - roi) / interval) + 1, start_i = min(i * interval, size - roi),
constant importance map, out = sum(pred) / count, symmetric zero pad when the image is smaller than the roi.

Unlike the reference (sw_batch_size must stay 1 with instance_cond because the one-element `modalities` is passed unchanged to every
window batch), windows of one volume are batched with the volume's modality broadcast.

On a HIP device the stitching is sized for 288 GB of HBM: every window's logits stay resident (700 windows x 6 classes x 96^3 fp32 =
14.9 GB for the 512 x 512 x 363 volume of BASELINE configs[4]) and ONE gather kernel (csrc/training.hip::stitch_kernel) writes
out = sum / count with the windows summed in window-index order -- MONAI's accumulation order, so the result is bit-identical to the
sequential `out[window] += pred` loop, without its 700 read-modify-write passes over the 2.3 GB accumulator and without atomics.
With `device=cpu` (the reference's --infer_cpu: logits stitched in host memory) or CPU inputs the plain loop below runs."""
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


RESIDENT_LIMIT_BYTES = 200e9     # window logits kept in HBM for the gather (288 GB card)


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
    starts = tuple(_starts(s, r, overlap) for s, r in zip(size, roi))
    grid = [(d, h, w) for d in starts[0] for h in starts[1] for w in starts[2]]
    mods = None
    if modalities is not None:
        mods = [int(m) for m in (modalities.reshape(-1).tolist() if isinstance(modalities, torch.Tensor) else modalities)]
    on_hip = inputs.is_cuda and (device is None or torch.device(device).type == "cuda")
    out = cnt = win = None
    for b in range(B):
        for i in range(0, len(grid), sw_batch_size):
            chunk = grid[i:i + sw_batch_size]
            x = torch.cat([inputs[b:b + 1, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]] for (d, h, w) in chunk], 0)
            pred = predictor(x, [mods[b]] * len(chunk), **kwargs) if mods is not None else predictor(x, **kwargs)
            if out is None:
                dev = pred.device if on_hip else (device or pred.device)
                out = torch.empty((B, pred.shape[1]) + size, dtype=torch.float32, device=dev) if on_hip else \
                    torch.zeros((B, pred.shape[1]) + size, dtype=torch.float32, device=dev)
                if on_hip:
                    need = len(grid) * pred.shape[1] * roi[0] * roi[1] * roi[2] * 4
                    if need > RESIDENT_LIMIT_BYTES:
                        raise NotImplementedError(f"{need / 1e9:.0f} GB of window logits exceed the resident budget; stitch the volume in slabs")
                    win = torch.empty((len(grid), pred.shape[1]) + roi, dtype=torch.float32, device=dev)
                else:
                    cnt = torch.zeros((1, 1) + size, dtype=torch.float32, device=dev)
            if on_hip:
                win[i:i + len(chunk)].copy_(pred)
                continue
            for j, (d, h, w) in enumerate(chunk):
                out[b, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += pred[j].to(out.device, torch.float32)
                if b == 0:
                    cnt[0, 0, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += 1.0
        if on_hip:
            from ..hip import ops
            ops.stitch_windows(win, out[b], starts, roi)
    if not on_hip:
        out = out / cnt
    if any(pads):
        sl = [slice(None), slice(None)] + [slice(p // 2, p // 2 + s) for p, s in zip(pads, orig)]
        out = out[tuple(sl)]
    return out
