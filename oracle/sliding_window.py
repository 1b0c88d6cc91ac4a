"""TEST INFRASTRUCTURE ONLY (imported by tests/ and the checker legs of smoke / bench, never by the product path).

An independent CPU restatement of the window arithmetic of MONAI 1.1.0's `sliding_window_inference(mode="constant")`, which the reference calls
at networks/lightning_monai.py:86-93,187 (`partial(sliding_window_inference, roi_size=..., sw_batch_size=..., overlap=...)`).  MONAI is a
third-party dependency that is absent from /root/reference (requirements.txt:2 pins monai~=1.1.0) and the reference holds no test or
fixture of it: PARITY UNPINNED BY THE REFERENCE.  The algorithm is restated from MONAI's published behaviour:

  * scan interval per axis: the roi itself when image == roi, else int(roi * (1 - overlap)), at least 1;
  * windows per axis: scan positions 0, interval, 2 interval, ... up to and including the FIRST one whose window reaches the end of the
    image; a window that sticks out is moved back so that it ends at the image's edge;
  * windows are visited with the last axis fastest; every window carries weight 1; out = sum of window predictions / number of windows that
    cover the voxel; an image smaller than the roi is zero-padded symmetrically (the odd voxel at the high end) and the result cropped back.

The product (mi-seg_amd/training/inferer.py) derives its grid from a closed form (ceil((size - roi) / interval) + 1 windows, start
min(i * interval, size - roi)); this module deliberately walks the positions one by one instead, so that the two can be compared
(tests/test_host_api.py::test_window_grid_against_the_independent_restatement)."""
import itertools

import torch
import torch.nn.functional as F


def scan_interval(image, roi, overlap):
    if image == roi:
        return roi
    step = int(roi * (1.0 - overlap))
    return step if step > 0 else 1


def axis_starts(image, roi, overlap):
    """window origins along one axis of an image that is at least as large as the roi (smaller images are padded first)"""
    if image < roi:
        raise ValueError("pad the image to the roi first")
    step = scan_interval(image, roi, overlap)
    starts, pos = [], 0
    while True:
        end = pos + roi
        starts.append(pos - max(end - image, 0))      # a window that sticks out is pulled back to end at the edge
        if end >= image:                              # the first scan position whose window reaches the end is the last one
            return starts
        pos += step


def window_origins(image_size, roi_size, overlap):
    """[(d, h, w)] in visiting order (last axis fastest)"""
    return list(itertools.product(*(axis_starts(s, r, overlap) for s, r in zip(image_size, roi_size))))


def symmetric_pad(image_size, roi_size):
    """[(low, high)] zeros added per axis so that the image is no smaller than the roi"""
    out = []
    for s, r in zip(image_size, roi_size):
        miss = max(r - s, 0)
        out.append((miss // 2, miss - miss // 2))
    return out


@torch.no_grad()
def sliding_window_reference(inputs, roi_size, predictor, overlap=0.5):
    """inputs [B, C, D, H, W] on the CPU; predictor(window [1, C, roi...]) -> [1, K, roi...]; one window at a time (the reference's
    sw_batch_size = 1), plain read-modify-write accumulation in visiting order, division by the coverage count"""
    roi = (roi_size,) * 3 if isinstance(roi_size, int) else tuple(roi_size)
    orig = tuple(inputs.shape[2:])
    pad = symmetric_pad(orig, roi)
    if any(lo or hi for lo, hi in pad):
        inputs = F.pad(inputs, [v for lo_hi in reversed(pad) for v in lo_hi])
    size = tuple(inputs.shape[2:])
    total = count = None
    for b in range(inputs.shape[0]):
        for d, h, w in window_origins(size, roi, overlap):
            win = (slice(d, d + roi[0]), slice(h, h + roi[1]), slice(w, w + roi[2]))
            pred = predictor(inputs[(slice(b, b + 1), slice(None)) + win]).float()
            if total is None:
                total = torch.zeros((inputs.shape[0], pred.shape[1]) + size)
                count = torch.zeros(size)
            total[(b, slice(None)) + win] += pred[0]
            if b == 0:
                count[win] += 1.0
    out = total / count
    crop = tuple(slice(lo, lo + s) for (lo, _), s in zip(pad, orig))
    return out[(slice(None), slice(None)) + crop]
