"""Sliding-window inference with MONAI 1.1.0's geometry (reference lightning_monai.py:86-93,187):
interval = int(roi * (1 - overlap)) per axis, num = ceil((size - roi) / interval) + 1, start_i = min(i * interval, size - roi),
constant importance map, out = sum(pred) / count, symmetric zero pad when the image is smaller than the roi.

Unlike the reference (sw_batch_size must stay 1 with instance_cond because the one-element `modalities` is passed unchanged to every
window batch), windows of one volume are batched with the volume's modality broadcast.

On a HIP device the stitching is sized for 288 GB of HBM: every window's logits stay resident (700 windows x 6 classes x 96^3 fp32 =
14.9 GB for the 512 x 512 x 363 volume of BASELINE configs[4]) and ONE gather kernel (csrc/training.hip::stitch_kernel) writes
out = sum / count with the windows summed in window-index order -- MONAI's accumulation order, so the result is bit-identical to the
sequential `out[window] += pred` loop, without its 700 read-modify-write passes over the 2.3 GB accumulator and without atomics.
A volume whose window logits exceed the resident budget (a share of the card's FREE memory, `resident_budget`) is stitched in slabs of
depth layers: whenever the buffer is full, the depths that no later layer can touch are written (slab form of the same kernel, same
accumulation order) and the layers that still overlap the next depths move to the front of the buffer - MONAI's loop has no size limit
and neither has this one.
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


RESIDENT_LIMIT_BYTES = None      # None: RESIDENT_FRACTION of the device's free memory at the first window; a number overrides it (tests)
RESIDENT_FRACTION = 0.7


def resident_budget(device):
    """bytes of window logits kept in HBM at a time for the gather"""
    if RESIDENT_LIMIT_BYTES is not None:
        return int(RESIDENT_LIMIT_BYTES)
    return int(_free_bytes(device) * RESIDENT_FRACTION)


def _free_bytes(device):
    """bytes this process can still allocate on `device`: what the driver reports free PLUS what torch's caching allocator holds reserved but
    unallocated (after a training run most of the card sits there: mem_get_info alone would size the buffer at its minimum or refuse it)"""
    free, _ = torch.cuda.mem_get_info(device)
    return int(free) + int(torch.cuda.memory_reserved(device)) - int(torch.cuda.memory_allocated(device))


class _SlabStitcher:
    """resident window logits of the depth layers [lo, ...) of one volume + the slab-wise gather (see the module docstring)"""

    def __init__(self, starts, roi, channels, size, device, sw_batch_size):
        self.starts, self.roi, self.size = starts, roi, size
        self.layer = len(starts[1]) * len(starts[2])                       # windows per depth layer
        nd = len(starts[0])
        per_window = channels * roi[0] * roi[1] * roi[2] * 4
        # layers that can overlap one depth (+ the one being filled); the buffer never holds fewer
        span = 1 + max(sum(1 for j in range(i) if starts[0][j] + roi[0] > starts[0][i]) for i in range(nd))
        budget = resident_budget(device)
        self.cap = min(nd, max(budget // (per_window * self.layer), span + 1))       # whole layers kept resident
        need = (self.cap * self.layer + (sw_batch_size if self.cap < nd else 0)) * per_window
        if need > _free_bytes(device):
            raise MemoryError(f"sliding-window stitching needs {need / 1e9:.1f} GB for {self.cap} resident depth layers of {self.layer} windows")
        self.win = torch.empty((self.cap * self.layer + (sw_batch_size if self.cap < nd else 0), channels) + tuple(roi), dtype=torch.float32, device=device)
        self.lo = self.computed = self.done = 0                            # first resident layer, windows computed, depths written

    def reset(self):
        self.lo = self.computed = self.done = 0

    def add(self, pred, out):
        """append the next windows' logits (window-index order); stitches a slab first when they would not fit"""
        n = pred.shape[0]
        if self.computed - self.lo * self.layer + n > self.win.shape[0]:
            self._flush(out, self.computed // self.layer)
        at = self.computed - self.lo * self.layer
        self.win[at:at + n].copy_(pred)
        self.computed += n

    def finish(self, out):
        self._flush(out, len(self.starts[0]))

    def _flush(self, out, upto):
        """write the depths that only the complete layers [lo, upto) cover; keep the layers that reach beyond them"""
        from ..hip import ops
        sd, rd = self.starts[0], self.roi[0]
        nxt = sd[upto] if upto < len(sd) else self.size[0]
        if nxt <= self.done or upto <= self.lo:
            if self.computed - self.lo * self.layer >= self.win.shape[0]:
                raise RuntimeError("sliding-window stitching: the resident buffer holds no complete depth layer to write")
            return
        whole = self.lo == 0 and upto == len(sd)
        ops.stitch_windows(self.win[:(upto - self.lo) * self.layer], out, (sd[self.lo:upto], self.starts[1], self.starts[2]), self.roi,
                           slab=None if whole else (self.done, nxt - self.done))
        self.done = nxt
        keep = next((j for j in range(self.lo, upto) if sd[j] + rd > nxt), upto)
        a, b = (keep - self.lo) * self.layer, self.computed - self.lo * self.layer
        if a and b > a:
            # move the kept layers to the front in place, front to back in non-overlapping pieces of <= a windows (destination below source: a
            # piece never overwrites windows that are still to be moved) - a clone of the tail was a second buffer of nearly the budget's size
            for o in range(0, b - a, a):
                n = min(a, b - a - o)
                self.win[o:o + n].copy_(self.win[a + o:a + o + n])
        self.lo = keep


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
    stitcher = None
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
                    stitcher = _SlabStitcher(starts, roi, pred.shape[1], size, dev, sw_batch_size)
                else:
                    cnt = torch.zeros((1, 1) + size, dtype=torch.float32, device=dev)
            if on_hip:
                stitcher.add(pred, out[b])
                continue
            for j, (d, h, w) in enumerate(chunk):
                out[b, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += pred[j].to(out.device, torch.float32)
                if b == 0:
                    cnt[0, 0, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += 1.0
        if on_hip:
            stitcher.finish(out[b])
            stitcher.reset()
    if not on_hip:
        out = out / cnt
    if any(pads):
        sl = [slice(None), slice(None)] + [slice(p // 2, p // 2 + s) for p, s in zip(pads, orig)]
        out = out[tuple(sl)]
    return out
