"""Name-keyed deterministic parameter fill.

Every ``state_dict`` entry is overwritten with values that depend only on its key and
shape, so the reference modules (fixture generator), the CPU oracle and the HIP product
hold bit-identical weights without shipping checkpoints or depending on init order.
"""
import zlib

import numpy as np
import torch


def det_values(name: str, shape) -> np.ndarray:
    """float32 values for the tensor called ``name`` (counter-based Philox stream keyed by crc32(name))."""
    shape = tuple(int(s) for s in shape)
    rng = np.random.Generator(np.random.Philox(key=zlib.crc32(name.encode("utf-8"))))
    n = rng.standard_normal(size=shape, dtype=np.float64)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "relative_position_bias_table":
        v = 0.5 * n
    elif leaf == "position_embeddings" or leaf == "cls_token":
        v = 0.2 * n
    elif name.endswith(".A.weight"):          # PReLU slope (UNet ADN), reference acti_norm.py
        v = 0.25 + 0.05 * n
    elif leaf == "weight" and len(shape) == 1:  # norm scale
        v = 1.0 + 0.2 * n
    elif leaf == "bias":
        v = 0.1 * n
    else:                                      # conv / linear / transposed-conv weight
        fan_in = max(1, int(np.prod(shape[1:])))
        v = n / np.sqrt(fan_in)
    return v.astype(np.float32)


@torch.no_grad()
def fill_state_dict_(sd) -> None:
    """In-place deterministic fill of every floating-point entry of ``sd`` (integer buffers are kept)."""
    for k, t in sd.items():
        if not torch.is_floating_point(t):
            continue
        t.copy_(torch.from_numpy(det_values(k, t.shape)).to(t.dtype))


@torch.no_grad()
def fill_module_(module) -> None:
    fill_state_dict_(module.state_dict())


def det_input(seed: int, shape) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def block_labels(shape, n_classes: int, block: int = 16) -> torch.Tensor:
    """deterministic piecewise-constant label volume [B, D, H, W]: class = (d//block + h//block + w//block + b) % n_classes."""
    B, D, H, W = shape
    d = torch.arange(D).view(1, D, 1, 1) // block
    h = torch.arange(H).view(1, 1, H, 1) // block
    w = torch.arange(W).view(1, 1, 1, W) // block
    b = torch.arange(B).view(B, 1, 1, 1)
    return (d + h + w + b) % n_classes


def ce_cotangent(logits: torch.Tensor) -> torch.Tensor:
    """d(mean voxel cross-entropy against block_labels)/d(logits) = (softmax - onehot) / n_voxels.  A spatially coherent
    cotangent: unlike white noise it does not make every parameter gradient a sqrt(N)-cancelling random sum, so it is the
    one used to judge the bf16 path."""
    B, C = logits.shape[0], logits.shape[1]
    lab = block_labels((B,) + tuple(logits.shape[2:]), C).to(logits.device)
    p = torch.softmax(logits.detach().float(), dim=1)
    onehot = torch.nn.functional.one_hot(lab, C).movedim(-1, 1).to(p.dtype)
    return (p - onehot) / float(lab.numel())
