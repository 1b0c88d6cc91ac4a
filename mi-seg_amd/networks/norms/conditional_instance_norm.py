"""Modality-conditional InstanceNorm -- drop-in for reference networks/norms/conditional_instance_norm.py.

Same constructor, same ``norms.{s}.{weight,bias}`` parameters, same errors; the arithmetic is one statistics kernel +
one fused normalise/affine kernel per call for the whole batch (no per-sample Python loop, no device sync: the
style ids are read on the device).  ``forward(input, styles)`` takes the reference's channels-first tensors;
the networks in this package call :func:`apply_norm` directly on their channels-last activations instead.
"""
import warnings
from typing import List, Union

import torch
import torch.nn as nn
from torch import Tensor

from ...hip import functional as HF
from ...hip import lib as L

__all__ = ["ConditionalInstanceNorm1d", "ConditionalInstanceNorm2d", "ConditionalInstanceNorm3d", "styles_to_device"]


def _check_range(host, num_styles):
    """the reference indexes an nn.ModuleList with the style id (conditional_instance_norm.py:59-60): ids outside [-num_styles, num_styles)
    raise IndexError there.  Here the kernels index by-value argument arrays of MISEG_MAX_STYLES rows with ids read on the DEVICE, so an
    id that slipped through would be a wild pointer - every id is checked on the host before any launch."""
    if num_styles is None:
        return host
    out = []
    for s in host:
        if not -num_styles <= s < num_styles:
            raise IndexError(f"index {s} is out of range (modality / style ids must lie in [0, {num_styles}))")
        out.append(s + num_styles if s < 0 else s)
    return tuple(out)


def styles_to_device(styles, device, batch, num_styles=None):
    """-> (int32 device tensor [B], python tuple) ; at most one host sync when `styles` lives on the device.
    A ready-made (device tensor, host tuple) pair is passed through (hipGraph capture: no host traffic).
    num_styles: the ids are range-checked against it (IndexError, like the reference's ModuleList lookup)."""
    if isinstance(styles, tuple) and len(styles) == 2 and isinstance(styles[0], Tensor) and isinstance(styles[1], tuple):
        if len(styles[1]) != batch:
            raise ValueError("Expected number of styles as batch size.")
        if _check_range(styles[1], num_styles) != styles[1]:
            raise IndexError("negative style ids must be normalised before they are handed over as a (device, host) pair")
        return styles
    if isinstance(styles, Tensor):
        host = tuple(int(s) for s in styles.reshape(-1).tolist())
    elif isinstance(styles, int):
        host = (styles,)
    else:
        host = tuple(int(s) for s in styles)
    if len(host) != batch:
        raise ValueError("Expected number of styles as batch size.")
    host = _check_range(host, num_styles)
    return torch.tensor(host, dtype=torch.int32, device=device), host


def styles_limit(module):
    """the smallest num_styles over the conditional norms of a net (None without any): cached on the module"""
    lim = module.__dict__.get("_miseg_styles_limit", 0)
    if lim == 0:
        ns = [m.num_styles for m in module.modules() if isinstance(m, _ConditionalInstanceNorm)]
        lim = module.__dict__["_miseg_styles_limit"] = min(ns) if ns else None
    return lim


class _ConditionalInstanceNorm(nn.Module):
    def __init__(self, num_styles: int, num_features: int, eps: float = 1e-5, momentum: float = 0.1, affine: bool = True,
                 track_running_stats: bool = False, device=None, dtype=None) -> None:
        super().__init__()
        if not affine:
            warnings.warn("Ignored affine=False for ConditionalInstanceNorm1D, set to True")
        if track_running_stats:
            raise NotImplementedError("track_running_stats=True is not supported by the HIP path")
        if not 1 <= num_styles <= L.MAX_STYLES:
            raise NotImplementedError(f"num_styles={num_styles}: the HIP kernels take at most {L.MAX_STYLES} affine rows (MISEG_MAX_STYLES)")
        self.num_styles = num_styles
        self.num_features = num_features
        self.eps = eps
        kw = {"device": device, "dtype": dtype}
        # nn.InstanceNorm*d modules are used as parameter containers only (identical state_dict keys / init)
        self.norms = nn.ModuleList([self._get_norm()(num_features, eps, momentum, True, False, **kw) for _ in range(num_styles)])

    def _get_norm(self):
        raise NotImplementedError

    def _get_no_batch_dim(self):
        raise NotImplementedError

    def _check_input_dim(self, input):
        raise NotImplementedError

    def _check_input_styles(self, input, styles):
        # reference conditional_instance_norm.py:40-47
        if input.dim() == self._get_no_batch_dim():
            if not isinstance(styles, (int, list, Tensor)) or (isinstance(styles, Tensor) and torch.numel(styles) != 1) \
                    or (isinstance(styles, list) and len(styles) != 1):
                raise ValueError("Expected one style when input is not a batch.")
        else:
            if not isinstance(styles, (list, Tensor)) or len(styles) != len(input):
                raise ValueError("Expected number of styles as batch size.")

    def style_params(self):
        return [(n.weight, n.bias) for n in self.norms]

    def forward(self, input: Tensor, styles: Union[List, Tensor, int]) -> Tensor:
        self._check_input_dim(input)
        self._check_input_styles(input, styles)
        unbatched = input.dim() == self._get_no_batch_dim()
        x = input.unsqueeze(0) if unbatched else input
        sd, sh = styles_to_device(styles, x.device, x.shape[0], self.num_styles)
        xl = x.movedim(1, -1).contiguous()          # NC* -> N*C rows (boundary conversion only)
        y = HF.instance_norm(xl, self.style_params(), sd, sh, eps=self.eps)
        y = y.movedim(-1, 1)
        return y.squeeze(0) if unbatched else y


class ConditionalInstanceNorm1d(_ConditionalInstanceNorm):
    def _get_norm(self):
        return nn.InstanceNorm1d

    def _get_no_batch_dim(self):
        return 2

    def _check_input_dim(self, input):
        if input.dim() not in (2, 3):
            raise ValueError("expected 2D or 3D input (got {}D input)".format(input.dim()))


class ConditionalInstanceNorm2d(_ConditionalInstanceNorm):
    def _get_norm(self):
        return nn.InstanceNorm2d

    def _get_no_batch_dim(self):
        return 3

    def _check_input_dim(self, input):
        if input.dim() not in (3, 4):
            raise ValueError("expected 2D or 3D input (got {}D input)".format(input.dim()))


class ConditionalInstanceNorm3d(_ConditionalInstanceNorm):
    def _get_norm(self):
        return nn.InstanceNorm3d

    def _get_no_batch_dim(self):
        return 4

    def _check_input_dim(self, input):
        if input.dim() not in (4, 5):
            raise ValueError("expected 4D or 5D input (got {}D input)".format(input.dim()))
