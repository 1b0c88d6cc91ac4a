"""PatchMerging / PatchMergingV2 (reference networks/blocks/patch_merging.py): 2x2x2 gather with the reference's slice
tables (incl. the duplicated v0.9 slices) -> (cond-)norm over 8C -> bias-free Linear 8C -> 2C."""
from typing import Tuple, Union

import torch.nn as nn

from ...hip import functional as HF
from ..layers.utils import apply_norm, get_norm_layer
from .swin_transformer_block import norm_spec_with_shape

__all__ = ["PatchMerging", "PatchMergingV2"]


class PatchMergingV2(nn.Module):
    offsets = HF.STD_OFFSETS

    def __init__(self, dim: int, norm_type: Union[Tuple, str] = "instance_cond", spatial_dims: int = 3) -> None:
        super().__init__()
        if spatial_dims != 3:
            raise NotImplementedError("only spatial_dims=3 is implemented by the MI355X path")
        self.norm_type = norm_type[0] if isinstance(norm_type, tuple) else norm_type
        self.dim = dim
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = get_norm_layer(name=norm_spec_with_shape(norm_type, 8 * dim), spatial_dims=spatial_dims, channels=8 * dim)

    def forward(self, x, styles=None):
        if x.dim() != 5:
            raise ValueError(f"expecting 5D x, got {x.shape}.")
        x = HF.space_to_channel(x, self.offsets)          # odd grids are zero-padded by the gather itself
        x = apply_norm(self.norm, x, styles)
        # the result feeds instance norms (the stage's output norm and the next block's norm1): statistics from the GEMM's epilogue where it has one
        return HF.linear(x, self.reduction.weight, None, want_stat=self.norm_type.startswith("instance"))


class PatchMerging(PatchMergingV2):
    """v0.9 slice order: offsets (0,1,0) and (0,0,1) appear twice, (1,1,0) and (0,1,1) never (patch_merging.py:120-127)."""
    offsets = HF.MERGE_V1_OFFSETS
