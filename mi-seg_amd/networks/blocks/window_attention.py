"""WindowAttention -- parameters and buffers laid out as reference networks/blocks/window_attention.py:51-97;
forward (:99-122) = qkv GEMM -> fused attention core kernel -> proj GEMM, on the UNPARTITIONED channels-last grid."""
from typing import Sequence

import torch
import torch.nn as nn

from ...hip import functional as HF

__all__ = ["WindowAttention"]


class WindowAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int, window_size: Sequence[int], qkv_bias: bool = False, attn_drop: float = 0.0,
                 proj_drop: float = 0.0) -> None:
        super().__init__()
        self.attn_drop = float(attn_drop)       # window_attention.py:93,114: drawn inside the fused attention kernels, the scores never leave them
        self.proj_drop = float(proj_drop)
        if len(window_size) != 3:
            raise NotImplementedError("only 3D windows are implemented by the MI355X path")
        self.dim = dim
        self.window_size = tuple(window_size)
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        ws = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1), num_heads))
        # buffer kept for checkpoint compatibility (window_attention.py:58-72,90); the kernel recomputes the index
        coords = torch.stack(torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), torch.arange(ws[2]), indexing="ij"))
        cf = torch.flatten(coords, 1)
        rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws[0] - 1
        rel[:, :, 1] += ws[1] - 1
        rel[:, :, 2] += ws[2] - 1
        rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
        rel[:, :, 1] *= 2 * ws[2] - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        if not (ws[0] == ws[1] == ws[2]):
            raise NotImplementedError("non-cubic table windows")

    def forward(self, x, window, shift, res=None, want_stat=False, qkv=None):
        """x: normalised tokens on the unpadded grid [B, D, H, W, C]; window/shift already clamped.
        qkv: the caller already formed qkv(x) (the Swin block folds its norm1 into that GEMM: HF.norm_linear); x is then ignored."""
        if qkv is None:
            qkv = HF.linear(x, self.qkv.weight, self.qkv.bias)
        o = HF.window_attention(qkv, self.qkv.bias, self.relative_position_bias_table, self.num_heads, window, shift,
                                self.window_size[0], self.scale, self.attn_drop, self.training)
        if self.proj_drop > 0.0 and self.training:        # window_attention.py:120-121: proj_drop(proj(x)); the residual is then added separately
            y = HF.dropout(HF.linear(o, self.proj.weight, self.proj.bias), self.proj_drop)
            return HF.add(res, y) if res is not None else y
        return HF.linear(o, self.proj.weight, self.proj.bias, res, want_stat)     # res: the block's residual, added in the GEMM epilogue
