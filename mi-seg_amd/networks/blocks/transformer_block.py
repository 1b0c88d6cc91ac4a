"""TransformerBlock (reference networks/blocks/transformer_block.py:22-110): x + SA(norm1(x)); x + MLP(norm2(x)) on [B, L, C]."""
from typing import Tuple, Union

import torch.nn as nn

from ...hip import functional as HF
from ..layers.utils import apply_norm_fork, get_norm_layer
from ..norms.conditional_instance_norm import _ConditionalInstanceNorm
from .mlp import MLPBlock
from .selfattention import SABlock
from .swin_transformer_block import norm_spec_with_shape


class TransformerBlock(nn.Module):
    def __init__(self, hidden_size: int, mlp_dim: int, num_heads: int, dropout_rate: float = 0.0, qkv_bias: bool = False,
                 norm_type: Union[Tuple, str] = "layer") -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        self.norm_type = norm_type[0] if isinstance(norm_type, tuple) else norm_type
        self.mlp = MLPBlock(hidden_size, mlp_dim, dropout_rate)
        self.attn = SABlock(hidden_size, num_heads, dropout_rate, qkv_bias)
        spec = norm_spec_with_shape(norm_type, hidden_size)
        self.norm1 = get_norm_layer(name=spec, spatial_dims=1, channels=hidden_size)   # spatial_dims 1: (B, C, L)
        self.norm2 = get_norm_layer(name=spec, spatial_dims=1, channels=hidden_size)

    def forward(self, x, styles=None, grid=None):
        if isinstance(self.norm1, _ConditionalInstanceNorm) and styles is None:
            raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
        # as in the Swin block: the residual adds ride in the epilogues of out_proj / linear2, the fan-out sums of the backward pass in the
        # norm-backward kernels (round 3: 48 stand-alone add launches per C-UNETR step before)
        xn, xs = apply_norm_fork(self.norm1, x, styles)
        x = self.attn(xn, grid, res=xs)
        xn, xs = apply_norm_fork(self.norm2, x, styles)
        return self.mlp(xn, res=xs)
