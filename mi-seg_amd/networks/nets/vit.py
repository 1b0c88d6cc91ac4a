"""ViT encoder (reference networks/nets/vit.py) -- segmentation backbone only (classification heads are never built by
UNETR: unetr.py:95,115)."""
from typing import Sequence, Tuple, Union

import torch
import torch.nn as nn

from ...hip import functional as HF
from ..blocks.patch_embedding import PatchEmbeddingBlock
from ..blocks.swin_transformer_block import norm_spec_with_shape
from ..blocks.transformer_block import TransformerBlock
from ..layers.utils import apply_norm, get_norm_layer
from ..norms.conditional_instance_norm import _ConditionalInstanceNorm

__all__ = ["ViT"]


class ViT(nn.Module):
    def __init__(self, in_channels: int, img_size: Union[Sequence[int], int], patch_size: Union[Sequence[int], int], hidden_size: int = 768,
                 mlp_dim: int = 3072, num_layers: int = 12, num_heads: int = 12, pos_embed: str = "conv", classification: bool = False,
                 num_classes: int = 2, dropout_rate: float = 0.0, spatial_dims: int = 3, post_activation="Tanh", qkv_bias: bool = False,
                 norm_type: Union[Tuple, str] = "layer", classification_reverse_gradient: bool = False, alpha_reversal: float = 1.0) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        if classification:
            raise NotImplementedError("ViT classification heads are outside the segmentation hot path")
        self.norm_type = norm_type[0] if isinstance(norm_type, tuple) else norm_type
        self.classification = False
        img = (img_size,) * spatial_dims if isinstance(img_size, int) else tuple(img_size)
        ps = (patch_size,) * spatial_dims if isinstance(patch_size, int) else tuple(patch_size)
        self.grid = tuple(i // p for i, p in zip(img, ps))
        self.patch_embedding = PatchEmbeddingBlock(in_channels=in_channels, img_size=img_size, patch_size=patch_size, hidden_size=hidden_size,
                                                   num_heads=num_heads, pos_embed=pos_embed, dropout_rate=dropout_rate, spatial_dims=spatial_dims)
        self.blocks = nn.ModuleList([TransformerBlock(hidden_size, mlp_dim, num_heads, dropout_rate, qkv_bias, norm_type=norm_type)
                                     for _ in range(num_layers)])
        self.norm = get_norm_layer(name=norm_spec_with_shape(norm_type, hidden_size), spatial_dims=1, channels=hidden_size)

    def forward(self, x, styles=None, dtype=torch.float32):
        if isinstance(self.norm, _ConditionalInstanceNorm) and styles is None:
            raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
        x = self.patch_embedding(x, dtype)
        hidden = []
        for blk in self.blocks:
            x = blk(x, styles, self.grid)
            a, x = HF.fork(x)
            hidden.append(a)
        return apply_norm(self.norm, x, styles), hidden
