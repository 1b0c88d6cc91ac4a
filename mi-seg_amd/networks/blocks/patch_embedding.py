"""PatchEmbed (Swin; reference networks/blocks/patch_embedding.py:125-237): Conv3d(k=s=2)+bias straight from the NCDHW
network input to channels-last tokens, optional norm.  PatchEmbeddingBlock (ViT, MONAI semantics) for UNETR."""
from typing import Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import functional as HF
from ..layers.utils import apply_norm, get_norm_layer
from .swin_transformer_block import norm_spec_with_shape

__all__ = ["PatchEmbed", "PatchEmbeddingBlock"]


class PatchEmbed(nn.Module):
    def __init__(self, patch_size: Union[Sequence[int], int] = 2, in_chans: int = 1, embed_dim: int = 48,
                 norm_type: Union[Tuple, str, None] = "layer", spatial_dims: int = 3) -> None:
        super().__init__()
        if spatial_dims != 3:
            raise ValueError("spatial dimension should be 2 or 3." if spatial_dims not in (2, 3) else
                             "only spatial_dims=3 is implemented by the MI355X path")
        ps = (patch_size,) * 3 if isinstance(patch_size, int) else tuple(patch_size)
        if ps != (2, 2, 2):
            raise NotImplementedError("PatchEmbed: only patch_size 2 is implemented by the MI355X path")
        self.patch_size = ps
        self.embed_dim = embed_dim
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=ps, stride=ps)   # parameter container
        self.norm = None
        if norm_type is not None:
            self.norm = get_norm_layer(name=norm_spec_with_shape(norm_type, embed_dim), spatial_dims=3, channels=embed_dim)

    def forward(self, x, styles=None, dtype=torch.float32):
        _, _, d, h, w = x.shape
        if (d % 2) or (h % 2) or (w % 2):   # patch_embedding.py:189-195 (host-side zero pad of the raw input)
            x = F.pad(x, (0, w % 2, 0, h % 2, 0, d % 2))
        y = HF.patch_embed(x.contiguous(), self.proj.weight, self.proj.bias, dtype)
        if self.norm is not None:
            y = apply_norm(self.norm, y, styles)
        return y


class PatchEmbeddingBlock(nn.Module):
    """MONAI PatchEmbeddingBlock ("perceptron" / "conv"), vendored by the reference at patch_embedding.py:32-123.
    Output [B, n_patches, hidden]."""

    def __init__(self, in_channels: int, img_size, patch_size, hidden_size: int, num_heads: int, pos_embed: str,
                 dropout_rate: float = 0.0, spatial_dims: int = 3) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden size should be divisible by num_heads.")
        if pos_embed not in ("conv", "perceptron"):
            raise ValueError(f"Unsupported option '{pos_embed}'")
        if spatial_dims != 3:
            raise NotImplementedError("only spatial_dims=3 is implemented by the MI355X path")
        self.dropout_rate = float(dropout_rate)
        self.pos_embed = pos_embed
        img_size = (img_size,) * 3 if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size,) * 3 if isinstance(patch_size, int) else tuple(patch_size)
        for m, p in zip(img_size, patch_size):
            if m < p:
                raise ValueError("patch_size should be smaller than img_size.")
            if self.pos_embed == "perceptron" and m % p != 0:
                raise ValueError("patch_size should be divisible by img_size for perceptron.")
        self.patch_size = patch_size
        self.n_patches = int(np.prod([i // p for i, p in zip(img_size, patch_size)]))
        self.patch_dim = int(in_channels * np.prod(patch_size))
        if self.pos_embed == "conv":
            self.patch_embeddings = nn.Conv3d(in_channels, hidden_size, kernel_size=patch_size, stride=patch_size)
        else:
            # index 0 is einops' Rearrange in MONAI (no parameters) -> keys patch_embeddings.1.{weight,bias}
            self.patch_embeddings = nn.Sequential(nn.Identity(), nn.Linear(self.patch_dim, hidden_size))
        self.position_embeddings = nn.Parameter(torch.zeros(1, self.n_patches, hidden_size))
        nn.init.trunc_normal_(self.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x, dtype=torch.float32):
        b, c, H, W, D = x.shape
        p0, p1, p2 = self.patch_size
        # patch gather of the raw fp32 input (a view + one strided copy; the arithmetic starts at the GEMM below)
        if self.pos_embed == "perceptron":
            # "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)"
            t = x.view(b, c, H // p0, p0, W // p1, p1, D // p2, p2).permute(0, 2, 4, 6, 3, 5, 7, 1)
            w2 = self.patch_embeddings[1].weight
            bias = self.patch_embeddings[1].bias
        else:
            # conv k=s=p == linear over (c p1 p2 p3)
            t = x.view(b, c, H // p0, p0, W // p1, p1, D // p2, p2).permute(0, 2, 4, 6, 1, 3, 5, 7)
            w2 = self.patch_embeddings.weight       # [hidden, c, p1, p2, p3] used as the [hidden, c*p1*p2*p3] matrix: the PARAMETER itself goes to
                                                    # HF.linear (a view of it would carry neither its arena slot nor its cached cast)
            bias = self.patch_embeddings.bias
        t = t.reshape(b, self.n_patches, self.patch_dim).to(dtype)
        e = HF.linear(t, w2, bias)
        return HF.dropout(HF.add_position(e, self.position_embeddings), self.dropout_rate, self.training)      # patch_embedding.py:121-122
