"""SwinTransformer / BasicLayer (reference networks/nets/swin_transformer.py) -- channels-last throughout; the
5 returned feature maps are [B, D, H, W, C] tensors (the reference's internal NCDHW round trips are gone)."""
from typing import Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from ...hip import functional as HF
from ..blocks.patch_embedding import PatchEmbed
from ..blocks.patch_merging import PatchMerging, PatchMergingV2
from ..blocks.swin_transformer_block import SwinTransformerBlock

__all__ = ["BasicLayer", "SwinTransformer", "MERGING_MODE"]

MERGING_MODE = {"merging": PatchMerging, "mergingv2": PatchMergingV2}


def look_up_option(key, supported):
    if key in supported:
        return supported[key] if isinstance(supported, dict) else key
    raise ValueError(f"Unsupported option '{key}', available: {list(supported)}")


class BasicLayer(nn.Module):
    def __init__(self, dim: int, depth: int, num_heads: int, window_size: Sequence[int], drop_path: list, mlp_ratio: float = 4.0,
                 qkv_bias: bool = False, drop: float = 0.0, attn_drop: float = 0.0, downsample=None, use_checkpoint: bool = False,
                 norm_type: Union[Tuple, str] = "layer") -> None:
        super().__init__()
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in window_size)
        self.no_shift = tuple(0 for _ in window_size)
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, num_heads=num_heads, window_size=self.window_size,
                                 shift_size=self.no_shift if (i % 2 == 0) else self.shift_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                 drop=drop, attn_drop=attn_drop, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 use_checkpoint=use_checkpoint, norm_type=norm_type)
            for i in range(depth)])
        self.downsample = downsample
        if callable(self.downsample):
            self.downsample = downsample(dim=dim, norm_type=norm_type, spatial_dims=len(self.window_size))

    def forward(self, x, styles=None):
        """x [B, D, H, W, C] -> [B, D/2, H/2, W/2, 2C].  (the shift mask of swin_transformer.py:237 is computed inside
        the attention kernel from region labels)"""
        for blk in self.blocks:
            x = blk(x, styles)
        if self.downsample is not None:
            x = self.downsample(x, styles)
        return x


class SwinTransformer(nn.Module):
    def __init__(self, in_chans: int, embed_dim: int, window_size: Sequence[int], patch_size: Sequence[int], depths: Sequence[int],
                 num_heads: Sequence[int], mlp_ratio: float = 4.0, qkv_bias: bool = True, drop_rate: float = 0.0,
                 attn_drop_rate: float = 0.0, drop_path_rate: float = 0.0, patch_norm: bool = False, use_checkpoint: bool = False,
                 spatial_dims: int = 3, downsample="merging", norm_type: Union[Tuple, str] = "layer") -> None:
        super().__init__()
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.patch_norm = patch_norm
        self.window_size = tuple(window_size)
        self.patch_size = tuple(patch_size)
        self.norm_type = norm_type[0] if isinstance(norm_type, tuple) else norm_type
        self.patch_embed = PatchEmbed(patch_size=self.patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_type=norm_type if self.patch_norm else None, spatial_dims=spatial_dims)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.drop_rate = float(drop_rate)
        # stochastic depth decay rule of the reference (swin_transformer.py:88): linspace(0, drop_path_rate, sum(depths))
        dpr = [float(v) for v in torch.linspace(0, drop_path_rate, sum(depths), device="cpu")]      # (a model may be built under torch.device("meta"))
        self.layers1, self.layers2, self.layers3, self.layers4 = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        down = look_up_option(downsample, MERGING_MODE) if isinstance(downsample, str) else downsample
        for i_layer in range(self.num_layers):
            layer = BasicLayer(dim=int(embed_dim * 2 ** i_layer), depth=depths[i_layer], num_heads=num_heads[i_layer],
                               window_size=self.window_size, drop_path=dpr[sum(depths[:i_layer]):sum(depths[:i_layer + 1])], mlp_ratio=mlp_ratio,
                               qkv_bias=qkv_bias,
                               drop=drop_rate, attn_drop=attn_drop_rate, downsample=down, use_checkpoint=use_checkpoint,
                               norm_type=norm_type)
            (self.layers1, self.layers2, self.layers3, self.layers4)[i_layer].append(layer)
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))

    def proj_out(self, x, normalize=False):
        """swin_transformer.py:121-145: affine-less norm of a returned feature map."""
        if not normalize:
            return x
        if self.norm_type == "layer":
            return HF.layer_norm(x, None, None)
        if self.norm_type in ("instance", "instance_cond"):
            return HF.instance_norm(x, None)
        return x

    def forward(self, x, normalize=True, styles=None, dtype=torch.float32, after_stage1=None, on_feature=None):
        """x: NCDHW fp32 network input.  Returns 5 channels-last feature maps.
        after_stage1: optional callable(first feature map), called once the launches of `layers1` are queued (SwinUNETR's inference
        forward forks its image-resolution encoder blocks there).
        on_feature: optional callable(i, feature map i), called as soon as the launches that produce returned feature map i are queued
        (SwinUNETR's training forward forks its side branch at one of them)."""
        x0 = self.patch_embed(x, styles, dtype)
        x0 = HF.dropout(x0, self.drop_rate, self.training)        # pos_drop (swin_transformer.py:149)
        outs = []
        cur = x0
        inst = normalize and self.norm_type in ("instance", "instance_cond")
        for layers in (self.layers1, self.layers2, self.layers3, self.layers4):
            if inst:        # (norm(x), x): the norm-backward kernel sums the gradients of the two branches, no separate add
                a, cur = HF.instance_norm(cur, None, fork=True)
                outs.append(a)
            else:
                a, cur = HF.fork(cur)
                outs.append(self.proj_out(a, normalize))
            if on_feature is not None:
                on_feature(len(outs) - 1, outs[-1])
            cur = layers[0](cur, styles)
            if after_stage1 is not None and layers is self.layers1:
                after_stage1(outs[0])
        outs.append(self.proj_out(cur, normalize))
        return outs
