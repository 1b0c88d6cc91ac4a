"""UNETR up / projection-up / basic blocks (reference networks/blocks/unetr_block.py)."""
from typing import Sequence, Tuple, Union

import torch.nn as nn

from ...hip import functional as HF
from .dynunet_block import UnetBasicBlock, UnetResBlock, get_conv_layer


class UnetrUpBlock(nn.Module):
    """unetr_block.py:21-85: ConvTranspose3d(k2,s2) -> cat with skip -> Res/Basic block."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size, upsample_kernel_size, norm_name,
                 res_block: bool = False) -> None:
        super().__init__()
        if upsample_kernel_size != 2:
            raise NotImplementedError("only upsample_kernel_size=2 is implemented by the MI355X path")
        self.transp_conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=2, stride=2, conv_only=True,
                                          is_transposed=True)
        blk = UnetResBlock if res_block else UnetBasicBlock
        self.conv_block = blk(spatial_dims, out_channels + out_channels, out_channels, kernel_size=kernel_size, stride=1,
                              norm_name=norm_name)

    def forward(self, inp, skip, styles=None):
        cat = HF.upconv_cat(inp, skip, self.transp_conv.conv.weight)
        return self.conv_block(cat, styles)


class _Seq2(nn.Sequential):
    pass


class UnetrPrUpBlock(nn.Module):
    """unetr_block.py:88-213."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, num_layer: int, kernel_size, stride,
                 upsample_kernel_size, norm_name, conv_block: bool = False, res_block: bool = False) -> None:
        super().__init__()
        if upsample_kernel_size != 2:
            raise NotImplementedError("only upsample_kernel_size=2 is implemented by the MI355X path")
        self.conv_block = conv_block
        self.transp_conv_init = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=2, stride=2, is_transposed=True)
        if conv_block:
            blk = UnetResBlock if res_block else UnetBasicBlock
            self.blocks = nn.ModuleList([
                _Seq2(get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size=2, stride=2, is_transposed=True),
                      blk(spatial_dims=spatial_dims, in_channels=out_channels, out_channels=out_channels, kernel_size=kernel_size,
                          stride=stride, norm_name=norm_name))
                for _ in range(num_layer)])
        else:
            self.blocks = nn.ModuleList([get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size=2, stride=2,
                                                        is_transposed=True) for _ in range(num_layer)])

    def forward(self, x, styles=None):
        x = HF.upconv_cat(x, None, self.transp_conv_init.conv.weight)
        for blk in self.blocks:
            if self.conv_block:
                x = HF.upconv_cat(x, None, blk[0].conv.weight)
                x = blk[1](x, styles)
            else:
                x = HF.upconv_cat(x, None, blk.conv.weight)
        return x


class UnetrBasicBlock(nn.Module):
    """unetr_block.py:216-266."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size, stride, norm_name,
                 res_block: bool = False) -> None:
        super().__init__()
        blk = UnetResBlock if res_block else UnetBasicBlock
        self.layer = blk(spatial_dims=spatial_dims, in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                         stride=stride, norm_name=norm_name)

    def forward(self, inp, styles=None, image=None, dtype=None, out_view=None):
        return self.layer(inp, styles, image=image, dtype=dtype, out_view=out_view)
