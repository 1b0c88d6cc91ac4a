"""UNETR / DynUNet CNN blocks (reference networks/blocks/dynunet_block.py) on channels-last activations.

``Convolution`` here is a parameter container with MONAI's child name ``conv`` so that state_dict keys match
(``...conv1.conv.weight``); the arithmetic is in the block forwards: implicit-GEMM 3x3x3 convs, the fused
(conditional) instance-norm + residual + LeakyReLU kernel, 1x1x1 convs as GEMMs.
"""
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from ...hip import functional as HF
from ...hip import lib as L
from ...hip import ops
from ..layers.utils import apply_norm, apply_res_norm_pair, get_norm_layer, stat_request
from ..norms.conditional_instance_norm import _ConditionalInstanceNorm

LEAKY_SLOPE = 0.01


def get_padding(kernel_size, stride):
    """reference dynunet_block.py:329-340."""
    k, s = np.atleast_1d(kernel_size), np.atleast_1d(stride)
    p = (k - s + 1) / 2
    if np.min(p) < 0:
        raise AssertionError("padding value should not be negative, please change the kernel size and/or stride.")
    p = tuple(int(v) for v in p)
    return p if len(p) > 1 else p[0]


def get_output_padding(kernel_size, stride, padding):
    k, s, p = np.atleast_1d(kernel_size), np.atleast_1d(stride), np.atleast_1d(padding)
    o = 2 * p + s - k
    if np.min(o) < 0:
        raise AssertionError("out_padding value should not be negative, please change the kernel size and/or stride.")
    o = tuple(int(v) for v in o)
    return o if len(o) > 1 else o[0]


class Convolution(nn.Module):
    """bare conv holder (MONAI ``Convolution`` with conv_only / no ADN as built by get_conv_layer :295-326)."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, bias=False, is_transposed=False):
        super().__init__()
        if spatial_dims != 3:
            raise NotImplementedError("only spatial_dims=3 is implemented by the MI355X path")
        pad = get_padding(kernel_size, stride)
        if is_transposed:
            self.conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=pad,
                                           output_padding=get_output_padding(kernel_size, stride, pad), bias=bias)
        else:
            self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=pad, bias=bias)
        self.kernel_size, self.stride, self.is_transposed = kernel_size, stride, is_transposed


def get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=3, stride=1, act=None, norm=None, dropout=None, bias=False,
                   conv_only=True, is_transposed=False):
    if dropout:
        raise NotImplementedError("dropout is not implemented by the MI355X path")
    return Convolution(spatial_dims, in_channels, out_channels, kernel_size, stride, bias=bias, is_transposed=is_transposed)


def _check_block_args(kernel_size, stride):
    if tuple(np.atleast_1d(kernel_size).tolist()) not in ((3,), (3, 3, 3)) or not np.all(np.atleast_1d(stride) == 1):
        raise NotImplementedError("UNETR blocks: only kernel_size=3, stride=1 are implemented by the MI355X path")


def _needs_modalities(norm, styles):
    if isinstance(norm, _ConditionalInstanceNorm) and styles is None:
        raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")


class UnetResBlock(nn.Module):
    """dynunet_block.py:26-126: conv3 -> norm -> lrelu -> conv3 -> norm (+ 1x1x1 conv + norm shortcut) -> add -> lrelu."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size, stride, norm_name: Union[Tuple, str],
                 act_name=("leakyrelu", {"inplace": True, "negative_slope": 0.01}), dropout=None):
        super().__init__()
        _check_block_args(kernel_size, stride)
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=kernel_size, stride=stride)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size=kernel_size, stride=1)
        self.lrelu = nn.LeakyReLU(inplace=True, negative_slope=LEAKY_SLOPE)
        self.norm_name = norm_name[0] if isinstance(norm_name, tuple) else norm_name
        self.norm1 = get_norm_layer(name=norm_name, spatial_dims=spatial_dims, channels=out_channels)
        self.norm2 = get_norm_layer(name=norm_name, spatial_dims=spatial_dims, channels=out_channels)
        self.downsample = in_channels != out_channels
        if self.downsample:
            self.conv3 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=1, stride=stride)
            self.norm3 = get_norm_layer(name=norm_name, spatial_dims=spatial_dims, channels=out_channels)
        self.in_channels = in_channels

    def forward(self, inp, styles=None, image=None, dtype=None, out_view=None):
        """inp: channels-last activation; for a block fed by the raw image pass ``image`` (NCDHW fp32) instead.
        out_view: rows view the result is written into (HF.concat_buffer)."""
        _needs_modalities(self.norm1, styles)
        shortcut_done = False
        if image is not None and image.shape[1] != 1:
            # multi-channel image (--in_channels > 1, dynunet_block.py:82-98): rows of C channels through the ordinary kernels; the
            # one-channel shortcuts below (rank-1 1x1x1 convolution, image-as-rows view) are a special case of the headline configuration
            inp, image = HF.image_rows(image, dtype), None
        if image is not None:
            out = HF.conv3_thin(image, self.conv1.conv.weight, dtype)
            residual = _image_rows(image, dtype)
            st1 = None
        elif inp.requires_grad:
            # conv1 hands its input back as the residual branch (its data-gradient epilogue adds that branch's gradient) and the
            # statistics of its output come from its epilogue
            sc3 = None
            if self.downsample:      # (only where the launch does not split its reduction: a "defer" request would not defer there either)
                # round 5: the shortcut convolution rides in conv1's launches - forward as a second output, backward as one more centre tap
                sc3 = HF.conv3_shortcut(inp, self.conv1.conv.weight, self.conv3.conv.weight, want_stat=True)
            if sc3 is not None:
                out, st1, residual = sc3
                shortcut_done = True
            else:
                out, st1, residual = HF.conv3(inp, self.conv1.conv.weight, want_stat=stat_request(self.norm1), fork=True)
        else:
            (out, st1), residual = HF.conv3(inp, self.conv1.conv.weight, want_stat=stat_request(self.norm1)), inp
        out = apply_norm(self.norm1, out, styles, act=L.ACT_LEAKY, slope=LEAKY_SLOPE, stat=st1)
        # (with a shortcut convolution the pair kernels below read the finished tensor and its statistics)
        out, st2 = HF.conv3(out, self.conv2.conv.weight, want_stat=True if self.downsample else stat_request(self.norm2),
                            dx_to_norm=stat_request(self.norm1) == "defer")      # (norm1's output has no other reader)
        if self.downsample:
            if image is not None:       # one-channel image: the shortcut convolution is a rank-1 product, formed inside the norm kernels
                y = apply_res_norm_pair(self.norm2, out, self.norm3, residual, styles, slope=LEAKY_SLOPE, stat_a=st2, out=out_view,
                                        w1=self.conv3.conv.weight)
                if y is not None:
                    return y
            if not shortcut_done:
                residual = HF.conv1(residual, self.conv3.conv.weight, want_stat=True)
            y = apply_res_norm_pair(self.norm2, out, self.norm3, residual, styles, slope=LEAKY_SLOPE, stat_a=st2, out=out_view)
            if y is not None:           # the shortcut's norm rides in the final apply pass
                return y
            residual = apply_norm(self.norm3, residual, styles)
        return apply_norm(self.norm2, out, styles, res=residual, act=L.ACT_LEAKY, slope=LEAKY_SLOPE, stat=st2, out=out_view)


class UnetBasicBlock(nn.Module):
    """dynunet_block.py:129-201: conv3 -> norm -> lrelu -> conv3 -> norm -> lrelu."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size, stride, norm_name: Union[Tuple, str],
                 act_name=("leakyrelu", {"inplace": True, "negative_slope": 0.01}), dropout=None):
        super().__init__()
        _check_block_args(kernel_size, stride)
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=kernel_size, stride=stride)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size=kernel_size, stride=1)
        self.lrelu = nn.LeakyReLU(inplace=True, negative_slope=LEAKY_SLOPE)
        self.norm_name = norm_name[0] if isinstance(norm_name, tuple) else norm_name
        self.norm1 = get_norm_layer(name=norm_name, spatial_dims=spatial_dims, channels=out_channels)
        self.norm2 = get_norm_layer(name=norm_name, spatial_dims=spatial_dims, channels=out_channels)

    def forward(self, inp, styles=None, image=None, dtype=None, out_view=None):
        _needs_modalities(self.norm1, styles)
        if image is not None and image.shape[1] > 4:
            inp, image = HF.image_rows(image, dtype), None
        out, st1 = ((HF.conv3_thin(image, self.conv1.conv.weight, dtype), None) if image is not None
                    else HF.conv3(inp, self.conv1.conv.weight, want_stat=stat_request(self.norm1)))
        out = apply_norm(self.norm1, out, styles, act=L.ACT_LEAKY, slope=LEAKY_SLOPE, stat=st1)
        out, st2 = HF.conv3(out, self.conv2.conv.weight, want_stat=stat_request(self.norm2), dx_to_norm=stat_request(self.norm1) == "defer")
        return apply_norm(self.norm2, out, styles, act=L.ACT_LEAKY, slope=LEAKY_SLOPE, stat=st2, out=out_view)


def _image_rows(image, dtype):
    """raw NCDHW fp32 image -> channels-last rows in the compute dtype (C == 1: same memory order)."""
    b, c, d, h, w = image.shape
    if c != 1:
        raise NotImplementedError("in_channels > 1 for the residual shortcut of the stem block")
    v = image.view(b, d, h, w, 1)
    if dtype == torch.float32:
        return v
    return ops.copy2d(v, torch.empty(b, d, h, w, 1, dtype=dtype, device=image.device))


class UnetOutBlock(nn.Module):
    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, dropout=None):
        super().__init__()
        self.conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=1, stride=1, bias=True)

    def forward(self, inp):
        return HF.head(inp, self.conv.conv.weight, self.conv.conv.bias)
