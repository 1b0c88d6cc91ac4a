"""MONAI-style residual UNet with modality-aware ADN blocks (reference networks/nets/unet.py, blocks/convolutions.py,
blocks/acti_norm.py).

BASELINE config 1 is "plumbing, no GPU": constructor, ``from_argparse_args`` (incl. the ``fs * 2**i, i = 1..num_layers`` channel rule
of unet.py:218-219) and the state_dict layout are reproduced here so checkpoints and the model factory work unchanged; its
arithmetic is pinned on the CPU oracle (oracle/nets.py::unet_forward, tests/test_oracle_golden.py::test_unet).

The modules below are parameter containers with the reference's child names; ``UNet.forward`` walks them over the HIP kernels on
channels-last activations: a stride-2 3x3x3 convolution is the stride-1 implicit-GEMM kernel followed by the even-voxel pick
(``miseg_resample2``), ConvTranspose3d k3 s2 p1 op1 is zero insertion followed by the same kernel on the mirrored / channel-swapped pack
(exactly each other's adjoints, so the backward passes reuse them); bias, PReLU and the NCDHW output transpose are small row kernels.
This is the correctness route for the plumbing config, not a tuned one (the strided convolution computes 8x the needed voxels)."""
import warnings
from typing import Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from ...hip import functional as HF
from ...hip import ops
from ..layers.utils import apply_norm, get_norm_layer
from ..norms.conditional_instance_norm import _ConditionalInstanceNorm, styles_limit, styles_to_device
from ..norms.utils import parse_normalization

__all__ = ["UNet", "Unet"]


class ADN(nn.Sequential):
    """children N / D / A in `ordering` (acti_norm.py:68-102); Dropout(p=0) is kept as a parameter-free child like the reference."""

    def __init__(self, ordering, in_channels, act, norm, norm_dim, dropout):
        super().__init__()
        ops = {"N": None, "D": None, "A": None}
        if norm is not None:
            ops["N"] = get_norm_layer(name=norm, spatial_dims=norm_dim, channels=in_channels)
        if act is not None:
            # the reference hands --activation to MONAI's Act factory (networks/nets/unet.py:116-134, utils/parser.py:57): the kinds the HIP
            # path has kernels for - prelu (one learned slope), relu / leakyrelu (the same kernels with a constant slope), gelu (exact erf)
            kind = str(act[0] if isinstance(act, (tuple, list)) else act).lower()
            kw = dict(act[1]) if isinstance(act, (tuple, list)) and len(act) > 1 else {}
            makers = {"prelu": nn.PReLU, "relu": nn.ReLU, "leakyrelu": nn.LeakyReLU, "gelu": nn.GELU}
            if kind not in makers:
                raise NotImplementedError(f"UNet activation '{act}' is not implemented by the MI355X path (supported: {sorted(makers)})")
            ops["A"] = makers[kind](**kw)
            if kind == "gelu" and getattr(ops["A"], "approximate", "none") != "none":
                raise NotImplementedError("GELU(approximate='tanh') is not implemented by the MI355X path (exact erf only)")
        if dropout is not None:
            ops["D"] = nn.Dropout(float(dropout))
        for item in ordering.upper():
            if item not in ops:
                raise ValueError(f"ordering must be a string of {ops}, got {item} in it.")
            if ops[item] is not None:
                self.add_module(item, ops[item])


class Convolution(nn.Sequential):
    """conv (+ adn) with MONAI's child names (convolutions.py:98-171)."""

    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, adn_ordering="NDA", act="PRELU", norm="INSTANCE",
                 dropout=None, bias=True, conv_only=False, is_transposed=False):
        super().__init__()
        pad = (kernel_size - 1) // 2
        if is_transposed:
            conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, strides, pad, output_padding=strides - 1, bias=bias)
        else:
            conv = nn.Conv3d(in_channels, out_channels, kernel_size, strides, pad, bias=bias)
        self.add_module("conv", conv)
        if conv_only or (act is None and norm is None and dropout is None):
            return
        self.add_module("adn", ADN(adn_ordering, out_channels, act, norm, spatial_dims, dropout))


class ResidualUnit(nn.Module):
    """convolutions.py:255-329."""

    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, subunits=2, adn_ordering="NDA", act="PRELU",
                 norm="INSTANCE", dropout=None, bias=True, last_conv_only=False):
        super().__init__()
        self.conv = nn.Sequential()
        self.residual = nn.Identity()
        sc, ss = in_channels, strides
        subunits = max(1, subunits)
        for su in range(subunits):
            self.conv.add_module(f"unit{su:d}", Convolution(spatial_dims, sc, out_channels, strides=ss, kernel_size=kernel_size,
                                                            adn_ordering=adn_ordering, act=act, norm=norm, dropout=dropout, bias=bias,
                                                            conv_only=last_conv_only and su == subunits - 1))
            sc, ss = out_channels, 1
        if np.prod(strides) != 1 or in_channels != out_channels:
            rk, rp = (kernel_size, (kernel_size - 1) // 2) if np.prod(strides) != 1 else (1, 0)
            self.residual = nn.Conv3d(in_channels, out_channels, rk, strides, rp, bias=bias)


class SkipConnection(nn.Module):
    def __init__(self, submodule, dim: int = 1, mode: str = "cat") -> None:
        super().__init__()
        self.submodule, self.dim, self.mode = submodule, dim, mode


class SequentialWIthModalities(nn.Sequential):
    pass


class UNet(nn.Module):
    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, channels: Sequence[int], strides: Sequence[int], kernel_size=3,
                 up_kernel_size=3, num_res_units: int = 0, act="PRELU", norm_down="INSTANCE", norm_up="INSTANCE", dropout: float = 0.0,
                 bias: bool = True, adn_ordering: str = "NDA", dimensions=None, freeze_encoder: bool = False) -> None:
        super().__init__()
        if len(channels) < 2:
            raise ValueError("the length of `channels` should be no less than 2.")
        delta = len(strides) - (len(channels) - 1)
        if delta < 0:
            raise ValueError("the length of `strides` should equal to `len(channels) - 1`.")
        if delta > 0:
            warnings.warn(f"`len(strides) > len(channels) - 1`, the last {delta} values of strides will not be used.")
        if dimensions is not None:
            spatial_dims = dimensions
        if spatial_dims != 3 or isinstance(kernel_size, (list, tuple)) or isinstance(up_kernel_size, (list, tuple)):
            raise NotImplementedError("only spatial_dims=3 with scalar kernel sizes is reproduced")
        self.dimensions, self.in_channels, self.out_channels = spatial_dims, in_channels, out_channels
        self.channels, self.strides, self.kernel_size, self.up_kernel_size = channels, strides, kernel_size, up_kernel_size
        self.num_res_units, self.act, self.norm_down, self.norm_up = num_res_units, act, norm_down, norm_up
        self.dropout, self.bias, self.adn_ordering = dropout, bias, adn_ordering

        def down(cin, cout, s):
            if num_res_units > 0:
                return ResidualUnit(3, cin, cout, strides=s, kernel_size=kernel_size, subunits=num_res_units, act=act, norm=norm_down,
                                    dropout=dropout, bias=bias, adn_ordering=adn_ordering)
            return Convolution(3, cin, cout, strides=s, kernel_size=kernel_size, act=act, norm=norm_down, dropout=dropout, bias=bias,
                               adn_ordering=adn_ordering)

        def up(cin, cout, s, is_top):
            conv = Convolution(3, cin, cout, strides=s, kernel_size=up_kernel_size, act=act, norm=norm_up, dropout=dropout, bias=bias,
                               conv_only=is_top and num_res_units == 0, is_transposed=True, adn_ordering=adn_ordering)
            if num_res_units > 0:
                ru = ResidualUnit(3, cout, cout, strides=1, kernel_size=kernel_size, subunits=1, act=act, norm=norm_up, dropout=dropout,
                                  bias=bias, last_conv_only=is_top, adn_ordering=adn_ordering)
                conv = SequentialWIthModalities(conv, ru)
            return conv

        def block(inc, outc, chans, strs, is_top):
            c, s = chans[0], strs[0]
            if len(chans) > 2:
                sub, upc = block(c, c, chans[1:], strs[1:], False), c * 2
            else:
                sub, upc = down(c, chans[1], 1), c + chans[1]
                if freeze_encoder:
                    sub.requires_grad_(False)
            d, u = down(inc, c, s), up(upc, outc, s, is_top)
            if freeze_encoder:
                d.requires_grad_(False)
            return SequentialWIthModalities(d, SkipConnection(sub), u)

        self.model = block(in_channels, out_channels, list(channels), list(strides), True)

    @classmethod
    def from_argparse_args(cls, args):
        d = parse_normalization(args.decoder_norm_name, not args.decoder_norm_no_affine, args.num_groups, args.num_styles)
        e = parse_normalization(args.encoder_norm_name, not args.encoder_norm_no_affine, args.num_groups, args.num_styles)
        fs = args.feature_size[0] if isinstance(args.feature_size, (list, tuple)) else args.feature_size
        channels = [fs * 2 ** i for i in range(1, args.num_layers + 1)]      # reference rule (unet.py:218-219), kept as is
        ks = args.kernel_size[0] if isinstance(args.kernel_size, (list, tuple)) and len(args.kernel_size) == 1 else args.kernel_size
        uks = args.up_kernel_size[0] if isinstance(args.up_kernel_size, (list, tuple)) and len(args.up_kernel_size) == 1 else args.up_kernel_size
        return cls(spatial_dims=args.spatial_dims, in_channels=args.in_channels, out_channels=args.out_channels, channels=channels,
                   strides=args.strides, kernel_size=ks, up_kernel_size=uks, num_res_units=args.num_res_units, act=args.activation,
                   norm_down=e, norm_up=d, dropout=args.dropout_rate, bias=not args.no_bias, adn_ordering=args.adn_ordering,
                   freeze_encoder=args.freeze_encoder)

    # ------------------------------------------------------------------------------------------------------------------
    compute_dtype = torch.float32

    def set_compute_dtype(self, dtype):
        """torch.float32 (parity mode) or torch.bfloat16 (bf16 activations / MFMA, fp32 statistics and parameters)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        self.compute_dtype = dtype
        return self

    def forward(self, x: torch.Tensor, modalities=None) -> torch.Tensor:
        """x [B, C, D, H, W] float; modalities None | list[int] | int64 Tensor[B].  Returns fp32 logits [B, out, D, H, W]
        (unet.py:351-353; every grid size must be even wherever a stride-2 layer halves it, as in the reference's configs)."""
        if not x.is_cuda:
            raise RuntimeError("UNet (MI355X path) needs a HIP device tensor; there is no CPU fallback")
        cond = any(isinstance(m, _ConditionalInstanceNorm) for m in self.modules())
        if cond and modalities is None:
            raise ValueError("Modalities must be passed to the forward step when a norm type is 'instance_cond'.")
        styles = styles_to_device(modalities, x.device, x.shape[0], styles_limit(self)) if modalities is not None else None
        ops.begin_forward(self.parameters())      # statistics-pool lifetime: hip/ops.py::_ZeroPool
        x = x.float().contiguous()
        if self.in_channels > 4:      # (up to 4 channels the first convolution reads the NCDHW image itself)
            y = _run_block(self.model, HF.image_rows(x, self.compute_dtype), styles)
        else:
            y = _run_block(self.model, None, styles, image=x, dtype=self.compute_dtype)
        return HF.to_ncdhw(y)


def _conv_layer(conv: nn.Module, x, image=None, dtype=None):
    """nn.Conv3d / nn.ConvTranspose3d container -> channels-last result (convolutions.py:115-139): kernel 3 or 1, stride 1 or 2."""
    stride = conv.stride[0]
    k = conv.kernel_size[0]
    if any(v != stride for v in conv.stride) or any(v != k for v in conv.kernel_size) or stride not in (1, 2) or k not in (1, 3):
        raise NotImplementedError(f"UNet conv kernel {conv.kernel_size} stride {conv.stride}")
    if isinstance(conv, nn.ConvTranspose3d):
        if k != 3:
            raise NotImplementedError("transposed convolution with kernel != 3")
        if stride == 2:
            fine = tuple(2 * v for v in x.shape[1:4])
            x = HF.upsample2_zero(x, fine)                      # zero insertion, then the stride-1 kernel on the transposed pack
        y = HF.conv3_transposed_weight(x, conv.weight)
    else:
        if image is not None:                                   # first layer: straight from the NCDHW fp32 image
            y = HF.conv3_thin(image, conv.weight, dtype) if k == 3 else None
            if y is None:
                raise NotImplementedError("1x1x1 convolution on the raw image")
        elif k == 3:
            y = HF.conv3(x, conv.weight)
        else:
            y = HF.conv1(x, conv.weight)
        if stride == 2:
            if any(v % 2 for v in y.shape[1:4]):
                raise NotImplementedError("stride-2 layer on an odd grid")
            y = HF.subsample2(y)
    return HF.rowbias(y, conv.bias)


def _run_convolution(m: "Convolution", x, styles, image=None, dtype=None):
    y = _conv_layer(m.conv, x, image, dtype)
    if hasattr(m, "adn"):
        for name, child in m.adn.named_children():
            if name == "N":
                y = apply_norm(child, y, styles)
            elif name == "A":
                if isinstance(child, nn.PReLU):
                    y = HF.prelu(y, child.weight)
                elif isinstance(child, nn.GELU):
                    y = HF.gelu(y)
                elif isinstance(child, nn.LeakyReLU):
                    y = HF.leaky_relu(y, child.negative_slope)
                else:
                    y = HF.leaky_relu(y, 0.0)                # ReLU
            elif name == "D":       # MONAI ADN, dropout_dim 1: nn.Dropout (the counter-based mask of miseg_dropout, identity in eval mode)
                y = HF.dropout(y, child.p, child.training)
    return y


def _run_residual_unit(m: "ResidualUnit", x, styles, image=None, dtype=None):
    if image is not None:
        res_in = None
    elif x.requires_grad:
        x, res_in = HF.fork(x)
    else:
        res_in = x
    cx = x
    for i, unit in enumerate(m.conv.children()):
        cx = _run_convolution(unit, cx, styles, image if i == 0 else None, dtype)
    if isinstance(m.residual, nn.Identity):
        if image is not None:
            raise NotImplementedError("identity residual on the raw image")
        res = res_in
    else:
        res = _conv_layer(m.residual, res_in, image, dtype)
    return HF.add(cx, res)


def _run_block(m, x, styles, image=None, dtype=None):
    """recursive Sequential(down, SkipConnection(sub), up) of unet.py:169-205."""
    if isinstance(m, ResidualUnit):
        return _run_residual_unit(m, x, styles, image, dtype)
    if isinstance(m, Convolution):
        return _run_convolution(m, x, styles, image, dtype)
    if isinstance(m, SkipConnection):
        xa, xs = HF.fork(x) if x.requires_grad else (x, x)
        return HF.cat_channels(xs, _run_block(m.submodule, xa, styles))          # cat([x, submodule(x)], C) (simplelayers.py:37-38)
    if isinstance(m, nn.Sequential):
        for i, child in enumerate(m.children()):
            x = _run_block(child, x, styles, image if i == 0 else None, dtype)
        return x
    raise NotImplementedError(type(m))


Unet = UNet
