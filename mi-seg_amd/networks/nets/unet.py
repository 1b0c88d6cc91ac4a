"""MONAI-style residual UNet with modality-aware ADN blocks (reference networks/nets/unet.py, blocks/convolutions.py,
blocks/acti_norm.py).

BASELINE config 1 is "plumbing, no GPU": constructor, ``from_argparse_args`` (incl. the ``fs * 2**i, i = 1..num_layers`` channel rule
of unet.py:218-219) and the state_dict layout are reproduced here so checkpoints and the model factory work unchanged; its
arithmetic is pinned on the CPU oracle (oracle/nets.py::unet_forward, tests/test_oracle_golden.py::test_unet).  The strided 3x3x3 /
k3-s2 transposed convolutions have no HIP kernel yet, so ``forward`` raises rather than falling back to PyTorch ops."""
import warnings
from typing import Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from ..layers.utils import get_norm_layer
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
            if str(act).lower() != "prelu":
                raise NotImplementedError(f"UNet activation '{act}' (only prelu is reproduced)")
            ops["A"] = nn.PReLU()
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

    def forward(self, x: torch.Tensor, modalities=None) -> torch.Tensor:
        raise NotImplementedError("UNet forward on MI355X needs the strided 3x3x3 / k3-s2 transposed-conv kernels (not built yet); the CPU "
                                  "plumbing config is served by oracle/nets.py::unet_forward in the tests. No PyTorch fallback is taken.")


Unet = UNet
