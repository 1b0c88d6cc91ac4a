"""UNETR -- drop-in for reference networks/nets/unetr.py (ViT-B/16 encoder + UNETR CNN decoder)."""
from typing import Sequence, Tuple, Union

import os

import torch
import torch.nn as nn

from ...hip import ops
from ..blocks.dynunet_block import UnetOutBlock
from ..blocks.unetr_block import UnetrBasicBlock, UnetrPrUpBlock, UnetrUpBlock
from ..norms.conditional_instance_norm import styles_limit, styles_to_device
from ..norms.utils import parse_normalization
from .swin_unetr import ensure_tuple_rep
from .vit import ViT

__all__ = ["UNETR"]


class UNETR(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, img_size: Union[Sequence[int], int], feature_size: int = 16, hidden_size: int = 768,
                 mlp_dim: int = 3072, num_heads: int = 12, pos_embed: str = "conv", conv_block: bool = True, res_block: bool = True,
                 dropout_rate: float = 0.0, spatial_dims: int = 3, qkv_bias: bool = False, vit_norm_name: Union[Tuple, str] = "layer",
                 decoder_norm_name: Union[Tuple, str] = "instance", encoder_norm_name: Union[Tuple, str] = "instance",
                 freeze_encoder: bool = False) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        if spatial_dims != 3:
            raise NotImplementedError("only spatial_dims=3 is implemented by the MI355X path")
        self.num_layers = 12
        img_size = ensure_tuple_rep(img_size, spatial_dims)
        self.patch_size = ensure_tuple_rep(16, spatial_dims)
        self.feat_size = tuple(i // p for i, p in zip(img_size, self.patch_size))
        self.hidden_size = hidden_size
        self.classification = False
        self.vit_norm_name = vit_norm_name[0] if isinstance(vit_norm_name, tuple) else vit_norm_name
        self.decoder_norm_name = decoder_norm_name[0] if isinstance(decoder_norm_name, tuple) else decoder_norm_name
        self.encoder_norm_name = encoder_norm_name[0] if isinstance(encoder_norm_name, tuple) else encoder_norm_name
        if self.decoder_norm_name == "layer" or self.encoder_norm_name == "layer":
            raise ValueError("Layer normalization not yet implemented for encoder and decoder blocks, please "
                             "select another normalization.")
        self.compute_dtype = torch.float32
        self.vit = ViT(in_channels=in_channels, img_size=img_size, patch_size=self.patch_size, hidden_size=hidden_size, mlp_dim=mlp_dim,
                       num_layers=self.num_layers, num_heads=num_heads, pos_embed=pos_embed, classification=False, dropout_rate=dropout_rate,
                       spatial_dims=spatial_dims, qkv_bias=qkv_bias, norm_type=vit_norm_name)
        fs = feature_size
        self.encoder1 = UnetrBasicBlock(spatial_dims, in_channels, fs, kernel_size=3, stride=1, norm_name=encoder_norm_name, res_block=res_block)

        def pr(cout, nl):
            return UnetrPrUpBlock(spatial_dims, hidden_size, cout, num_layer=nl, kernel_size=3, stride=1, upsample_kernel_size=2,
                                  norm_name=encoder_norm_name, conv_block=conv_block, res_block=res_block)

        def up(cin, cout):
            return UnetrUpBlock(spatial_dims, cin, cout, kernel_size=3, upsample_kernel_size=2, norm_name=decoder_norm_name, res_block=res_block)

        self.encoder2, self.encoder3, self.encoder4 = pr(fs * 2, 2), pr(fs * 4, 1), pr(fs * 8, 0)
        self.decoder5, self.decoder4, self.decoder3, self.decoder2 = up(hidden_size, fs * 8), up(fs * 8, fs * 4), up(fs * 4, fs * 2), up(fs * 2, fs)
        self.out = UnetOutBlock(spatial_dims=spatial_dims, in_channels=fs, out_channels=out_channels)
        if freeze_encoder:
            for m in (self.vit, self.encoder1, self.encoder2, self.encoder3, self.encoder4):
                m.requires_grad_(False)

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        self.compute_dtype = dtype
        return self

    @classmethod
    def from_argparse_args(cls, args):
        v = parse_normalization(args.vit_norm_name, not args.vit_norm_no_affine, args.num_groups, args.num_styles)
        d = parse_normalization(args.decoder_norm_name, not args.decoder_norm_no_affine, args.num_groups, args.num_styles)
        e = parse_normalization(args.encoder_norm_name, not args.encoder_norm_no_affine, args.num_groups, args.num_styles)
        fs = args.feature_size[0] if isinstance(args.feature_size, (list, tuple)) else args.feature_size
        return cls(in_channels=args.in_channels, out_channels=args.out_channels, img_size=(args.roi_x, args.roi_y, args.roi_z), feature_size=fs,
                   hidden_size=args.hidden_size, mlp_dim=args.mlp_dim, num_heads=args.num_heads, pos_embed=args.pos_embed,
                   conv_block=not args.no_conv_block, res_block=not args.no_res_block, dropout_rate=args.dropout_rate,
                   spatial_dims=args.spatial_dims, qkv_bias=args.qkv_bias, vit_norm_name=v, decoder_norm_name=d, encoder_norm_name=e,
                   freeze_encoder=args.freeze_encoder)

    def proj_feat(self, x):
        """[B, L, hidden] tokens -> channels-last feature map [B, d, h, w, hidden] (reference unetr.py:248-252: a pure view here)."""
        return x.view(x.shape[0], *self.feat_size, self.hidden_size)

    side_branch = os.environ.get("MISEG_NO_BRANCH") is None       # A/B switch: configs[2] 129.6 -> 133.2 patches/s

    def forward(self, x_in, modalities=None):
        if not x_in.is_cuda:
            raise RuntimeError("UNETR (MI355X path) needs a HIP device tensor; there is no CPU fallback")
        if "instance_cond" in (self.vit_norm_name, self.encoder_norm_name, self.decoder_norm_name) and modalities is None:
            raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
        styles = styles_to_device(modalities, x_in.device, x_in.shape[0], styles_limit(self)) if modalities is not None else None
        ops.begin_forward(self.parameters())      # statistics-pool lifetime: hip/ops.py::_ZeroPool
        x_in = x_in.float().contiguous()
        dt = self.compute_dtype
        x, hidden = self.vit(x_in, styles, dt)
        # side branch (see SwinUNETR.side_branch): encoder1 reads the image only and feeds the last decoder only, and autograd reaches its
        # backward pass right in front of the ViT's - 12 blocks of 216-token launches that leave the chip idle.  On the branch stream it runs
        # BESIDE them, in background form.  (Round 4: encoder2-4 on the branch as well - their backward passes beside the ViT's instead of in
        # front of it, correct, 3 more stream edges in each direction - replayed 20 % SLOWER, 210 -> 167 patches/s: DESIGN.md R4.3.)
        branch = (self.side_branch and dt == torch.bfloat16 and torch.is_grad_enabled() and not x_in.requires_grad)
        if branch:
            side, cur = ops.branch_stream(x_in.device), torch.cuda.current_stream()
            side.wait_stream(cur)
            for t in (x_in, styles[0] if styles is not None else None):      # allocated on this stream, read by the branch's kernels
                if t is not None:
                    t.record_stream(side)
            with torch.cuda.stream(side):
                enc1 = self.encoder1(None, styles, image=x_in, dtype=dt)
            # decoder2's two 96^3 weight gradients wait for the branch's backward pass (hip/ops.py::defer_to_branch), as decoder1's do in
            # SwinUNETR.  Round 3 left them inline (0.29 ms launches then: throttled they became the critical path, 133.2 -> 123 .. 132.9); with
            # the narrow-layer weight-gradient kernel they are 45 + 60 us and deferring pays: 204.8 / 207.6 -> 209.2 / 211.5 patches/s
            # (MISEG_UNETR_DEFER=0 keeps them inline)
            (ops.close_branch_deferral if os.environ.get("MISEG_UNETR_DEFER") == "0" else ops.open_branch_deferral)(self.parameters())
        else:
            enc1 = self.encoder1(None, styles, image=x_in, dtype=dt)
        enc2 = self.encoder2(self.proj_feat(hidden[3]), styles)
        enc3 = self.encoder3(self.proj_feat(hidden[6]), styles)
        enc4 = self.encoder4(self.proj_feat(hidden[9]), styles)
        dec3 = self.decoder5(self.proj_feat(x), enc4, styles)
        dec2 = self.decoder4(dec3, enc3, styles)
        dec1 = self.decoder3(dec2, enc2, styles)
        if branch:
            cur.wait_stream(side)
            enc1.record_stream(cur)
        out = self.decoder2(dec1, enc1, styles)
        return self.out(out)
