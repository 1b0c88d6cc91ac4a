"""SwinUNETR -- drop-in for reference networks/nets/swin_unetr.py (same constructor, from_argparse_args,
forward(x_in, modalities) and state_dict keys); all device arithmetic runs in csrc/libmiseg_hip.so."""
from typing import Sequence, Tuple, Union

import os

import numpy as np
import torch
import torch.nn as nn

from ...hip import functional as HF
from ...hip import ops
from ..blocks.dynunet_block import UnetOutBlock
from ..blocks.patch_merging import PatchMerging, PatchMergingV2
from ..blocks.unetr_block import UnetrBasicBlock, UnetrUpBlock
from ..norms.conditional_instance_norm import styles_limit, styles_to_device
from ..norms.utils import parse_normalization
from .swin_transformer import SwinTransformer, look_up_option

__all__ = ["SwinUNETR", "MERGING_MODE"]

MERGING_MODE = {"merging": PatchMerging, "mergingv2": PatchMergingV2}


def ensure_tuple_rep(x, n):
    if isinstance(x, (list, tuple)):
        if len(x) != n:
            raise ValueError(f"Sequence must have length {n}, got {len(x)}.")
        return tuple(x)
    return (x,) * n


class SwinUNETR(nn.Module):
    def __init__(self, img_size: Union[Sequence[int], int], in_channels: int, out_channels: int, depths: Sequence[int] = (2, 2, 2, 2),
                 num_heads: Sequence[int] = (3, 6, 12, 24), feature_size: int = 24, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 dropout_path_rate: float = 0.0, normalize: bool = True, use_checkpoint: bool = False, spatial_dims: int = 3,
                 downsample="merging", vit_norm_name: Union[Tuple, str] = "layer", decoder_norm_name: Union[Tuple, str] = "instance",
                 encoder_norm_name: Union[Tuple, str] = "instance", freeze_encoder: bool = False) -> None:
        super().__init__()
        if not (spatial_dims == 2 or spatial_dims == 3):
            raise ValueError("spatial dimension should be 2 or 3.")
        if spatial_dims != 3:
            raise NotImplementedError("only spatial_dims=3 is implemented by the MI355X path")
        img_size = ensure_tuple_rep(img_size, spatial_dims)
        patch_size = ensure_tuple_rep(2, spatial_dims)
        window_size = ensure_tuple_rep(7, spatial_dims)
        for m, p in zip(img_size, patch_size):
            for i in range(5):
                if m % np.power(p, i + 1) != 0:
                    raise ValueError("input image size (img_size) should be divisible by stage-wise image resolution.")
        if not (0 <= drop_rate <= 1):
            raise ValueError("dropout rate should be between 0 and 1.")
        if not (0 <= attn_drop_rate <= 1):
            raise ValueError("attention dropout rate should be between 0 and 1.")
        if not (0 <= dropout_path_rate <= 1):
            raise ValueError("drop path rate should be between 0 and 1.")
        if feature_size % 12 != 0:
            raise ValueError("feature_size should be divisible by 12.")
        self.vit_norm_name = vit_norm_name[0] if isinstance(vit_norm_name, tuple) else vit_norm_name
        self.decoder_norm_name = decoder_norm_name[0] if isinstance(decoder_norm_name, tuple) else decoder_norm_name
        self.encoder_norm_name = encoder_norm_name[0] if isinstance(encoder_norm_name, tuple) else encoder_norm_name
        if self.decoder_norm_name == "layer" or self.encoder_norm_name == "layer":
            raise ValueError("Layer normalization not yet implemented for encoder and decoder blocks, please "
                             "select another normalization.")
        self.normalize = normalize
        self.in_channels = in_channels
        self.compute_dtype = torch.float32
        # use_checkpoint (reference swin_transformer_block.py:241-252): every Swin block drops its activations after the forward pass and runs
        # again in the backward pass (networks/blocks/swin_transformer_block.py); off by default - a 288 GB card holds every activation of
        # the 96^3 step at any batch the reference trains with

        self.swinViT = SwinTransformer(
            in_chans=in_channels, embed_dim=feature_size, window_size=window_size, patch_size=patch_size, depths=depths,
            num_heads=num_heads, mlp_ratio=4.0, qkv_bias=True, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
            drop_path_rate=dropout_path_rate, use_checkpoint=use_checkpoint, spatial_dims=spatial_dims,
            downsample=look_up_option(downsample, MERGING_MODE) if isinstance(downsample, str) else downsample,
            norm_type=vit_norm_name)
        fs = feature_size

        def enc(cin, cout):
            return UnetrBasicBlock(spatial_dims=spatial_dims, in_channels=cin, out_channels=cout, kernel_size=3, stride=1,
                                   norm_name=encoder_norm_name, res_block=True)

        def dec(cin, cout):
            return UnetrUpBlock(spatial_dims=spatial_dims, in_channels=cin, out_channels=cout, kernel_size=3, upsample_kernel_size=2,
                                norm_name=decoder_norm_name, res_block=True)

        self.encoder1 = enc(in_channels, fs)
        self.encoder2 = enc(fs, fs)
        self.encoder3 = enc(2 * fs, 2 * fs)
        self.encoder4 = enc(4 * fs, 4 * fs)
        self.encoder10 = enc(16 * fs, 16 * fs)
        self.decoder5 = dec(16 * fs, 8 * fs)
        self.decoder4 = dec(8 * fs, 4 * fs)
        self.decoder3 = dec(4 * fs, 2 * fs)
        self.decoder2 = dec(2 * fs, fs)
        self.decoder1 = dec(fs, fs)
        self.out = UnetOutBlock(spatial_dims=spatial_dims, in_channels=fs, out_channels=out_channels)
        if freeze_encoder:
            for m in (self.swinViT, self.encoder1, self.encoder2, self.encoder3, self.encoder4, self.encoder10):
                m.requires_grad_(False)

    def set_compute_dtype(self, dtype):
        """torch.float32 (parity mode) or torch.bfloat16 (bf16 activations / MFMA, fp32 statistics and parameters)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        self.compute_dtype = dtype
        return self

    @classmethod
    def from_argparse_args(cls, args):
        vit_norm_name = parse_normalization(args.vit_norm_name, not args.vit_norm_no_affine, args.num_groups, args.num_styles)
        decoder_norm_name = parse_normalization(args.decoder_norm_name, not args.decoder_norm_no_affine, args.num_groups, args.num_styles)
        encoder_norm_name = parse_normalization(args.encoder_norm_name, not args.encoder_norm_no_affine, args.num_groups, args.num_styles)
        if len(args.depth_swin_block) == 1:
            depths = (args.depth_swin_block[0],) * 4
        else:
            assert len(args.depth_swin_block) == 4, "The length of depth_swin_block should be 4"
            depths = args.depth_swin_block
        num_heads = tuple(2 ** i * args.num_heads for i in range(0, 4))
        fs = args.feature_size[0] if isinstance(args.feature_size, (list, tuple)) else args.feature_size
        return cls(img_size=(args.roi_x, args.roi_y, args.roi_z), in_channels=args.in_channels, out_channels=args.out_channels,
                   depths=depths, num_heads=num_heads, feature_size=fs, drop_rate=args.dropout_rate,
                   attn_drop_rate=args.attn_drop_rate, dropout_path_rate=args.dropout_path_rate, normalize=not args.no_normalize_swin,
                   use_checkpoint=args.use_checkpoint, spatial_dims=args.spatial_dims, downsample=args.downsample,
                   vit_norm_name=vit_norm_name, encoder_norm_name=encoder_norm_name, decoder_norm_name=decoder_norm_name,
                   freeze_encoder=args.freeze_encoder)

    def load_from(self, weights):
        """MONAI self-supervised Swin-ViT weights (reference swin_unetr.py:303-351); plain-norm checkpoints only."""
        sd = {k.replace("module.", "").replace("fc1", "linear1").replace("fc2", "linear2"): v for k, v in weights["state_dict"].items()}
        # the reference copies an explicit list (patch_embed, every block of layers1-4, the downsample reduction / norm) and raises KeyError
        # when a checkpoint lacks one of them; extra keys of the checkpoint are ignored there as here
        want = [k for k in self.swinViT.state_dict() if k.startswith(("patch_embed.", "layers1.", "layers2.", "layers3.", "layers4."))
                and not k.endswith("relative_position_index")]
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"checkpoint lacks {len(missing)} Swin-ViT entries, e.g. '{missing[0]}'")
        return self.swinViT.load_state_dict({k: sd[k] for k in want}, strict=False)

    # parameters whose gradients are complete once the decoder side has been back-propagated (autograd runs it first): what a
    # data-parallel step can start all-reducing while the encoder / Swin half of the backward pass is still running
    late_backward_prefixes = ("encoder10.", "decoder5.", "decoder4.", "decoder3.", "decoder2.", "decoder1.", "out.")

    # The two image-resolution encoder blocks read the image / the first Swin feature map and feed the last two decoders only.  With
    # `side_branch` they run on a branch stream, taped in front of encoder10: autograd, which walks the tape backwards and runs every
    # node on the stream of its forward, reaches their backward pass - 0.7 ms of full-size kernels - behind encoder10's and runs it
    # BESIDE the small-grid launches of the deeper blocks' backward (the convolution kernels in their background form, see
    # hip/ops.py::_background), and their forward runs beside encoder10 / decoder5..3.  Measured A/B;
    # `MISEG_NO_BRANCH=1` switches it off.
    side_branch = os.environ.get("MISEG_NO_BRANCH") is None
    # Where the training forward forks that branch: "e10" = where it is taped (in front of encoder10, beside encoder10 / decoder5..3);
    # "s0" / "s1" / "s2" = as soon as Swin feature map 0 / 1 / 2 is queued, i.e. beside the Swin stages behind it, on a tape of its own that
    # joins the outer tape in front of encoder10 (HF.inner_tape_join): the backward pass is issued where it always was
    fork_at = os.environ.get("MISEG_FORK_AT", "e10")

    @staticmethod
    def _skip_block(block, inp, styles, shape, channels, dt, **kw):
        """an encoder block whose output is a decoder's skip tensor: it writes straight into the right half of that decoder's
        concat buffer (unetr_block.py:80-85 does torch.cat: a copy of the 96^3 x 48 tensor among others)"""
        dims = tuple(shape[2:5]) if kw.get("image") is not None else tuple(shape[1:4])
        cat, view = HF.concat_buffer((shape[0],) + dims, channels, dt, (inp if inp is not None else kw["image"]).device)
        return HF.tag_concat(block(inp, styles, out_view=view, **kw), cat)

    # split (data-parallel) step: True when the caller leaves `deferred_backward_parameters()` out of the range it all-reduces after the first
    # half (bench.py does): decoder1's two 96^3 weight gradients are then deferred to the branch's backward pass there as well
    split_defers = False

    def deferred_backward_parameters(self):
        """the parameters whose gradients the side branch defers into the second half of the backward pass (adjacent in registration order)"""
        blk = self.decoder1.conv_block
        return [blk.conv1.conv.weight, blk.conv2.conv.weight]

    def late_backward_parameters(self):
        return [p for k, p in self.named_parameters() if k.startswith(self.late_backward_prefixes)]

    def forward(self, x_in, modalities=None, cut=None, on_decoder_done=None):
        """x_in [B, C, D, H, W] float; modalities None | list[int] | int64 Tensor[B].  Returns fp32 logits [B, out, D, H, W].
        cut: optional list; when given, the six tensors that cross from the Swin / encoder side to the decoder side are replaced by
        detached leaves and (original, leaf) pairs are appended, so that `logits.backward(g)` stops at the leaves and
        `torch.autograd.backward([o for o, _ in cut], [l.grad for _, l in cut])` finishes the pass (runtime/graph.py).
        on_decoder_done: optional callable, run by the backward pass once the gradient of the deepest Swin feature map exists - autograd walks
        the tape backwards, so out / decoder1..5 / encoder10 (`late_backward_prefixes`) have been back-propagated by then and nothing else has:
        where a data-parallel step issues its decoder-side grouped launches and starts their all-reduce inside ONE captured step (round 4)."""
        if not x_in.is_cuda:
            raise RuntimeError("SwinUNETR (MI355X path) needs a HIP device tensor; there is no CPU fallback")
        needs = "instance_cond" in (self.vit_norm_name, self.encoder_norm_name, self.decoder_norm_name)
        if needs and modalities is None:
            raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
        styles = styles_to_device(modalities, x_in.device, x_in.shape[0], styles_limit(self)) if modalities is not None else None
        ops.begin_forward(self.parameters())      # statistics-pool lifetime: hip/ops.py::_ZeroPool
        x_in = x_in.float().contiguous()
        dt = self.compute_dtype
        # (bf16 only: in the fp32 parity mode the branch's convolutions are long enough to become the critical path when throttled - 31.2 -> 29.5)
        # (round 4: bench.py's roofline leg times every launch in place, so the branch stays on there too)
        branch = self.side_branch and dt == torch.bfloat16 and torch.is_grad_enabled() and not x_in.requires_grad
        # inference (no tape, so no ordering constraint from the backward pass): the two blocks are forked right behind `layers1` and run
        # beside the deep Swin stages, encoder3 / 4 / 10 and decoder5..3
        infer_branch = (self.side_branch and dt == torch.bfloat16 and not torch.is_grad_enabled()
                        and os.environ.get("MISEG_NO_INFER_BRANCH") is None)
        enc0 = enc1 = None

        def fork_inference(hs0):
            nonlocal enc0, enc1
            side, cur = ops.branch_stream(x_in.device), torch.cuda.current_stream()
            side.wait_stream(cur)
            for t in (hs0, x_in, styles[0] if styles is not None else None):
                if t is not None:
                    t.record_stream(side)
            with torch.cuda.stream(side):
                enc0 = self._skip_block(self.encoder1, None, styles, x_in.shape, self.encoder1.layer.conv2.conv.weight.shape[0], dt, image=x_in, dtype=dt)
                enc1 = self._skip_block(self.encoder2, hs0, styles, hs0.shape, hs0.shape[-1], dt)

        inner, feats = {}, []

        def fork_training(i, feat):
            feats.append(feat)
            if i != {"s0": 0, "s1": 1, "s2": 2}[self.fork_at] or not feats[0].requires_grad:
                return
            side, cur = ops.branch_stream(x_in.device), torch.cuda.current_stream()
            ops._MAIN_STREAM = cur
            side.wait_stream(cur)
            for t in (feats[0], x_in, styles[0] if styles is not None else None):
                if t is not None:
                    t.record_stream(side)
            with torch.cuda.stream(side):
                leaf0 = HF.inner_tape_leaf(feats[0])
                e0 = self._skip_block(self.encoder1, None, styles, x_in.shape, self.encoder1.layer.conv2.conv.weight.shape[0], dt, image=x_in, dtype=dt)
                e1 = self._skip_block(self.encoder2, leaf0, styles, leaf0.shape, leaf0.shape[-1], dt)
                ops.stamp("branch_fwd_end", fine=True)
            inner.update(outs=(e0, e1), outer=(feats[0],), leaves=(leaf0,))

        early = branch and self.fork_at in ("s0", "s1", "s2")
        hs = self.swinViT(x_in, self.normalize, styles, dt, after_stage1=fork_inference if infer_branch else None,
                          on_feature=fork_training if early else None)
        ops.stamp("swin_end", fine=True)
        if not branch and not infer_branch:
            enc0 = self._skip_block(self.encoder1, None, styles, x_in.shape, self.encoder1.layer.conv2.conv.weight.shape[0], dt, image=x_in, dtype=dt)
            enc1 = self._skip_block(self.encoder2, hs[0], styles, hs[0].shape, hs[0].shape[-1], dt)
        enc2 = self._skip_block(self.encoder3, hs[1], styles, hs[1].shape, hs[1].shape[-1], dt)
        enc3 = self._skip_block(self.encoder4, hs[2], styles, hs[2].shape, hs[2].shape[-1], dt)
        h4, h3 = hs[4], hs[3]
        if on_decoder_done is not None and cut is None:
            # (an autograd node at encoder10's input: it runs right behind encoder10's backward pass - a tensor hook on h4 only fires in front of
            # h4's own producer, which autograd reaches after encoder4 / encoder3: 0.3 ms later)
            h4 = HF.backward_mark(h4, on_decoder_done)
        if cut is not None:
            def leaf(t):
                l = t.detach().requires_grad_(True)
                if getattr(t, "_miseg_cat", None) is not None:
                    l._miseg_cat = t._miseg_cat
                cut.append((t, l))
                return l
            if branch:
                h4, h3, enc3, enc2 = (leaf(t) for t in (h4, h3, enc3, enc2))
            else:
                h4, h3, enc3, enc2, enc1, enc0 = (leaf(t) for t in (h4, h3, enc3, enc2, enc1, enc0))
        if branch:
            # taped HERE, in front of encoder10 and decoder5..3: forward, the branch's 0.36 ms of full-size kernels run beside those blocks' small
            # launches; backward, autograd reaches the branch behind encoder10's backward and runs it beside encoder4 / encoder3 / the deep
            # Swin stages (in front of decoder2, i.e. no forward overlap but a longer backward window, measured 138.7; here 143.4; right
            # behind the Swin transformer 143.3)
            side, cur = ops.branch_stream(x_in.device), torch.cuda.current_stream()
            ops._MAIN_STREAM = cur
            if inner:      # forked earlier, on a tape of its own: the backward pass is taped here (nothing is launched)
                with torch.cuda.stream(side):
                    enc0, enc1 = HF.inner_tape_join(inner["outs"], inner["outer"], inner["leaves"])
        if branch and not inner:
            side.wait_stream(cur)
            for t in (hs[0], x_in, styles[0] if styles is not None else None):      # allocated on this stream, read by the branch's kernels
                if t is not None:
                    t.record_stream(side)
            with torch.cuda.stream(side):
                enc0 = self._skip_block(self.encoder1, None, styles, x_in.shape, self.encoder1.layer.conv2.conv.weight.shape[0], dt, image=x_in, dtype=dt)
                enc1 = self._skip_block(self.encoder2, hs[0], styles, hs[0].shape, hs[0].shape[-1], dt)
                ops.stamp("branch_fwd_end", fine=True)
        def mark(t, name):      # measurement aid (MISEG_STEP_STAMPS=1): the device clock when the forward / the backward pass gets here
            if ops.STAMPS is not None and ops.STAMPS_FINE:
                ops.stamp("f:" + name)
                if t.requires_grad:
                    t.register_hook(lambda g, n=name: ops.stamp("b:" + n))
            return t
        for i_, t_ in enumerate(hs):
            mark(t_, f"hs{i_}")
        mark(enc2, "enc2"); mark(enc3, "enc3")
        dec4 = mark(self.encoder10(h4, styles), "dec4")
        dec3 = mark(self.decoder5(dec4, h3, styles), "dec3")
        dec2 = mark(self.decoder4(dec3, enc3, styles), "dec2")
        dec1 = mark(self.decoder3(dec2, enc2, styles), "dec1")
        if infer_branch:
            side, cur = ops.branch_stream(x_in.device), torch.cuda.current_stream()
        if branch or infer_branch:       # join: decoder2 is the first consumer of the branch
            ops.stamp("main_at_fwd_join", fine=True)
            cur.wait_stream(side)
            for t in (enc0, enc1):
                t.record_stream(cur)
                t._miseg_cat.record_stream(cur)
            if cut is not None:
                enc1, enc0 = leaf(enc1), leaf(enc0)
        if branch and cut is not None and not self.split_defers:
            ops.close_branch_deferral(self.parameters())     # split step: the decoder side's gradients are all-reduced right after the first half - nothing of it may wait
        if branch and (cut is None or self.split_defers) and os.environ.get("MISEG_NO_DEFER") is None:
            ops.open_branch_deferral(self.parameters())       # decoder1's two 96^3 weight gradients wait for the branch's backward pass (hip/ops.py::defer_to_branch)
        dec0 = mark(self.decoder2(dec1, enc1, styles), "dec0")
        if branch and cut is not None and self.split_defers == "early" and dec0.requires_grad:
            # split step, first half: the branch stream has no backward work of its own here (the branch's backward pass is in the second
            # half), so decoder1's two deferred weight gradients run on it in background form as soon as decoder1's backward pass has queued
            # them - beside decoder2 .. 5 / encoder10 - and are final when the first half ends: no hole in the early all-reduce range
            dec0.register_hook(lambda g, ps=self.out.parameters: ops.flush_deferred_on_branch(ps()))
        out = mark(self.decoder1(dec0, enc0, styles), "out")
        return self.out(out)
