"""SwinTransformerBlock (reference networks/blocks/swin_transformer_block.py:24-252) on channels-last activations."""
from typing import Sequence, Tuple, Union

import torch
import torch.nn as nn
from torch.utils import checkpoint

from ...hip import functional as HF
from ..layers.utils import apply_norm, apply_norm_fork, get_norm_layer, norm_fold_spec
from ..utils.swin_utils import get_window_size
from .mlp import MLPBlock as Mlp
from .window_attention import WindowAttention

__all__ = ["SwinTransformerBlock"]


def norm_spec_with_shape(norm_type, dim):
    """layer norm needs normalized_shape injected (reference :70-75); returns a fresh spec."""
    name = norm_type[0] if isinstance(norm_type, tuple) else norm_type
    if name == "layer":
        args = dict(norm_type[1]) if isinstance(norm_type, tuple) else {}
        args["normalized_shape"] = dim
        return (name, args)
    return norm_type


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim: int, num_heads: int, window_size: Sequence[int], shift_size: Sequence[int], mlp_ratio: float = 4.0,
                 qkv_bias: bool = True, drop: float = 0.0, attn_drop: float = 0.0, drop_path: float = 0.0, act_layer: str = "GELU",
                 use_checkpoint: bool = False, norm_type: Union[Tuple, str] = "layer") -> None:
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        self.window_size = tuple(window_size)
        self.shift_size = tuple(shift_size)
        self.mlp_ratio = mlp_ratio
        self.use_checkpoint = use_checkpoint  # reference :241-252: torch.utils.checkpoint around the two parts of the block; here around the block
        self.norm_type = norm_type[0] if isinstance(norm_type, tuple) else norm_type
        spec = norm_spec_with_shape(norm_type, dim)
        self.norm1 = get_norm_layer(name=spec, spatial_dims=len(self.window_size), channels=dim)
        self.norm2 = get_norm_layer(name=spec, spatial_dims=len(self.window_size), channels=dim)
        self.attn = WindowAttention(dim, window_size=self.window_size, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop,
                                    proj_drop=drop)
        self.drop_path_rate = float(drop_path)
        self.drop_path = nn.Identity()           # (a parameter-free placeholder with the reference's attribute name; the rate is applied below)
        self.mlp = Mlp(hidden_size=dim, mlp_dim=int(dim * mlp_ratio), act=act_layer, dropout_rate=drop, dropout_mode="swin")

    def forward(self, x, styles=None):
        """x [B, D, H, W, C].  part1 (:99-174) + residual, part2 (:176-205) + residual (:241-252).
        use_checkpoint (reference :241-252 checkpoints part1 and part2): the block's activations are dropped after the forward pass and the
        block is run again when the backward pass reaches it - same kernels on the same inputs, bit-identical activations.  The dropout masks
        are counter-based (hip/ops.py::_DropState): the second run draws the keys of the first (its call counter is rewound)."""
        if self.use_checkpoint and torch.is_grad_enabled() and x.requires_grad:
            from ...hip import ops
            first = []

            def run(x_, styles_):
                if not first:
                    first.append(ops.DROP.calls)
                    return self._forward(x_, styles_)
                keep, ops.DROP.calls = ops.DROP.calls, first[0]      # the recomputation: the same dropout keys as the first run
                try:
                    return self._forward(x_, styles_)
                finally:
                    ops.DROP.calls = keep
            # (preserve_rng_state=False: no torch RNG is drawn in here, and reading the generator state is illegal under hipGraph capture)
            return checkpoint.checkpoint(run, x, styles, use_reentrant=False, preserve_rng_state=False)
        return self._forward(x, styles)

    def _forward(self, x, styles=None):
        _, d, h, w, _ = x.shape
        window, shift = get_window_size((d, h, w), self.window_size, self.shift_size)
        inst = self.norm_type.startswith("instance")    # the GEMM that feeds an instance norm also produces its statistics
        plain = not (self.training and (self.drop_path_rate > 0.0 or self.attn.proj_drop > 0.0 or self.mlp.dropout_rate > 0.0))
        if inst and plain:
            # one sample, bf16, the tall-skinny GEMM path (stages 1 - 2 of the headline net): norm1's apply pass rides in the qkv GEMM's operand
            # load and norm2's in the MLP's first product; backward, the norms' sums ride in the data-gradient epilogues (HF._NormLinear / _NormMlp)
            spec = norm_fold_spec(self.norm1, styles)
            r = HF.norm_linear(x, *spec, self.attn.qkv.weight, self.attn.qkv.bias, fork=True) if spec is not None else None
            if r is not None:
                qkv, xs = r
                x = self.attn(None, window, shift, res=xs, want_stat=inst, qkv=qkv)
                spec2 = norm_fold_spec(self.norm2, styles)
                y = HF.norm_mlp(x, *spec2, self.mlp.linear1.weight, self.mlp.linear1.bias, self.mlp.linear2.weight, self.mlp.linear2.bias, want_stat=inst)
                if y is not None:
                    return y
                xn, xs = apply_norm_fork(self.norm2, x, styles)
                return self.mlp(xn, res=xs, want_stat=inst)
        xn, xs = apply_norm_fork(self.norm1, x, styles)
        if self.drop_path_rate > 0.0 and self.training:                 # stochastic depth (:247,:251): x + drop_path(branch)
            x = HF.add(xs, HF.drop_path(self.attn(xn, window, shift), self.drop_path_rate))
            xn, xs = apply_norm_fork(self.norm2, x, styles)
            return HF.add(xs, HF.drop_path(self.mlp(xn), self.drop_path_rate))
        x = self.attn(xn, window, shift, res=xs, want_stat=inst)        # x + attn(norm1(x)): add in the proj epilogue
        xn, xs = apply_norm_fork(self.norm2, x, styles)
        return self.mlp(xn, res=xs, want_stat=inst)                     # x + mlp(norm2(x)): add in the fc2 epilogue
