"""SABlock with MONAI 1.1.0's parameter names (qkv / out_proj), used by TransformerBlock (reference
transformer_block.py:59).  Global multi-head attention over the L tokens of a sample = the fused window-attention core with
one window covering the whole token grid and no relative-position bias."""
import torch.nn as nn

from ...hip import functional as HF


class SABlock(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, dropout_rate: float = 0.0, qkv_bias: bool = False) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden size should be divisible by num_heads.")
        self.dropout_rate = float(dropout_rate)     # MONAI SABlock: drop_weights on the probabilities, drop_output after out_proj
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=qkv_bias)
        self.head_dim = hidden_size // num_heads
        self.scale = self.head_dim ** -0.5

    def forward(self, x, grid, res=None):
        """x [B, L, C] with L = prod(grid) <= 384.  res: the caller's residual (`x + attn(norm(x))`), added in out_proj's epilogue."""
        b, l, c = x.shape
        qkv = HF.linear(x, self.qkv.weight, self.qkv.bias)          # "b h (qkv l d)": q | k | v blocks, head-major inside
        qkv5 = qkv.view(b, grid[0], grid[1], grid[2], 3 * c)
        o = HF.window_attention(qkv5, self.qkv.bias, None, self.num_heads, grid, (0, 0, 0), 1, self.scale, self.dropout_rate, self.training)
        if self.dropout_rate > 0.0 and self.training:      # drop_output sits between out_proj and the residual add
            y = HF.dropout(HF.linear(o.view(b, l, c), self.out_proj.weight, self.out_proj.bias), self.dropout_rate, self.training)
            return HF.add(res, y) if res is not None else y
        return HF.linear(o.view(b, l, c), self.out_proj.weight, self.out_proj.bias, res=res)
