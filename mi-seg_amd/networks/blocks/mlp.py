"""MLPBlock with MONAI 1.1.0's parameter names (linear1 / linear2), as used at reference
swin_transformer_block.py:97 and transformer_block.py:58: Linear -> exact GELU -> Dropout -> Linear -> Dropout."""
import torch.nn as nn

from ...hip import functional as HF


class MLPBlock(nn.Module):
    def __init__(self, hidden_size: int, mlp_dim: int, dropout_rate: float = 0.0, act="GELU", dropout_mode="vit"):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        self.dropout_rate = float(dropout_rate)
        if str(act).upper() != "GELU":
            raise NotImplementedError(f"activation {act}")
        mlp_dim = mlp_dim or hidden_size
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)

    def forward(self, x, res=None, want_stat=False):
        """res: optional residual added in the second GEMM's epilogue (the caller's `x + mlp(norm(x))`).
        want_stat: the result feeds an instance norm (see HF.linear)."""
        if self.dropout_rate > 0.0 and self.training:
            # MONAI MLPBlock: drop2(linear2(drop1(gelu(linear1(x))))) - the fused two-GEMM form has no place for the masks
            h = HF.dropout(HF.gelu(HF.linear(x, self.linear1.weight, self.linear1.bias)), self.dropout_rate)
            y = HF.dropout(HF.linear(h, self.linear2.weight, self.linear2.bias), self.dropout_rate)
            return HF.add(res, y) if res is not None else y
        return HF.mlp(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, res, want_stat)
