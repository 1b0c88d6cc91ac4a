import torch.nn as nn


class MLPBlock(nn.Module):
    """MONAI 1.1.0 MLPBlock restated: linear1 -> GELU(erf) -> drop1 -> linear2 -> drop2.
    Parameter names linear1/linear2 are confirmed by reference swin_transformer_block.py:236-239."""

    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0, act="GELU", dropout_mode="vit"):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        mlp_dim = mlp_dim or hidden_size
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)
        if str(act).upper() != "GELU":
            raise NotImplementedError(act)
        self.fn = nn.GELU()
        self.drop1 = nn.Dropout(dropout_rate)
        self.drop2 = self.drop1 if dropout_mode == "swin" else nn.Dropout(dropout_rate)

    def forward(self, x):
        return self.drop2(self.linear2(self.drop1(self.fn(self.linear1(x)))))
