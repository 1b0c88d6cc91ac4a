import torch
import torch.nn as nn
from einops import rearrange


class SABlock(nn.Module):
    """MONAI 1.1.0 SABlock restated (parameter names qkv / out_proj)."""

    def __init__(self, hidden_size, num_heads, dropout_rate=0.0, qkv_bias=False):
        super().__init__()
        if hidden_size % num_heads != 0:
            raise ValueError("hidden size should be divisible by num_heads.")
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=qkv_bias)
        self.drop_output = nn.Dropout(dropout_rate)
        self.drop_weights = nn.Dropout(dropout_rate)
        self.head_dim = hidden_size // num_heads
        self.scale = self.head_dim ** -0.5

    def forward(self, x):
        q, k, v = rearrange(self.qkv(x), "b h (qkv l d) -> qkv b l h d", qkv=3, l=self.num_heads)
        att = (torch.einsum("blxd,blyd->blxy", q, k) * self.scale).softmax(dim=-1)
        att = self.drop_weights(att)
        x = torch.einsum("bhxy,bhyd->bhxd", att, v)
        x = rearrange(x, "b h l d -> b l (h d)")
        return self.drop_output(self.out_proj(x))
