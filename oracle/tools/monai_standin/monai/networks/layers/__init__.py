import torch
from networks.layers.factories import Act, Conv, Norm, split_args  # reference's vendored copy
from networks.layers.utils import get_act_layer  # reference's vendored copy  # noqa: F401


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(torch.nn.Module):
    """Stochastic depth; identity at p == 0 or eval (MONAI 1.1.0 semantics)."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask
