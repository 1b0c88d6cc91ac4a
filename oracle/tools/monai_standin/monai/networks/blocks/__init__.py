from .mlp import MLPBlock  # noqa: F401
from .selfattention import SABlock  # noqa: F401
from .convolutions import Convolution  # noqa: F401


def __getattr__(name):  # lazy: the reference's convolutions.py imports us while loading
    if name == "ResidualUnit":
        from networks.blocks.convolutions import ResidualUnit
        return ResidualUnit
    raise AttributeError(name)
