from networks.layers.factories import *  # noqa: F401,F403  (reference's vendored copy)
from networks.layers.factories import Act, Conv, Norm, split_args  # noqa: F401
