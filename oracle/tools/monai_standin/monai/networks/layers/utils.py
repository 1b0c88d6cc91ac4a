from networks.layers.utils import get_act_layer, get_norm_layer  # noqa: F401
