"""miseg-mi355x: MI355X-native (gfx950) training path for MI-Seg's 3D cross-modality segmentation nets.

Drop-in surface (same names/signatures/state_dict keys as matteo-bastico/MI-Seg):
``networks.nets.{SwinUNETR,UNETR,UNet}``, ``networks.utils.utils.model_from_argparse_args``,
``networks.lightning_monai.LitMonai``, ``networks.norms.ConditionalInstanceNorm{1,2,3}d``.
All device arithmetic goes through the C-ABI library ``csrc/libmiseg_hip.so`` (include/miseg_hip.h).
"""
__version__ = "0.1.0"
