"""model_from_argparse_args (reference networks/utils/utils.py:17-65): name -> constructor, optional checkpoint loading."""
import warnings

import torch

from ..nets.swin_unetr import SwinUNETR
from ..nets.unet import UNet
from ..nets.unetr import UNETR

__all__ = ["model_from_argparse_args"]


def model_from_argparse_args(args):
    name = args.model_name
    if name == "unetr":
        model = UNETR.from_argparse_args(args)
    elif name == "unet":
        model = UNet.from_argparse_args(args)
    elif name in ("swin_unetr", "pre_swin_unetr"):
        model = SwinUNETR.from_argparse_args(args)
        if name == "pre_swin_unetr":
            # MONAI self-supervised Swin-ViT weights: strip "module.", fc1/fc2 -> linear1/linear2 (reference :28-37)
            state = torch.load(args.pre_swin, map_location="cpu")["state_dict"]
            state = {k.replace("module.", "").replace("fc1", "linear1").replace("fc2", "linear2"): v for k, v in state.items()}
            print("Loaded pre-trained Swin-ViT")
            print(model.swinViT.load_state_dict(state, strict=False))
    else:
        raise ValueError("Model {} not implemented. Please chose another model.".format(name))

    if getattr(args, "pretrained", None):
        print("Loading pre-trained weights ...")
        state = torch.load(args.pretrained, map_location="cpu")["state_dict"]
        if "out.conv.conv.weight" in state and state["out.conv.conv.weight"].shape[0] != args.out_channels:
            warnings.warn("Number of out channels of the pre-trained model different from model out_channels, "
                          "skipping loading of output layer.")
            del state["out.conv.conv.weight"], state["out.conv.conv.bias"]
        for key in [k for k in state if "model.2" in k]:     # UNet head (reference :57-62)
            del state[key]
        print(model.load_state_dict(state, strict=False))
    return model
