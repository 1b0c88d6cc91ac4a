"""CPU oracle: whole-network forwards over a flat state_dict (test infrastructure only).

Restates reference networks/nets/{swin_unetr,unetr,unet}.py forward wiring; backward is torch
autograd.  ``cfg`` dictionaries are produced by the ``*_cfg`` helpers below.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import functional as OF


def swin_unetr_cfg(feature_size=48, num_heads=(3, 6, 12, 24), depths=(2, 2, 2, 2), normalize=True,
                   downsample="merging", vit_norm="instance_cond", encoder_norm="instance_cond",
                   decoder_norm="instance"):
    return dict(feature_size=feature_size, num_heads=tuple(num_heads), depths=tuple(depths), normalize=normalize,
                downsample=downsample, vit_norm=vit_norm, encoder_norm=encoder_norm, decoder_norm=decoder_norm)


def swin_unetr_forward(sd, x, modalities, cfg):
    """reference networks/nets/swin_unetr.py:353-366."""
    hs = OF.swin_transformer(sd, "swinViT.", x, modalities, cfg)
    en, dn = cfg["encoder_norm"], cfg["decoder_norm"]
    if en == "instance_cond" and modalities is None:
        raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
    enc0 = OF.unet_res_block(sd, "encoder1.layer.", x, modalities, en)
    enc1 = OF.unet_res_block(sd, "encoder2.layer.", hs[0], modalities, en)
    enc2 = OF.unet_res_block(sd, "encoder3.layer.", hs[1], modalities, en)
    enc3 = OF.unet_res_block(sd, "encoder4.layer.", hs[2], modalities, en)
    dec4 = OF.unet_res_block(sd, "encoder10.layer.", hs[4], modalities, en)
    dec3 = OF.unetr_up_block(sd, "decoder5.", dec4, hs[3], modalities, dn)
    dec2 = OF.unetr_up_block(sd, "decoder4.", dec3, enc3, modalities, dn)
    dec1 = OF.unetr_up_block(sd, "decoder3.", dec2, enc2, modalities, dn)
    dec0 = OF.unetr_up_block(sd, "decoder2.", dec1, enc1, modalities, dn)
    out = OF.unetr_up_block(sd, "decoder1.", dec0, enc0, modalities, dn)
    return OF.out_block(sd, "out.", out)


def unetr_cfg(img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12,
              pos_embed="perceptron", conv_block=True, res_block=True, vit_norm="instance_cond",
              encoder_norm="instance_cond", decoder_norm="instance", num_layers=12):
    return dict(img_size=tuple(img_size), feature_size=feature_size, hidden_size=hidden_size, mlp_dim=mlp_dim,
                num_heads=num_heads, pos_embed=pos_embed, conv_block=conv_block, res_block=res_block,
                vit_norm=vit_norm, encoder_norm=encoder_norm, decoder_norm=decoder_norm, num_layers=num_layers)


def unetr_forward(sd, x, modalities, cfg):
    """reference networks/nets/unetr.py:254-276 with ViT.forward (vit.py:167-197)."""
    vn, en, dn = cfg["vit_norm"], cfg["encoder_norm"], cfg["decoder_norm"]
    if "instance_cond" in (vn, en, dn) and modalities is None:
        raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
    t = OF.patch_embedding_block(sd, "vit.patch_embedding.", x, 16, cfg["pos_embed"])
    hidden = []
    for i in range(cfg["num_layers"]):
        t = OF.transformer_block(sd, f"vit.blocks.{i}.", t, modalities, cfg["num_heads"], vn)
        hidden.append(t)
    t = OF.norm_channels_last(sd, "vit.norm.", vn, t, modalities)
    feat = tuple(s // 16 for s in cfg["img_size"])

    def proj_feat(z):  # unetr.py:248-252
        return z.view(z.shape[0], *feat, cfg["hidden_size"]).permute(0, 4, 1, 2, 3).contiguous()

    cb, rb = cfg["conv_block"], cfg["res_block"]
    enc1 = OF._block(sd, "encoder1.layer.", x, modalities, en, rb)
    enc2 = OF.unetr_pr_up_block(sd, "encoder2.", proj_feat(hidden[3]), modalities, en, 2, cb, rb)
    enc3 = OF.unetr_pr_up_block(sd, "encoder3.", proj_feat(hidden[6]), modalities, en, 1, cb, rb)
    enc4 = OF.unetr_pr_up_block(sd, "encoder4.", proj_feat(hidden[9]), modalities, en, 0, cb, rb)
    dec4 = proj_feat(t)
    dec3 = OF.unetr_up_block(sd, "decoder5.", dec4, enc4, modalities, dn, rb)
    dec2 = OF.unetr_up_block(sd, "decoder4.", dec3, enc3, modalities, dn, rb)
    dec1 = OF.unetr_up_block(sd, "decoder3.", dec2, enc2, modalities, dn, rb)
    out = OF.unetr_up_block(sd, "decoder2.", dec1, enc1, modalities, dn, rb)
    return OF.out_block(sd, "out.", out)


# ------------------------------------------------------------------------------------------
# MONAI-style residual UNet (reference networks/nets/unet.py, blocks/convolutions.py, acti_norm.py)
# ------------------------------------------------------------------------------------------
def unet_cfg(channels=(32, 64, 128, 256), strides=(2, 2, 2), num_res_units=2, norm_down="instance",
             norm_up="instance", adn_ordering="NDA", act="prelu"):
    return dict(channels=tuple(channels), strides=tuple(strides), num_res_units=num_res_units,
                norm_down=norm_down, norm_up=norm_up, adn_ordering=adn_ordering, act=act)


def _adn(sd, prefix, x, modalities, norm_kind, ordering):
    """acti_norm.py:104-110: modules in `ordering`; D is Dropout(p=0) -> identity; A is PReLU."""
    for item in ordering.upper():
        if item == "N":
            x = OF.norm_channels_first(sd, prefix + "N.", norm_kind, x, modalities)
        elif item == "A":
            x = F.prelu(x, sd[prefix + "A.weight"])
    return x


def _convolution(sd, prefix, x, modalities, norm_kind, ordering, stride, transposed=False):
    """convolutions.py:173-179: conv (+bias) then ADN unless conv_only (no `adn.` keys)."""
    w, b = sd[prefix + "conv.weight"], sd.get(prefix + "conv.bias")
    if transposed:  # unet.py up layer: k3 s2 p1 output_padding=s-1 (convolutions.py:131-133)
        x = F.conv_transpose3d(x, w, b, stride=stride, padding=1, output_padding=stride - 1)
    else:
        x = F.conv3d(x, w, b, stride=stride, padding=1)
    if any(k.startswith(prefix + "adn.") for k in sd):
        x = _adn(sd, prefix + "adn.", x, modalities, norm_kind, ordering)
    return x


def _residual_unit(sd, prefix, x, modalities, norm_kind, ordering, stride, subunits):
    """convolutions.py:323-329; residual conv is k3 (strided) or k1 (channel change only) (:311-320)."""
    res = x
    if prefix + "residual.weight" in sd:
        w = sd[prefix + "residual.weight"]
        res = F.conv3d(x, w, sd.get(prefix + "residual.bias"), stride=stride, padding=1 if w.shape[-1] == 3 else 0)
    cx = x
    for su in range(max(1, subunits)):
        cx = _convolution(sd, f"{prefix}conv.unit{su}.", cx, modalities, norm_kind, ordering, stride if su == 0 else 1)
    return cx + res


def unet_forward(sd, x, modalities, cfg):
    """unet.py:169-205,351-353: recursive Sequential(down, SkipConnection(sub), up)."""
    ch, st, nru, od = cfg["channels"], cfg["strides"], cfg["num_res_units"], cfg["adn_ordering"]

    def down(prefix, z, stride):
        if nru > 0:
            return _residual_unit(sd, prefix, z, modalities, cfg["norm_down"], od, stride, nru)
        return _convolution(sd, prefix, z, modalities, cfg["norm_down"], od, stride)

    def up(prefix, z, stride):
        if nru > 0:
            z = _convolution(sd, prefix + "0.", z, modalities, cfg["norm_up"], od, stride, transposed=True)
            return _residual_unit(sd, prefix + "1.", z, modalities, cfg["norm_up"], od, 1, 1)
        return _convolution(sd, prefix, z, modalities, cfg["norm_up"], od, stride, transposed=True)

    def block(prefix, z, level):
        z = down(prefix + "0.", z, st[level])
        if level < len(ch) - 2:
            sub = block(prefix + "1.submodule.", z, level + 1)
        else:
            sub = down(prefix + "1.submodule.", z, 1)          # bottom layer (unet.py:276-284)
        z = torch.cat([z, sub], dim=1)                          # SkipConnection "cat" (simplelayers.py:37-38)
        return up(prefix + "2.", z, st[level])

    return block("model.", x, 0)
