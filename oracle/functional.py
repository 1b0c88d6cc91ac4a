"""CPU oracle: functional restatement of the reference's hot-path blocks (plain PyTorch fp32).

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product path under
``mi-seg_amd/`` never does (it fails loudly when the HIP library is missing).

Every function works on a flat ``state_dict``-style mapping ``sd`` (name -> tensor) laid out
exactly like the reference's checkpoints, so oracle, product and reference fixtures share one
set of tensors.  Backward is torch autograd through these restated forwards.  Line numbers cite
files under /root/reference.

Pinning: checked against golden vectors produced by running the reference's own modules
(oracle/tools/make_golden.py).  MONAI-owned arithmetic (MLPBlock, SABlock, perceptron patch
embedding, losses, sliding window) is restated from MONAI 1.1.0's public API and is *unpinned by
the reference* -- the fixtures for those paths come from the stand-in in oracle/tools.
"""
from __future__ import annotations

import itertools
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

EPS = 1e-5


class _ContiguousGrad(torch.autograd.Function):
    """torch 2.10's CPU instance_norm backward is wrong for a non-contiguous grad_output (gradcheck
    fails); every instance norm here hands its backward a contiguous gradient.  Values are untouched."""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def _inorm(x, weight=None, bias=None, eps=EPS):
    return _ContiguousGrad.apply(F.instance_norm(x, weight=weight, bias=bias, eps=eps))


# --------------------------------------------------------------------------------------
# norms  (networks/norms/conditional_instance_norm.py, networks/layers/utils.py:22-50)
# --------------------------------------------------------------------------------------
def _check_styles(x, styles):
    # conditional_instance_norm.py:40-47
    if styles is None or not isinstance(styles, (list, torch.Tensor)) or len(styles) != x.shape[0]:
        raise ValueError("Expected number of styles as batch size.")


def cond_instance_norm(x, styles, weights, biases, eps: float = EPS):
    """x [B,C,*spatial]; per-sample affine row picked by styles[i]  (conditional_instance_norm.py:59-60).
    Biased variance, no running stats (track_running_stats=False -> train == eval)."""
    _check_styles(x, styles)
    outs = []
    for i in range(x.shape[0]):
        s = int(styles[i])
        outs.append(_inorm(x[i:i + 1], weights[s], biases[s], eps)[0])
    return torch.stack(outs)


def norm_channels_first(sd, prefix: str, kind: str, x, modalities, num_styles: int = 2):
    """Apply the norm module stored under ``prefix`` to an NC* tensor.
    kind: 'instance_cond' | 'instance' | 'layer'  (networks/norms/utils.py:1-16)."""
    if kind == "instance_cond":
        if modalities is None:
            raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is "
                             "'instance_cond'.")
        w = [sd[f"{prefix}norms.{s}.weight"] for s in range(num_styles)]
        b = [sd[f"{prefix}norms.{s}.bias"] for s in range(num_styles)]
        return cond_instance_norm(x, modalities, w, b)
    if kind == "instance":
        return _inorm(x, sd.get(f"{prefix}weight"), sd.get(f"{prefix}bias"))
    if kind == "layer":
        xl = x.movedim(1, -1)
        xl = F.layer_norm(xl, (xl.shape[-1],), sd.get(f"{prefix}weight"), sd.get(f"{prefix}bias"), EPS)
        return xl.movedim(-1, 1)
    raise ValueError(f"Normalization {kind} not implemented. Please chose another model.")


def norm_channels_last(sd, prefix, kind, x, modalities, num_styles=2):
    """Swin / ViT blocks hold NDHWC (or NLC) tensors and rearrange around every non-layer norm
    (swin_transformer_block.py:103-112, transformer_block.py:82-90)."""
    if kind == "layer":
        return F.layer_norm(x, (x.shape[-1],), sd.get(f"{prefix}weight"), sd.get(f"{prefix}bias"), EPS)
    return norm_channels_first(sd, prefix, kind, x.movedim(-1, 1), modalities, num_styles).movedim(1, -1)


# --------------------------------------------------------------------------------------
# Swin helpers  (networks/utils/swin_utils.py)
# --------------------------------------------------------------------------------------
def get_window_size(x_size, window_size, shift_size=None):
    """swin_utils.py:80-104: clamp the window to the grid, force shift 0 on clamped axes."""
    ws = list(window_size)
    ss = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            ws[i] = x_size[i]
            if ss is not None:
                ss[i] = 0
    return tuple(ws) if ss is None else (tuple(ws), tuple(ss))


def window_partition(x, ws):
    """swin_utils.py:57-72: [B,D,H,W,C] -> [B*nW, wd*wh*ww, C]; windows row-major, tokens row-major."""
    b, d, h, w, c = x.shape
    x = x.view(b, d // ws[0], ws[0], h // ws[1], ws[1], w // ws[2], ws[2], c)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, ws[0] * ws[1] * ws[2], c)


def window_reverse(windows, ws, dims):
    """swin_utils.py:26-38."""
    b, d, h, w = dims
    x = windows.view(b, d // ws[0], h // ws[1], w // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(b, d, h, w, -1)


def compute_mask(dims, ws, ss):
    """swin_utils.py:107-143: 27 region labels -> [nW, n, n] of {0, -100}."""
    d, h, w = dims
    img = torch.zeros((1, d, h, w, 1))
    cnt = 0
    for sd_ in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for sh in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for sw in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img[:, sd_, sh, sw, :] = cnt
                cnt += 1
    mw = window_partition(img, ws).squeeze(-1)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def relative_position_index(ws=(7, 7, 7)):
    """window_attention.py:58-72,90: idx = (dd+6)*169 + (dh+6)*13 + (dw+6) for a 7^3 window."""
    coords = torch.stack(torch.meshgrid(*[torch.arange(s) for s in ws], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws[0] - 1
    rel[:, :, 1] += ws[1] - 1
    rel[:, :, 2] += ws[2] - 1
    rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
    rel[:, :, 1] *= 2 * ws[2] - 1
    return rel.sum(-1)


def window_attention(sd, prefix, x, mask, num_heads):
    """window_attention.py:99-122.  x [nW*B, n, C]; q scaled before QK^T (:103); bias from the 7^3
    index sliced [:n,:n] (:105-107, the 6^3-stage quirk); mask added per window (:110-113)."""
    b, n, c = x.shape
    hd = c // num_heads
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(b, n, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = sd[prefix + "relative_position_index"][:n, :n].reshape(-1)
    bias = sd[prefix + "relative_position_bias_table"][idx].reshape(n, n, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = attn.view(b // nw, nw, num_heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, num_heads, n, n)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(out, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


def mlp_block(sd, prefix, x):
    """MONAI MLPBlock: linear1 -> exact GELU -> linear2 (dropouts p=0).  Unpinned by the reference."""
    h = F.gelu(F.linear(x, sd[prefix + "linear1.weight"], sd[prefix + "linear1.bias"]))
    return F.linear(h, sd[prefix + "linear2.weight"], sd[prefix + "linear2.bias"])


def swin_block(sd, prefix, x, mask_matrix, modalities, num_heads, window_size, shift_size, norm_kind):
    """swin_transformer_block.py:99-174 (part1), :176-205 (part2), :241-252 (residuals).  x is NDHWC."""
    shortcut = x
    b, d, h, w, c = x.shape
    y = norm_channels_last(sd, prefix + "norm1.", norm_kind, x, modalities)
    ws, ss = get_window_size((d, h, w), window_size, shift_size)
    pd = (ws[0] - d % ws[0]) % ws[0]
    ph = (ws[1] - h % ws[1]) % ws[1]
    pw = (ws[2] - w % ws[2]) % ws[2]
    y = F.pad(y, (0, 0, 0, pw, 0, ph, 0, pd))             # zero pad AFTER the norm, at the high end
    _, dp, hp, wp, _ = y.shape
    if any(i > 0 for i in ss):
        y = torch.roll(y, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        attn_mask = mask_matrix
    else:
        attn_mask = None                                     # padded zero tokens attend un-masked
    win = window_partition(y, ws)
    win = window_attention(sd, prefix + "attn.", win, attn_mask, num_heads)
    y = window_reverse(win.view(-1, *(ws + (c,))), ws, (b, dp, hp, wp))
    if any(i > 0 for i in ss):
        y = torch.roll(y, shifts=ss, dims=(1, 2, 3))
    if pd > 0 or ph > 0 or pw > 0:
        y = y[:, :d, :h, :w, :].contiguous()
    x = shortcut + y
    z = norm_channels_last(sd, prefix + "norm2.", norm_kind, x, modalities)
    return x + mlp_block(sd, prefix + "mlp.", z)


MERGE_OFFSETS_V1 = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1))
"""patch_merging.py:120-127: the v0.9 slice list; slots 5 and 6 duplicate slots 2 and 3."""
MERGE_OFFSETS_V2 = tuple(itertools.product(range(2), range(2), range(2)))
"""patch_merging.py:69-71."""


def patch_merging(sd, prefix, x, modalities, norm_kind, mode="merging"):
    """patch_merging.py:109-143 / :62-103.  x NDHWC -> [B, D/2, H/2, W/2, 2C]."""
    b, d, h, w, c = x.shape
    if (d % 2) or (h % 2) or (w % 2):
        x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2, 0, d % 2))
    offs = MERGE_OFFSETS_V1 if mode == "merging" else MERGE_OFFSETS_V2
    x = torch.cat([x[:, i::2, j::2, k::2, :] for (i, j, k) in offs], -1)
    x = norm_channels_last(sd, prefix + "norm.", norm_kind, x, modalities)
    return F.linear(x, sd[prefix + "reduction.weight"])


def proj_out(x, normalize, norm_kind):
    """swin_transformer.py:121-145: affine-less norm of the returned feature maps (NCDHW)."""
    if not normalize:
        return x
    if norm_kind == "layer":
        xl = x.movedim(1, -1)
        return F.layer_norm(xl, (xl.shape[-1],)).movedim(-1, 1)
    if norm_kind in ("instance", "instance_cond"):
        return _inorm(x)
    return x


def swin_transformer(sd, prefix, x, modalities, cfg):
    """swin_transformer.py:147-159 + BasicLayer.forward :228-258.  Returns the 5 feature maps (NCDHW)."""
    kind, normalize = cfg["vit_norm"], cfg["normalize"]
    ps = 2
    _, _, d, h, w = x.shape
    if w % ps:
        x = F.pad(x, (0, ps - w % ps))
    if h % ps:
        x = F.pad(x, (0, 0, 0, ps - h % ps))
    if d % ps:
        x = F.pad(x, (0, 0, 0, 0, 0, ps - d % ps))
    x0 = F.conv3d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"], stride=ps)
    outs = [proj_out(x0, normalize, kind)]
    cur = x0
    window, shift = (7, 7, 7), (3, 3, 3)
    for li in range(4):
        lp = f"{prefix}layers{li + 1}.0."
        b, c, d, h, w = cur.shape
        ws, ss = get_window_size((d, h, w), window, shift)
        y = cur.permute(0, 2, 3, 4, 1).contiguous()
        dp = -(-d // ws[0]) * ws[0]
        hp = -(-h // ws[1]) * ws[1]
        wp = -(-w // ws[2]) * ws[2]
        # the reference rebuilds the mask every forward (swin_transformer.py:237); with a clamped
        # window the shift is 0 and the mask is never used.
        mask = compute_mask((dp, hp, wp), ws, ss) if any(s > 0 for s in ss) else None
        for bi in range(cfg["depths"][li]):
            y = swin_block(sd, f"{lp}blocks.{bi}.", y, mask, modalities, cfg["num_heads"][li], window,
                           (0, 0, 0) if bi % 2 == 0 else shift, kind)
        y = patch_merging(sd, lp + "downsample.", y, modalities, kind, cfg["downsample"])
        cur = y.permute(0, 4, 1, 2, 3).contiguous()
        outs.append(proj_out(cur, normalize, kind))
    return outs


# --------------------------------------------------------------------------------------
# UNETR CNN blocks  (networks/blocks/dynunet_block.py, unetr_block.py)
# --------------------------------------------------------------------------------------
def leaky(x):
    return F.leaky_relu(x, 0.01)


def unet_res_block(sd, prefix, x, modalities, norm_kind, stride=1):
    """dynunet_block.py:100-126.  Bias-free 3^3 convs (:304), padding (k-s+1)//2 (:329-340);
    1^3 shortcut conv + norm3 when channels differ (:82-98)."""
    out = F.conv3d(x, sd[prefix + "conv1.conv.weight"], None, stride=stride, padding=1)
    out = leaky(norm_channels_first(sd, prefix + "norm1.", norm_kind, out, modalities))
    out = F.conv3d(out, sd[prefix + "conv2.conv.weight"], None, padding=1)
    out = norm_channels_first(sd, prefix + "norm2.", norm_kind, out, modalities)
    res = x
    if prefix + "conv3.conv.weight" in sd:
        res = F.conv3d(x, sd[prefix + "conv3.conv.weight"], None, stride=stride)
        res = norm_channels_first(sd, prefix + "norm3.", norm_kind, res, modalities)
    return leaky(out + res)


def unet_basic_block(sd, prefix, x, modalities, norm_kind, stride=1):
    """dynunet_block.py:185-201."""
    out = F.conv3d(x, sd[prefix + "conv1.conv.weight"], None, stride=stride, padding=1)
    out = leaky(norm_channels_first(sd, prefix + "norm1.", norm_kind, out, modalities))
    out = F.conv3d(out, sd[prefix + "conv2.conv.weight"], None, padding=1)
    return leaky(norm_channels_first(sd, prefix + "norm2.", norm_kind, out, modalities))


def _block(sd, prefix, x, modalities, norm_kind, res_block=True):
    f = unet_res_block if res_block else unet_basic_block
    return f(sd, prefix, x, modalities, norm_kind)


def unetr_up_block(sd, prefix, x, skip, modalities, norm_kind, res_block=True):
    """unetr_block.py:80-85: ConvTranspose3d k2 s2 (no bias) -> cat([up, skip], C) -> block."""
    up = F.conv_transpose3d(x, sd[prefix + "transp_conv.conv.weight"], None, stride=2)
    return _block(sd, prefix + "conv_block.", torch.cat((up, skip), dim=1), modalities, norm_kind, res_block)


def unetr_pr_up_block(sd, prefix, x, modalities, norm_kind, num_layer, conv_block=True, res_block=True):
    """unetr_block.py:203-213."""
    x = F.conv_transpose3d(x, sd[prefix + "transp_conv_init.conv.weight"], None, stride=2)
    for i in range(num_layer):
        if conv_block:
            x = F.conv_transpose3d(x, sd[f"{prefix}blocks.{i}.0.conv.weight"], None, stride=2)
            x = _block(sd, f"{prefix}blocks.{i}.1.", x, modalities, norm_kind, res_block)
        else:
            x = F.conv_transpose3d(x, sd[f"{prefix}blocks.{i}.conv.weight"], None, stride=2)
    return x


def out_block(sd, prefix, x):
    """dynunet_block.py:273-292: 1^3 conv with bias."""
    return F.conv3d(x, sd[prefix + "conv.conv.weight"], sd[prefix + "conv.conv.bias"])


# --------------------------------------------------------------------------------------
# ViT pieces (networks/nets/vit.py, networks/blocks/transformer_block.py + MONAI SABlock)
# --------------------------------------------------------------------------------------
def sa_block(sd, prefix, x, num_heads):
    """MONAI SABlock ("b h (qkv l d) -> qkv b l h d"); unpinned by the reference."""
    b, n, c = x.shape
    hd = c // num_heads
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(b, n, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = ((q @ k.transpose(-2, -1)) * (hd ** -0.5)).softmax(dim=-1)
    out = (att @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(out, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def transformer_block(sd, prefix, x, modalities, num_heads, norm_kind):
    """transformer_block.py:76-110.  x [B, L, C]; cond-norm1d runs over (B, C, L)."""
    x = x + sa_block(sd, prefix + "attn.", norm_channels_last(sd, prefix + "norm1.", norm_kind, x, modalities),
                     num_heads)
    return x + mlp_block(sd, prefix + "mlp.", norm_channels_last(sd, prefix + "norm2.", norm_kind, x, modalities))


def patch_embedding_block(sd, prefix, x, patch, pos_embed):
    """MONAI PatchEmbeddingBlock (vendored copy at patch_embedding.py:32-123)."""
    if pos_embed == "perceptron":
        b, c, H, W, D = x.shape
        p = patch
        # "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)"
        t = x.view(b, c, H // p, p, W // p, p, D // p, p).permute(0, 2, 4, 6, 3, 5, 7, 1)
        t = t.reshape(b, (H // p) * (W // p) * (D // p), p * p * p * c)
        e = F.linear(t, sd[prefix + "patch_embeddings.1.weight"], sd[prefix + "patch_embeddings.1.bias"])
    else:
        e = F.conv3d(x, sd[prefix + "patch_embeddings.weight"], sd[prefix + "patch_embeddings.bias"], stride=patch)
        e = e.flatten(2).transpose(-1, -2)
    return e + sd[prefix + "position_embeddings"]
