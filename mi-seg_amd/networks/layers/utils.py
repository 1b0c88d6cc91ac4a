"""Norm-layer lookup with the reference's contract (networks/layers/utils.py:22-50, factories.py:219-257) and the
channels-last application used by every block of this package."""
from typing import Optional, Tuple, Union

import torch.nn as nn

from ...hip import functional as HF
from ...hip import lib as L
from ..norms.conditional_instance_norm import (ConditionalInstanceNorm1d, ConditionalInstanceNorm2d, ConditionalInstanceNorm3d,
                                               _ConditionalInstanceNorm)

_INSTANCE = (nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d)
_COND = (ConditionalInstanceNorm1d, ConditionalInstanceNorm2d, ConditionalInstanceNorm3d)


def split_args(args):
    if isinstance(args, str):
        return args, {}
    name_obj, name_args = args
    if not (isinstance(name_obj, str) or callable(name_obj)) or not isinstance(name_args, dict):
        raise TypeError("Layer specifiers must be single strings or pairs of the form (name/object-types, argument dict)")
    return name_obj, name_args


def get_norm_layer(name: Union[Tuple, str], spatial_dims: Optional[int] = 1, channels: Optional[int] = 1):
    """``Norm[name, spatial_dims](**args)`` for the norm families the HIP path implements."""
    if name == "":
        return nn.Identity()
    norm_name, norm_args = split_args(name)
    kw = dict(norm_args)
    key = norm_name.lower() if isinstance(norm_name, str) else norm_name
    if key == "instance_cond":
        kw.setdefault("num_features", channels)
        return _COND[spatial_dims - 1](**kw)
    if key == "instance":
        kw.setdefault("num_features", channels)
        return _INSTANCE[spatial_dims - 1](**kw)
    if key == "layer":
        kw.setdefault("normalized_shape", channels)
        return nn.LayerNorm(**kw)
    if key == "batch":                                   # factories.py:240-243
        kw.setdefault("num_features", channels)
        return (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)[spatial_dims - 1](**kw)
    if key == "group":                                   # factories.py:246-248 (num_groups comes with the spec: norms/utils.py:13-14)
        kw.setdefault("num_channels", channels)
        return nn.GroupNorm(**kw)
    if key in ("localresponse", "syncbatch", "instance_nvfuser"):
        raise NotImplementedError(f"normalisation '{key}' is not implemented by the MI355X path "
                                  "(supported: instance_cond, instance, layer, batch, group)")
    raise ValueError(f"Unsupported option '{norm_name}'")


def apply_norm_fork(norm: nn.Module, x, styles=None):
    """(norm(x), x) for the `x + f(norm(x))` pattern: with an instance norm the two branch gradients are summed inside the
    norm-backward kernel; other norms fall back to the explicit fork."""
    if isinstance(norm, _ConditionalInstanceNorm):
        if styles is None:
            raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
        return HF.instance_norm(x, norm.style_params(), styles[0], styles[1], eps=norm.eps, fork=True)
    if isinstance(norm, _INSTANCE):
        return HF.instance_norm(x, [(norm.weight, norm.bias)] if norm.affine else None, None, None, eps=norm.eps, fork=True)
    xa, xs = HF.fork(x)
    return apply_norm(norm, xa, styles), xs


def norm_fold_spec(norm: nn.Module, styles=None):
    """(params, styles_dev, styles_host, eps) of a (conditional) instance norm for the consumers that fold its apply pass into their operand
    load (HF.norm_linear / HF.norm_mlp), or None for the other norm kinds"""
    if isinstance(norm, _ConditionalInstanceNorm):
        if styles is None:
            raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
        return norm.style_params(), styles[0], styles[1], norm.eps
    if isinstance(norm, _INSTANCE):
        return ([(norm.weight, norm.bias)] if norm.affine else None), None, None, norm.eps
    return None


def apply_res_norm_pair(norm_a: nn.Module, xa, norm_b: nn.Module, xb, styles=None, slope=0.01, stat_a=None, out=None, w1=None):
    """LeakyReLU(norm_a(xa) + norm_b(xb)) in one apply pass (HF.res_norm_pair) where both norms are instance norms of the same kind (any
    size: the joint backward is ONE register-resident launch up to 2048 rows per sample, two chunked ones above); None where that does not apply.
    w1: xb is a one-channel image and norm_b sees conv1x1x1(xb; w1), which is then never materialised (one sample, bf16 / fp32 rows)."""
    if (xa.shape != xb.shape if w1 is None else (xa.shape[:-1] != xb.shape[:-1] or xb.shape[-1] != 1 or xa.shape[0] != 1
                                                   or xb.requires_grad or xa.shape[-1] % 8 != 0)):
        return None
    if isinstance(norm_a, _ConditionalInstanceNorm) and isinstance(norm_b, _ConditionalInstanceNorm):
        if styles is None:
            raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
        if norm_a.num_styles != norm_b.num_styles:
            return None
        return HF.res_norm_pair(xa, xb, norm_a.style_params(), norm_b.style_params(), styles[0], styles[1], slope=slope, eps_a=norm_a.eps,
                                eps_b=norm_b.eps, stat_a=stat_a, out=out, w1=w1)
    if isinstance(norm_a, _INSTANCE) and isinstance(norm_b, _INSTANCE) and norm_a.affine == norm_b.affine and norm_a.eps == norm_b.eps:
        pa = [(norm_a.weight, norm_a.bias)] if norm_a.affine else None
        pb = [(norm_b.weight, norm_b.bias)] if norm_b.affine else None
        return HF.res_norm_pair(xa, xb, pa, pb, None, None, slope=slope, eps_a=norm_a.eps, eps_b=norm_b.eps, stat_a=stat_a, out=out, w1=w1)
    return None


def stat_request(norm: nn.Module):
    """what to pass as HF.conv3(..., want_stat=) when `norm` is applied to the convolution's output by apply_norm right away: "defer" lets a
    split convolution of a small stage leave its partial slabs to the (conditional) instance norm's one launch (HF.ops.PendingSlabs); the
    other norm kinds read the finished tensor"""
    return "defer" if isinstance(norm, (_ConditionalInstanceNorm,) + _INSTANCE) else True


def apply_norm(norm: nn.Module, x, styles=None, res=None, act=L.ACT_NONE, slope=0.01, stat=None, out=None):
    """Apply a norm *module* (used as a parameter container) to a channels-last tensor through the HIP kernels.
    ``styles`` is the (device int32 tensor, host tuple) pair from styles_to_device; ``stat``: instance-norm statistics of x that the
    producer of x already computed (HF.conv3(..., want_stat=True)), or None."""
    if isinstance(norm, _ConditionalInstanceNorm):
        if styles is None:
            raise ValueError("Modalities must be passed to the forward step when encoder_norm_type is 'instance_cond'.")
        return HF.instance_norm(x, norm.style_params(), styles[0], styles[1], res=res, act=act, slope=slope, eps=norm.eps, stat=stat, out=out)
    if isinstance(norm, _INSTANCE):
        params = [(norm.weight, norm.bias)] if norm.affine else None
        return HF.instance_norm(x, params, None, None, res=res, act=act, slope=slope, eps=norm.eps, stat=stat, out=out)
    if isinstance(norm, (nn.GroupNorm, nn.modules.batchnorm._BatchNorm)):
        if isinstance(norm, nn.GroupNorm):
            y = HF.group_norm(x, norm.weight, norm.bias, norm.num_groups, norm.eps)
        else:
            y = HF.batch_norm(x, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.training, norm.momentum, norm.eps,
                              norm.num_batches_tracked)
        # (no fused residual / activation for these kinds: composed from the stand-alone kernels)
        if res is not None:
            y = HF.add(y, res)
        if act == L.ACT_LEAKY:
            y = HF.leaky_relu(y, slope)
        if out is not None:
            from ...hip import ops as _ops
            _ops.copy2d(y, out)
            y = out
        return y
    if isinstance(norm, nn.LayerNorm):
        if res is not None or act != L.ACT_NONE:
            raise NotImplementedError("LayerNorm with fused residual / activation")
        return HF.layer_norm(x, norm.weight, norm.bias, norm.eps)
    if isinstance(norm, nn.Identity):
        return x
    raise NotImplementedError(type(norm))
