"""Where does the fp32 parity mode's forward error come from?  (logits 1.27e-6 from the reference's float64 run where
the reference's own fp32 run is 7.6e-7.)  One op class at a time is replaced by torch in FLOAT64 on the device (result rounded to fp32) and the
C2 logits are compared with the float64 fixture: the class whose replacement moves the error is the one that produces it.

    python scripts/debug/f32_error_sources.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from conftest import sample  # noqa: E402
from mi_seg_amd.hip import functional as HF  # noqa: E402
from mi_seg_amd.hip import lib as L  # noqa: E402
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR  # noqa: E402
from mi_seg_amd.networks.norms.utils import parse_normalization  # noqa: E402
from mi_seg_amd.utils.detfill import det_input, fill_module_  # noqa: E402

ORIG = {k: getattr(HF, k) for k in ("conv3", "conv3_thin", "instance_norm", "res_norm_pair", "linear", "mlp", "conv1", "upconv_cat", "window_attention")}


def _place(y, out):
    if out is not None:
        out.copy_(y)
        return out
    return y.contiguous()


def conv3_64(x, weight, want_stat=False, fork=False):
    y = F.conv3d(x.permute(0, 4, 1, 2, 3).double(), weight.double(), padding=1).permute(0, 2, 3, 4, 1).float().contiguous()
    outs = [y] + ([None] if want_stat else []) + ([x] if fork else [])
    return tuple(outs) if len(outs) > 1 else y


def conv3_thin_64(x_ncdhw, weight, dtype):
    return F.conv3d(x_ncdhw.double(), weight.double(), padding=1).permute(0, 2, 3, 4, 1).float().contiguous()


def _norm64(x, params, styles_host, eps):
    B, C = x.shape[0], x.shape[-1]
    xd = x.double().reshape(B, -1, C)
    m = xd.mean(1, keepdim=True)
    v = xd.var(1, unbiased=False, keepdim=True)
    y = (xd - m) / torch.sqrt(v + eps)
    if params is not None:
        st = styles_host if styles_host is not None else [0] * B
        g = torch.stack([params[s][0].double() for s in st])[:, None, :]
        b = torch.stack([params[s][1].double() for s in st])[:, None, :]
        y = y * g + b
    return y.reshape(x.shape)


def instance_norm_64(x, params=None, styles_dev=None, styles_host=None, res=None, act=L.ACT_NONE, slope=0.01, eps=1e-5, fork=False, stat=None, out=None):
    y = _norm64(x, params, styles_host, eps)
    if res is not None:
        y = y + res.double()
    if act == L.ACT_LEAKY:
        y = torch.where(y > 0, y, y * slope)
    y = _place(y.float(), out)
    return (y, x) if fork else y


def res_norm_pair_64(xa, xb, params_a, params_b, styles_dev=None, styles_host=None, slope=0.01, eps_a=1e-5, eps_b=1e-5, stat_a=None, out=None, w1=None):
    if w1 is not None:
        xb = xb.double() * w1.double().reshape(1, 1, 1, 1, -1)
    y = _norm64(xa, params_a, styles_host, eps_a) + _norm64(xb, params_b, styles_host, eps_b)
    return _place(torch.where(y > 0, y, y * slope).float(), out)


def linear_64(x, weight, bias=None, res=None, want_stat=False):
    y = x.double() @ weight.double().reshape(weight.shape[0], -1).t()
    if bias is not None:
        y = y + bias.double()
    if res is not None:
        y = y + res.double()
    return y.float()


def mlp_64(x, w1, b1, w2, b2, res=None, want_stat=False):
    h = F.gelu(x.double() @ w1.double().t() + b1.double())
    y = h @ w2.double().t() + b2.double()
    if res is not None:
        y = y + res.double()
    return y.float()


def conv1_64(x, weight, want_stat=False):
    return (x.double() @ weight.double().reshape(weight.shape[0], -1).t()).float()


PATCHES = {
    "none (the HIP fp32 path)": {},
    "conv 3x3x3 (+ stem)": {"conv3": conv3_64, "conv3_thin": conv3_thin_64},
    "instance norms": {"instance_norm": instance_norm_64, "res_norm_pair": res_norm_pair_64},
    "linears + MLP + 1x1x1 convs": {"linear": linear_64, "mlp": mlp_64, "conv1": conv1_64},
    "conv + norms": {"conv3": conv3_64, "conv3_thin": conv3_thin_64, "instance_norm": instance_norm_64, "res_norm_pair": res_norm_pair_64},
    "conv + norms + linears": {"conv3": conv3_64, "conv3_thin": conv3_thin_64, "instance_norm": instance_norm_64, "res_norm_pair": res_norm_pair_64,
                               "linear": linear_64, "mlp": mlp_64, "conv1": conv1_64},
}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    T = np.load(os.path.join(ROOT, "tests/golden/swin_unetr_c2_truth.npz"))
    R = np.load(os.path.join(ROOT, "tests/golden/swin_unetr_c2.npz"))
    cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
    m = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst)
    fill_module_(m)
    m = m.cuda().set_compute_dtype(torch.float32)
    x = det_input(1234, (1, 1, 96, 96, 96)).cuda()
    truth = T["c2_m0/logits64_samples"]
    print(f"reference fp32 vs float64: {rel(R['c2_m0/logits_samples'], truth):.2e}")
    for name, patch in PATCHES.items():
        for k, v in ORIG.items():
            setattr(HF, k, patch.get(k, v))
        with torch.no_grad():
            y = m(x, [0])
        print(f"{name:36s} logits vs float64 {rel(sample(y).numpy(), truth):.2e}", flush=True)
    for k, v in ORIG.items():
        setattr(HF, k, v)


if __name__ == "__main__":
    main()
