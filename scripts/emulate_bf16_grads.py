"""CPU emulation (on the oracle) of where bf16 rounding of GRADIENT tensors hurts the C2 parameter gradients.

modes: clean | bf16 (activations and every gradient tensor rounded to bf16) | mixed (bf16 activations; the gradient tensors
that ENTER an instance-norm backward / ride the residual stream stay fp32, the ones that are matrix-core operands are bf16).
Prints per-parameter relative errors against the clean run, worst first.  Design evidence for DESIGN.md section 3; not a test.

    python scripts/emulate_bf16_grads.py [fs] [size]
"""
import os
import sys
import types

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from oracle import functional as OF  # noqa: E402
from oracle import nets as ON  # noqa: E402
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR  # noqa: E402
from mi_seg_amd.networks.norms.utils import parse_normalization  # noqa: E402
from mi_seg_amd.utils.detfill import ce_cotangent, det_input, det_values  # noqa: E402

COT = [None]
MODE = {"fwd": False, "g_operand": False, "g_norm_in": False}


def r(t):
    return t.bfloat16().float()


class Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return r(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (r(g.contiguous()) if ctx.bwd else g.contiguous()), None, None


def fw(name):
    f = MODE["fwd"]
    return f is True or (isinstance(f, (set, tuple)) and name in f)


def wrap_matmul(fn, name):
    def f(x, w, *a, **k):
        x = Round.apply(x, False, MODE["g_norm_in"])                 # dx written by the data-gradient kernel
        w = Round.apply(w, fw("w"), False)
        return Round.apply(fn(x, w, *a, **k), fw(name), MODE["g_operand"])   # dy read as a matrix-core operand
    return f


Fp = types.SimpleNamespace(**{k: getattr(F, k) for k in dir(F) if not k.startswith("__")})
Fp.conv3d = wrap_matmul(F.conv3d, "conv")
Fp.linear = wrap_matmul(F.linear, "lin")
Fp.conv_transpose3d = wrap_matmul(F.conv_transpose3d, "convt")
Fp.gelu = lambda x: Round.apply(F.gelu(x), fw("gelu"), False)
OF.F = Fp
ON.F = Fp
_inorm0 = OF._inorm
OF._inorm = lambda x, weight=None, bias=None, eps=OF.EPS: Round.apply(_inorm0(x, weight, bias, eps), fw("norm"), MODE["g_norm_in"])


def run(fs, size, mode):
    MODE.update({"clean": dict(fwd=False, g_operand=False, g_norm_in=False), "bf16": dict(fwd=True, g_operand=True, g_norm_in=True),
                 "mixed": dict(fwd=True, g_operand=True, g_norm_in=False), "fwd_only": dict(fwd=True, g_operand=False, g_norm_in=False)}.get(mode) or
                dict(fwd=tuple(mode[2:].split("+")), g_operand=False, g_norm_in=False))
    cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
    with torch.device("meta"):
        m = SwinUNETR((size,) * 3, 1, 6, feature_size=fs, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst)
    sd = {}
    for k, v in m.state_dict().items():
        sd[k] = OF.relative_position_index() if k.endswith("relative_position_index") else torch.from_numpy(det_values(k, v.shape)).requires_grad_(True)
    x = det_input(1234, (1, 1, size, size, size))
    y = ON.swin_unetr_forward(sd, x, [0], ON.swin_unetr_cfg(feature_size=fs))
    y.backward(ce_cotangent(y) if COT[0] is None else COT[0])
    if COT[0] is None and os.environ.get("FIXED_COT"):
        COT[0] = ce_cotangent(y)
    return y.detach(), {k: v.grad for k, v in sd.items() if v.is_floating_point() and v.grad is not None}


def main():
    fs = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    torch.set_num_threads(os.cpu_count())
    y0, g0 = run(fs, size, "clean")
    rms = {k: float(g.norm()) / g.numel() ** 0.5 for k, g in g0.items()}
    med = sorted(rms.values())[len(rms) // 2]
    for mode in sys.argv[3:] or ("bf16", "mixed", "fwd_only"):
        y, g = run(fs, size, mode)
        errs = sorted(((float((g[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)), k) for k in g0 if rms[k] > 1e-3 * med), reverse=True)
        print(f"== {mode}: logits rel err {float((y - y0).norm() / y0.norm()):.3e}; parameters > 5e-2: {sum(e > 5e-2 for e, _ in errs)} / {len(errs)}; "
              f"median {errs[len(errs) // 2][0]:.3e}")
        for e, k in errs[:12]:
            print(f"   {e:.3e}  {k}  ({g0[k].numel()})")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
