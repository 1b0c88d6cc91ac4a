"""Which kernels are not bitwise reproducible run to run?  Wraps every tensor-returning function of hip/ops.py, checksums its tensor
arguments and results (exact integer sums of the raw bits) over two identical eager forward+backward steps and reports the calls whose
INPUTS agree between the runs while their OUTPUTS do not (the sources of run-to-run noise; everything downstream differs anyway).

    python scripts/find_nondeterminism.py [f32|bf16] [fs] [size] [arena]
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from mi_seg_amd.hip import ops  # noqa: E402
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR  # noqa: E402
from mi_seg_amd.networks.norms.utils import parse_normalization  # noqa: E402
from mi_seg_amd.utils.detfill import det_input, fill_module_  # noqa: E402

LOG = []
SKIP = {"rows", "begin_step", "wgrad_side", "join_wgrad", "pop_gemm_stat", "winattn_params", "flush_tn_reduces", "flush_gemm_tn", "flush_colsums",
        "flush_conv_wgrads"}


def csum(t):
    if t is None or not isinstance(t, torch.Tensor) or not t.is_cuda:
        return None
    c = t.detach().contiguous()
    w = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[c.element_size()]
    return int(c.view(w).to(torch.int64).sum())


def flat(x):
    if isinstance(x, torch.Tensor):
        yield x
    elif isinstance(x, (list, tuple)):
        for v in x:
            yield from flat(v)


def wrap(name, fn):
    def f(*a, **k):
        ins = tuple(csum(t) for t in flat(list(a) + list(k.values())))
        shapes = tuple(tuple(t.shape) for t in flat(list(a)))[:2]
        out = fn(*a, **k)
        torch.cuda.synchronize()
        LOG.append((name, shapes, ins, tuple(csum(t) for t in flat(out))))
        return out
    return f


for n, v in list(vars(ops).items()):
    if isinstance(v, types.FunctionType) and not n.startswith("_") and n not in SKIP:
        setattr(ops, n, wrap(n, v))


def main():
    dt = torch.float32 if (sys.argv[1] if len(sys.argv) > 1 else "f32") == "f32" else torch.bfloat16
    fs = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    use_arena = len(sys.argv) > 4 and sys.argv[4] == "arena"
    cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
    m = SwinUNETR((size,) * 3, 1, 6, feature_size=fs, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst)
    fill_module_(m)
    m = m.cuda().set_compute_dtype(dt)
    x = det_input(1234, (1, 1, size, size, size)).cuda()
    g = det_input(4321, (1, 6, size, size, size)).cuda()
    params = [p for p in m.parameters() if p.requires_grad]
    arena = None
    if use_arena:
        from mi_seg_amd.runtime.arena import ParamArena
        arena = ParamArena(params, dt)
    runs, grads, logits = [], [], []
    for it in range(3 if use_arena else 2):
        LOG.clear()
        if arena is not None:
            arena.begin_step()
        else:
            ops.begin_step()
            for p in params:
                p.grad = None
        y = m(x, [0])
        y.backward(g)
        if arena is not None:
            arena.publish()
        torch.cuda.synchronize()
        runs.append(list(LOG))
        logits.append(csum(y))
        grads.append({k: csum(p.grad) for k, p in m.named_parameters() if p.grad is not None})
    a, b = runs[-2], runs[-1]
    print(f"{len(a)} / {len(b)} wrapped calls; logits bitwise equal: {logits[-2] == logits[-1]}")
    bad = {}
    for i, (ca, cb) in enumerate(zip(a, b)):
        if ca[0] != cb[0]:
            print("call sequences diverge at", i, ca[0], cb[0])
            break
        if ca[2] == cb[2] and ca[3] != cb[3]:
            bad.setdefault((ca[0], ca[1]), []).append(i)
    for (n, sh), idx in bad.items():
        print(f"  SOURCE of noise: {n} {sh}: calls {idx[:6]}{'...' if len(idx) > 6 else ''} ({len(idx)} calls)")
    diff = [k for k in grads[-1] if grads[-1][k] != grads[-2][k]]
    print(f"parameter gradients differing bitwise: {len(diff)} / {len(grads[-1])}")
    for k in diff[:40]:
        print("   ", k)


if __name__ == "__main__":
    main()
