"""the patch-embedding kernels at the headline shape (96^3 x 1 -> 48^3 x 48, bf16 out); MISEG_HIP_LIB=<other build> for an A/B"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops


def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / iters)
    return sorted(ts)[3] * 1e6


x = torch.randn(1, 1, 96, 96, 96, device="cuda")
w, b = torch.randn(48, 1, 2, 2, 2, device="cuda") / 3, torch.randn(48, device="cuda")
gy = torch.randn(1, 48, 48, 48, 48, device="cuda").to(torch.bfloat16)
dw, db = torch.zeros_like(w), torch.zeros_like(b)
print(f"patch embed fwd {t(lambda: ops.patch_embed_fwd(x, w, b, torch.bfloat16)):6.1f} us   bwd {t(lambda: ops.patch_embed_bwd(x, gy, dw, db)):6.1f} us")
