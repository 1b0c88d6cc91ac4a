"""the three head kernels at the headline shape (96^3 x 48 -> 6 classes, bf16 activations, fp32 logits)"""
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

x = torch.randn(1, 96, 96, 96, 48, device="cuda").to(torch.bfloat16)
w, b = torch.randn(6, 48, 1, 1, 1, device="cuda") / 7, torch.randn(6, device="cuda")
g = torch.randn(1, 6, 96, 96, 96, device="cuda")
dw, db = torch.zeros_like(w), torch.zeros_like(b)
print(f"head fwd {t(lambda: ops.head_fwd(x, w, b)):6.1f} us   bwd (dx + dw + dbias) {t(lambda: ops.head_bwd(x, g, w, dw, db)):6.1f} us")
