"""fused MLP kernels against the two-GEMM path they replace, per stage shape and token count (hipGraph-captured loops)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops, lib as L

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

dt = torch.bfloat16
for M, C, H in ((110592, 48, 192), (442368, 48, 192), (13824, 96, 384), (27648, 96, 384), (55296, 96, 384)):
    x, res, dy = (torch.randn(1, M, C, device="cuda").to(dt) for _ in range(3))
    w1, w2 = (torch.randn(H, C, device="cuda") / C ** 0.5).to(dt), (torch.randn(C, H, device="cuda") / H ** 0.5).to(dt)
    b1, b2 = torch.randn(H, device="cuda") / 4, torch.randn(C, device="cuda") / 4
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    pre = torch.empty(1, M, H, device="cuda", dtype=dt)
    def unf_f():
        a = ops.gemm_nt(x, w1, b1, act=L.ACT_GELU, preact_out=pre)
        return ops.gemm_nt(a, w2, b2, res=res)
    def unf_b():
        dh = ops.gemm_nt(dy, w2t, gelu_grad_of=pre)
        return ops.gemm_nt(dh, w1t)
    fused = ops.mlp_fused(x, H)
    ff = f"{t(lambda: ops.mlp_fwd(x, w1, b1, w2, b2, res=res)):6.1f}" if fused else "   n/a"
    fb = f"{t(lambda: ops.mlp_bwd(x, dy, w1, b1, w2t, w1t)):6.1f}" if fused else "   n/a"
    print(f"M {M:7d} C {C:3d}: forward fused {ff} us / two GEMMs {t(unf_f):6.1f} us   backward fused {fb} us / two GEMMs {t(unf_b):6.1f} us", flush=True)
