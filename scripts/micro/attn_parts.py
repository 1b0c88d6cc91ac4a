"""stage-1 window-attention backward (1029 workgroups) with parts of its work switched off: how much of the launch are the bias-table
gradient's LDS atomics and the shift mask?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops


def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            fn()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / iters * 1e6)
    return sorted(ts)[2]


heads, C, dims = 3, 48, (48, 48, 48)
qkv = torch.randn(1, *dims, 3 * C, device="cuda").to(torch.bfloat16)
qb = torch.randn(3 * C, device="cuda") * 0.3
tab = torch.randn(2197, heads, device="cuda") * 0.5
for ss in ((0, 0, 0), (3, 3, 3)):
    out, lse = ops.winattn_fwd(qkv, qb, tab, heads, (7, 7, 7), ss, 7, 0.25)
    g = torch.randn_like(out)
    dqb, dt = torch.zeros_like(qb), torch.zeros_like(tab)
    full = t(lambda: ops.winattn_bwd(qkv, out, lse, g, qb, tab, heads, (7, 7, 7), ss, 7, 0.25, dqb, dt))
    nodt = t(lambda: ops.winattn_bwd(qkv, out, lse, g, qb, tab, heads, (7, 7, 7), ss, 7, 0.25, dqb, None))
    print(f"shift {ss}: backward {full:6.1f} us, without the table gradient (no bin atomics) {nodt:6.1f} us", flush=True)
