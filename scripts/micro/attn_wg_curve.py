"""window-attention backward / forward time against the number of workgroups (windows x heads) at a fixed window (7^3, head_dim 16):
how long is a workgroup alone on its CU, with a neighbour, and where do the rounds of the stage-1 launch (1029 workgroups) end?"""
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


heads, C = 3, 48
for dims in [(7, 7, 7), (14, 14, 14), (21, 21, 21), (28, 28, 28), (28, 28, 35), (35, 35, 35), (35, 35, 42), (35, 42, 42), (42, 42, 42), (42, 42, 49), (42, 49, 49), (48, 48, 48), (49, 49, 56)]:
    qkv = torch.randn(1, *dims, 3 * C, device="cuda").to(torch.bfloat16)
    qb = torch.randn(3 * C, device="cuda") * 0.3
    tab = torch.randn(2197, heads, device="cuda") * 0.5
    out, lse = ops.winattn_fwd(qkv, qb, tab, heads, (7, 7, 7), (3, 3, 3), 7, 0.25)
    g = torch.randn_like(out)
    dqb, dt = torch.zeros_like(qb), torch.zeros_like(tab)
    f = t(lambda: ops.winattn_fwd(qkv, qb, tab, heads, (7, 7, 7), (3, 3, 3), 7, 0.25))
    b = t(lambda: ops.winattn_bwd(qkv, out, lse, g, qb, tab, heads, (7, 7, 7), (3, 3, 3), 7, 0.25, dqb, dt))
    nw = 1
    for d in dims:
        nw *= -(-d // 7)
    print(f"{nw * heads:5d} workgroups ({nw} windows x {heads} heads): fwd {f:7.1f} us  bwd {b:7.1f} us", flush=True)
