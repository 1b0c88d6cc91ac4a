"""micro-benchmark of the fused window-attention core (stage shapes of the headline config)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops

def run(dims, heads, C, ws, ss, dtype, iters=10):
    qkv = torch.randn(1, *dims, 3 * C, device="cuda").to(dtype)
    qb = torch.randn(3 * C, device="cuda") * 0.3
    tab = torch.randn(2197, heads, device="cuda") * 0.5
    scale = (C // heads) ** -0.5
    out, lse = ops.winattn_fwd(qkv, qb, tab, heads, ws, ss, 7, scale)
    g = torch.randn_like(out)
    dqb, dt = torch.zeros_like(qb), torch.zeros_like(tab)
    def t(fn):      # hipGraph-captured loop: eager launches of ~60 us kernels measure the host
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(iters): fn()
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / iters * 1e6)
        return sorted(ts)[2]
    f = t(lambda: ops.winattn_fwd(qkv, qb, tab, heads, ws, ss, 7, scale))
    b = t(lambda: ops.winattn_bwd(qkv, out, lse, g, qb, tab, heads, ws, ss, 7, scale, dqb, dt))
    msg = f"dims {dims} heads {heads} C {C} shift {ss} {str(dtype)[6:]}: fwd {f:8.1f} us  bwd {b:8.1f} us"
    if "drop" in sys.argv:      # attn_drop = 0.1: the same kernels with the mask drawn inside (round 3; the query-lane kernels before)
        ops.begin_step()
        key = ops.DROP.next_key(qkv.device)
        outd, lsed = ops.winattn_fwd(qkv, qb, tab, heads, ws, ss, 7, scale, drop=(0.1, key))
        fd = t(lambda: ops.winattn_fwd(qkv, qb, tab, heads, ws, ss, 7, scale, drop=(0.1, key)))
        bd = t(lambda: ops.winattn_bwd(qkv, outd, lsed, g, qb, tab, heads, ws, ss, 7, scale, dqb, dt, drop=(0.1, key)))
        msg += f" | attn_drop 0.1: fwd {fd:8.1f} us  bwd {bd:8.1f} us"
    print(msg, flush=True)

for dt in (torch.bfloat16,):
    run((48, 48, 48), 3, 48, (7, 7, 7), (0, 0, 0), dt)
    run((48, 48, 48), 3, 48, (7, 7, 7), (3, 3, 3), dt)
    run((24, 24, 24), 6, 96, (7, 7, 7), (0, 0, 0), dt)
    run((24, 24, 24), 6, 96, (7, 7, 7), (3, 3, 3), dt)
    run((12, 12, 12), 12, 192, (7, 7, 7), (3, 3, 3), dt)
    run((6, 6, 6), 24, 384, (6, 6, 6), (0, 0, 0), dt)
