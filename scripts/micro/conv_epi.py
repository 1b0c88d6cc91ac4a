"""the 96^3 conv launches as the C2 step issues them: plain, with the fused residual, with residual + statistics, and into / out of
channel slices of a wider buffer (ld > C) - to see which of the epilogue pieces or strides costs what the in-step durations show."""
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

S = 96
for Cin, Cout in ((48, 48), (96, 48), (48, 96)):
    dt = torch.bfloat16
    x = torch.randn(1, S, S, S, Cin, device="cuda").to(dt)
    w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") / (27 * Cin) ** 0.5
    fwdp, _ = ops.pack_conv3(w, dt)
    res = torch.randn(1, S, S, S, Cout, device="cuda").to(dt)
    wide_in = torch.randn(1, S, S, S, 2 * Cin, device="cuda").to(dt)
    wide_out = torch.empty(1, S, S, S, 2 * Cout, device="cuda", dtype=dt)
    ops.begin_step()
    print(f"{Cin}->{Cout}: plain {t(lambda: ops.conv3_fwd(x, fwdp, Cout)):6.1f}  +res {t(lambda: ops.conv3_fwd(x, fwdp, Cout, res=res)):6.1f}"
          f"  +stat {t(lambda: ops.conv3_fwd(x, fwdp, Cout, want_stat=True)):6.1f}  +res+stat {t(lambda: ops.conv3_fwd(x, fwdp, Cout, res=res, want_stat=True)):6.1f}"
          f"  x slice of 2Cin {t(lambda: ops.conv3_fwd(wide_in[..., :Cin], fwdp, Cout)):6.1f}  out slice of 2Cout {t(lambda: ops.conv3_fwd(x, fwdp, Cout, out=wide_out[..., Cout:])):6.1f}", flush=True)
