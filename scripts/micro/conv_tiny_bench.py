"""3x3x3 forward / data gradient on the tiny volumes of the headline net (encoder10 / decoder5): us per launch incl. the slab sum.
Usage: python scripts/micro/conv_tiny_bench.py   (MISEG_HIP_LIB=... for another build)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops
dt = torch.bfloat16
for S, Cin, Cout in [(3, 768, 768), (6, 768, 384), (6, 384, 384), (6, 384, 768), (12, 384, 192), (12, 192, 192)]:
    x = torch.randn(1, S, S, S, Cin, device="cuda").to(dt)
    w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") / (27 * Cin) ** 0.5
    fwdp, _ = ops.pack_conv3(w, dt)
    fn = lambda: ops.conv3_fwd(x, fwdp, Cout)
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20)
    print(f"{S}^3 {Cin}->{Cout}: {sorted(ts)[2] * 1e6:7.1f} us   weights {Cin * Cout * 27 * 2 / 1e6:.1f} MB", flush=True)
