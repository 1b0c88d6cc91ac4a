"""micro-benchmark of the 3x3x3 implicit-GEMM kernels on the headline shapes (used for rocprofv3 --pmc passes too)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops

def run(S, Cin, Cout, dtype, iters=10, what=("fwd", "wgrad")):
    x = torch.randn(1, S, S, S, Cin, device="cuda").to(dtype)
    w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") / (27 * Cin) ** 0.5
    fwdp, bwdp = ops.pack_conv3(w, dtype)
    dy = torch.randn(1, S, S, S, Cout, device="cuda").to(dtype)
    fl = 2.0 * S ** 3 * 27 * Cin * Cout
    def t(fn):      # hipGraph-captured loop: the host launch path does not hide or add to the kernel time
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters): fn()
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):      # median of 7 replays: single replays scatter by +-5 %
            t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / iters)
        return sorted(ts)[3]
    msg = f"{S}^3 {Cin}->{Cout} {str(dtype)[6:]}:"
    if "fwd" in what:
        d = t(lambda: ops.conv3_fwd(x, fwdp, Cout)); msg += f" fwd {d*1e6:7.1f} us {fl/d/1e12:6.1f} TF"
    if "wgrad" in what:
        d = t(lambda: ops.conv3_wgrad(x, dy)); msg += f" | wgrad {d*1e6:7.1f} us {fl/d/1e12:6.1f} TF"
    print(msg, flush=True)

if __name__ == "__main__":
    what = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("fwd", "wgrad")
    shapes = [(96, 48, 48), (96, 96, 48), (96, 48, 96), (48, 48, 48), (24, 96, 96), (12, 192, 192), (6, 384, 384), (3, 768, 768)]
    if len(sys.argv) > 2:
        shapes = shapes[:int(sys.argv[2])]
    for s in shapes:
        run(*s, torch.bfloat16, what=what)
