import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops
dt = torch.bfloat16
def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
for S, Cin, Cout in [(96, 48, 48), (96, 96, 48), (48, 48, 48), (24, 96, 96)]:
    x = torch.randn(1, S, S, S, Cin, device="cuda").to(dt); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
    r = torch.randn(1, S, S, S, Cout, device="cuda").to(dt)
    fp, _ = ops.pack_conv3(w, dt)
    a = t(lambda: ops.conv3_fwd(x, fp, Cout))
    b = t(lambda: (ops.begin_step(), ops.conv3_fwd(x, fp, Cout, want_stat=True)))
    c = t(lambda: ops.conv3_fwd(x, fp, Cout, res=r))
    d = t(lambda: (ops.begin_step(), ops.instnorm_stats(r, 1, S ** 3)))
    e = t(lambda: ops.add(r, r))
    print(f"{S}^3 {Cin}->{Cout}: plain {a:6.1f}  +stat {b:6.1f} (incl. pool fill)  +res {c:6.1f} | separate stats {d:5.1f} add {e:5.1f}")
