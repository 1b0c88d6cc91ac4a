"""micro-benchmark of the instance-norm kernels on the headline tensor shapes (captured in a hipGraph so that the host
launch path does not hide the kernel time); prints achieved HBM bandwidth against the algorithmic bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops, lib as L

def timed(fn, iters=20):
    ops.begin_step(); fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.begin_step()
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

def run(S, C, dtype=torch.bfloat16, act=L.ACT_LEAKY):
    B = 1
    x = torch.randn(B, S, C, device="cuda").to(dtype)
    dy = torch.randn_like(x)
    styles = torch.zeros(B, dtype=torch.int32, device="cuda")
    gam = [torch.ones(C, device="cuda") for _ in range(2)]
    bet = [torch.zeros(C, device="cuda") for _ in range(2)]
    dg = [torch.zeros(C, device="cuda") for _ in range(2)]
    db = [torch.zeros(C, device="cuda") for _ in range(2)]
    nb = x.numel() * x.element_size()
    stat = ops.instnorm_stats(x, B, S)
    y = ops.instnorm_apply(x, B, S, stat, styles, gam, bet, act=act)
    keep = stat.clone()
    t_f = timed(lambda: ops.instnorm_fwd(x, B, S, styles, gam, bet, act=act))
    t_s = timed(lambda: ops.instnorm_stats(x, B, S))
    t_a = timed(lambda: ops.instnorm_apply(x, B, S, keep, styles, gam, bet, act=act, out=y))
    t_b = timed(lambda: ops.instnorm_bwd(dy, y, x, B, S, keep, styles, gam, dg, db, act=act))
    print(f"S {S:7d} C {C:4d}: fwd {t_f:6.1f} us | stats {t_s:7.1f} us ({nb/t_s/1e6:5.2f} TB/s)  apply {t_a:7.1f} us ({2*nb/t_a/1e6:5.2f} TB/s)  "
          f"bwd {t_b:7.1f} us ({7*nb/t_b/1e6:5.2f} TB/s of 7N)", flush=True)

shapes = [(96**3, 48), (48**3, 96), (48**3, 48), (24**3, 192), (24**3, 96), (12**3, 384), (6**3, 768), (27, 768)]
if len(sys.argv) > 1 and sys.argv[1] == "mid":
    shapes = [(24**3, 96), (24**3, 192), (12**3, 192), (12**3, 384), (6**3, 384), (6**3, 768)]
for S, C in shapes:
    run(S, C)
