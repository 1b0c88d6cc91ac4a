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
if not (len(sys.argv) > 1 and sys.argv[1] == "cold"):
    for S, C in shapes:
        run(S, C)


def run_cold(S, C, sets=5, dtype=torch.bfloat16, act=L.ACT_LEAKY):
    """the same kernels on tensors that are NOT in the 256 MB Infinity Cache: `sets` disjoint operand sets visited round robin (what the
    training step sees: every tensor was written ~a millisecond and ~a gigabyte of other traffic ago)"""
    B = 1
    xs = [torch.randn(B, S, C, device="cuda").to(dtype) for _ in range(sets)]
    dys = [torch.randn_like(x) for x in xs]
    ys = [torch.empty_like(x) for x in xs]
    styles = torch.zeros(B, dtype=torch.int32, device="cuda")
    gam = [torch.ones(C, device="cuda") for _ in range(2)]
    bet = [torch.zeros(C, device="cuda") for _ in range(2)]
    dg = [torch.zeros(C, device="cuda") for _ in range(2)]
    db = [torch.zeros(C, device="cuda") for _ in range(2)]
    nb = xs[0].numel() * xs[0].element_size()
    keep = ops.instnorm_stats(xs[0], B, S).clone()
    for i in range(sets):
        ops.instnorm_apply(xs[i], B, S, keep, styles, gam, bet, act=act, out=ys[i])
    k = [0]

    def nxt():
        k[0] = (k[0] + 1) % sets
        return k[0]
    iters = 3 * sets
    t_s = timed(lambda: ops.instnorm_stats(xs[nxt()], B, S), iters)
    t_a = timed(lambda: (lambda i: ops.instnorm_apply(xs[i], B, S, keep, styles, gam, bet, act=act, out=ys[i]))(nxt()), iters)
    t_r = timed(lambda: (lambda i: ops.instnorm_apply(xs[i], B, S, keep, styles, gam, bet, res=dys[i], act=act, out=ys[i]))(nxt()), iters)
    t_b = timed(lambda: (lambda i: ops.instnorm_bwd(dys[i], ys[i], xs[i], B, S, keep, styles, gam, dg, db, act=act))(nxt()), iters)
    t_b2 = timed(lambda: (lambda i: ops.instnorm_bwd(dys[i], None, xs[i], B, S, keep, styles, gam, dg, db, act=act, betas=bet))(nxt()), iters)
    print(f"cold S {S:7d} C {C:4d}: stats {t_s:7.1f} us ({nb/t_s/1e6:5.2f} TB/s)  apply {t_a:7.1f} us ({2*nb/t_a/1e6:5.2f} TB/s)  apply+res {t_r:7.1f} us ({3*nb/t_r/1e6:5.2f} TB/s)  "
          f"bwd(y) {t_b:7.1f} us ({7*nb/t_b/1e6:5.2f} TB/s of 7N)  bwd(sign from x) {t_b2:7.1f} us ({5*nb/t_b2/1e6:5.2f} TB/s of 5N)", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "cold":
    for S, C in [(96**3, 48), (48**3, 96), (48**3, 48), (24**3, 192)]:
        run_cold(S, C)
