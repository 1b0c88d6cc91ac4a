"""run each kernel twice on identical inputs and report the relative difference of the outputs (fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops, lib as L
torch.manual_seed(0)
def rd(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
dt = torch.float32 if len(sys.argv) < 2 else torch.bfloat16
for (S, Cin, Cout) in [(32, 12, 12), (32, 24, 12), (64, 12, 12), (16, 48, 24), (8, 96, 96)]:
    x = torch.randn(2, S, S, S, Cin, device="cuda").to(dt); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
    dy = torch.randn(2, S, S, S, Cout, device="cuda").to(dt)
    fp, bp = ops.pack_conv3(w, dt)
    y1 = ops.conv3_fwd(x, fp, Cout).clone(); y2 = ops.conv3_fwd(x, fp, Cout).clone()
    d1 = ops.conv3_fwd(dy, bp, Cin).clone(); d2 = ops.conv3_fwd(dy, bp, Cin).clone()
    w1 = ops.conv3_wgrad(x, dy).clone(); w2 = ops.conv3_wgrad(x, dy).clone()
    print(f"conv {S}^3 {Cin}->{Cout}: fwd {rd(y1,y2):.2e} dgrad {rd(d1,d2):.2e} wgrad {rd(w1,w2):.2e}", flush=True)
for (S, C) in [(32**3, 12), (16**3, 24), (64**3, 12)]:
    B = 2
    x = torch.randn(B, S, C, device="cuda").to(dt); dy = torch.randn_like(x)
    styles = torch.zeros(B, dtype=torch.int32, device="cuda")
    gam = [torch.rand(C, device="cuda") + 0.5]; bet = [torch.zeros(C, device="cuda")]
    outs = []
    for _ in range(2):
        ops.begin_step()
        st = ops.instnorm_stats(x, B, S); y = ops.instnorm_apply(x, B, S, st, styles, gam, bet, act=L.ACT_LEAKY)
        dg = [torch.zeros(C, device="cuda")]; db = [torch.zeros(C, device="cuda")]
        dx, _ = ops.instnorm_bwd(dy, y, x, B, S, st, styles, gam, dg, db, act=L.ACT_LEAKY)
        outs.append((y.clone(), dx.clone(), dg[0].clone(), db[0].clone()))
    print(f"instnorm S={S} C={C}: y {rd(outs[0][0],outs[1][0]):.2e} dx {rd(outs[0][1],outs[1][1]):.2e} dgamma {rd(outs[0][2],outs[1][2]):.2e} dbeta {rd(outs[0][3],outs[1][3]):.2e}", flush=True)
for (M, K, N) in [(32768, 12, 48), (4096, 96, 24), (65536, 24, 12)]:
    a = torch.randn(M, N, device="cuda").to(dt); b = torch.randn(M, K, device="cuda").to(dt)
    g1 = ops.gemm_tn(a, b).clone(); g2 = ops.gemm_tn(a, b).clone()
    wt = torch.randn(N, K, device="cuda").to(dt)
    n1 = ops.gemm_nt(b, wt).clone(); n2 = ops.gemm_nt(b, wt).clone()
    print(f"gemm M={M} K={K} N={N}: tn {rd(g1,g2):.2e} nt {rd(n1,n2):.2e}", flush=True)
for dims, heads, C, ws, ss in [((32, 32, 32), 3, 12, (7, 7, 7), (3, 3, 3)), ((16, 16, 16), 6, 24, (7, 7, 7), (0, 0, 0))]:
    qkv = torch.randn(2, *dims, 3 * C, device="cuda").to(dt); qb = torch.randn(3 * C, device="cuda") * 0.3
    tab = torch.randn(2197, heads, device="cuda") * 0.5; scale = (C // heads) ** -0.5
    res = []
    for _ in range(2):
        out, lse = ops.winattn_fwd(qkv, qb, tab, heads, ws, ss, 7, scale)
        g = torch.ones_like(out) * 0.3 + torch.sin(torch.arange(out.numel(), device="cuda").float()).view_as(out).to(dt)
        dqb, dtb = torch.zeros_like(qb), torch.zeros_like(tab)
        dq = ops.winattn_bwd(qkv, out, lse, g, qb, tab, heads, ws, ss, 7, scale, dqb, dtb)
        res.append((out.clone(), dq.clone(), dqb.clone(), dtb.clone()))
    print(f"attn {dims} hd={C//heads}: out {rd(res[0][0],res[1][0]):.2e} dqkv {rd(res[0][1],res[1][1]):.2e} dqb {rd(res[0][2],res[1][2]):.2e} dtab {rd(res[0][3],res[1][3]):.2e}", flush=True)
