"""accuracy of the instance-norm backward kernels in fp32 against a float64 torch reference: the register-resident fused launch
(S <= 2048 rows) next to the chunked reduce + apply pair (S > 2048), same statistics of the data"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import lib as L, ops

def ref(x, dy, gam, bet, slope=0.01, eps=1e-5):
    x = x.double().requires_grad_(True)
    g = gam.double().requires_grad_(True); b = bet.double().requires_grad_(True)
    m = x.mean(1, keepdim=True); v = x.var(1, unbiased=False, keepdim=True)
    y = torch.nn.functional.leaky_relu((x - m) / torch.sqrt(v + eps) * g + b, slope)
    y.backward(dy.double())
    return x.grad, g.grad, b.grad

rel = lambda a, b: float((a.double() - b).norm() / b.norm())
torch.manual_seed(0)
for S, C in [(512, 128), (512, 256), (64, 256), (1728, 192), (2048, 192), (2049, 192), (216, 384), (6912, 192)]:
    for mean in (0.0, 3.0):
        x = (torch.randn(1, S, C, device="cuda") + mean)
        dy = torch.randn_like(x)
        gam = torch.rand(C, device="cuda") + 0.5; bet = torch.randn(C, device="cuda") * 0.1
        styles = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.begin_step()
        y, stat = ops.instnorm_fwd(x, 1, S, styles, [gam], [bet], act=L.ACT_LEAKY)
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dx, _ = ops.instnorm_bwd(dy, None, x, 1, S, stat, styles, [gam], [dg], [db], act=L.ACT_LEAKY, betas=[bet])
        rx, rg, rb = ref(x, dy, gam, bet)
        xd = x.double(); yd = torch.nn.functional.leaky_relu((xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5) * gam.double() + bet.double(), 0.01)
        print(f"S {S:5d} C {C} mean {mean}: y {rel(y, yd):.2e}  dx {rel(dx, rx):.2e}  dgamma {rel(dg, rg):.2e}  dbeta {rel(db, rb):.2e}  max|dx err| {float((dx.double() - rx).abs().max()):.2e}")
