import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops
torch.manual_seed(0)
for (M, N, K) in [(512, 256, 128), (512, 256, 96), (512, 128, 128), (512, 256, 64), (2048, 64, 128), (512, 32, 128)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    y = ops.gemm_nt(a, w)
    yr = a.float() @ w.float().t()
    print(M, N, K, "nan:", bool(torch.isnan(y.float()).any()), "err", float((y.float() - yr).norm() / yr.norm()))
M, N, K = 512, 256, 128
a = torch.ones(M, K, device="cuda").bfloat16(); w = torch.ones(N, K, device="cuda").bfloat16()
y = ops.gemm_nt(a, w).float()
bad = (y != K)
print("bad count", int(bad.sum()), "rows with bad", bad.any(1).nonzero().flatten()[:20].tolist(), "cols with bad", bad.any(0).nonzero().flatten()[:40].tolist())
print(y[0, :40].tolist())
