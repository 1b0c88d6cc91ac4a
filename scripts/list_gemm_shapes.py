"""which GEMM problems of one C-Swin-UNETR step land in which kernel (dispatch rules of miseg_gemm restated)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
import bench
from mi_seg_amd.hip import ops, lib as L
model = bench.build_model(torch.bfloat16)
x = torch.randn(1, 1, 96, 96, 96, device="cuda"); cot = torch.randn(1, 6, 96, 96, 96, device="cuda")
seen = collections.Counter()
orig = ops._call
def spy(name, p, prof=None):
    if name == "miseg_gemm":
        kind = "TN" if p.ta else "NT"
        if kind == "NT":
            k = "stream" if (p.K in (48, 96, 192) and p.M >= 4096) else "small" if (p.M <= 2048 and p.K % 32 == 0) else "generic"
        else:
            k = "tn"
        seen[(kind, k, p.M, p.N, p.K)] += 1
    return orig(name, p, prof)
ops._call = spy
ops.begin_step()
model(x, [0]).backward(cot)
torch.cuda.synchronize()
for (kind, k, M, N, K), n in sorted(seen.items(), key=lambda kv: (kv[0][0], kv[0][1], -kv[0][2])):
    print(f"{kind} {k:8s} M {M:7d} N {N:5d} K {K:5d}  x{n}")
