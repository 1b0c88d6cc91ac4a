"""how long does the host need to ENQUEUE one eager step (python + ctypes + torch autograd), vs the device time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
import bench
from mi_seg_amd.hip import ops
m = bench.build_model(torch.bfloat16)
x = torch.rand(1, 1, 96, 96, 96, device="cuda"); cot = torch.randn(1, 6, 96, 96, 96, device="cuda")
params = [p for p in m.parameters()]
def step():
    ops.begin_step()
    for p in params: p.grad = None
    y = m(x, [0]); y.backward(cot)
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms   enqueue+drain {1e3*(t2-t0):.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
