"""what a live weight refresh costs (runtime/arena.py: param_cast_batch + pack_conv3_batch after the parameters changed): C-Swin-UNETR fs=48.
Usage: python scripts/micro/refresh_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
import bench
from mi_seg_amd.runtime.arena import ParamArena

model = bench.build_model(torch.bfloat16, "c2")
arena = ParamArena([p for p in model.parameters() if p.requires_grad], torch.bfloat16)
x = torch.rand(1, 1, 96, 96, 96, device="cuda"); cot = torch.randn(1, 6, 96, 96, 96, device="cuda")
for _ in range(2):
    arena.begin_step(); model(x, [0]).backward(cot); arena.publish()
torch.cuda.synchronize()
def once():
    arena.invalidate(); arena.epoch += 1; arena._refresh()
def timed(label):
    once(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): once()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e6)
    print(f"live refresh, {label}: {sorted(ts)[2]:.1f} us")
timed("counter_add + param_cast_batch + pack_conv3_batch")
pt, arena._ptable = arena._ptable, None
timed("counter_add + param_cast_batch")
arena._ptable, tb, arena._table = pt, arena._table, None
timed("counter_add + pack_conv3_batch")
arena._table = tb
print("cast descriptors", arena._table[1], "tiles", arena._table[2], "| pack descriptors", arena._ptable[1], "tiles", arena._ptable[2])
