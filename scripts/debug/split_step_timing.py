"""where the data-parallel (two-graph) step loses time against the single-graph step, one rank, no collective: event-timed replays of graph A
(forward + decoder-side backward), graph B (encoder / Swin side + side branch) and of the single graph, back to back"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
import bench
from mi_seg_amd.runtime.arena import ParamArena
from mi_seg_amd.runtime.graph import GraphedStep

m = bench.build_model(torch.bfloat16)
params = [p for p in m.parameters() if p.requires_grad]
arena = ParamArena(params, torch.bfloat16)
x = torch.rand(1, 1, 96, 96, 96, device="cuda")
cot = torch.randn(1, 6, 96, 96, 96, device="cuda")
single = GraphedStep(m, x.shape, cot.shape, arena=arena)
single(x, [0], cot)
m.split_defers = True
split = GraphedStep(m, x.shape, cot.shape, arena=arena, split=True)
split(x, [0], cot)
(gA, gB), _, _ = split.graphs[(0, 1)]
(gS, _), _, _ = single.graphs[(0, 1)]
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tS = timed(lambda: gS.replay())
tAB = timed(lambda: (gA.replay(), gB.replay()))
tA = timed(lambda: gA.replay())
tB = timed(lambda: gB.replay())
print(f"single graph {tS:.3f} ms | A then B {tAB:.3f} ms (A alone {tA:.3f}, B alone {tB:.3f}, sum {tA + tB:.3f})")
