"""three replays of (400 tiny kernels beside 1 long one) for a rocprofv3 --kernel-trace: MODE=torch|split"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.runtime import graph as G

dev = torch.device("cuda:0")
a = torch.zeros(1 << 14, device=dev)
big = torch.zeros(1 << 26, device=dev)
s_side, s_run = torch.cuda.Stream(), torch.cuda.Stream()
G.SPLIT_REPLAY = os.environ.get("MODE", "split") == "split"
g = G._Graph()
with G._graph_capture(g):
    cur = torch.cuda.current_stream()
    s_side.wait_stream(cur)
    with torch.cuda.stream(s_side):
        big.add_(1.0)
    for _ in range(400):
        a.add_(1.0)
    cur.wait_stream(s_side)
torch.cuda.synchronize()
with torch.cuda.stream(s_run):
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
print("done", g.info)
