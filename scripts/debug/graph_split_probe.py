"""the split replay (csrc/graphsplit.cpp) on the shapes of scripts/debug/graph_edges_probe.py: a chain of tiny kernels beside a few long
ones on a side stream, replayed by torch (one multi-stream graph) and as single-stream pieces.  DEVICE time: every replay is queued behind
a ~3 ms blocker on the launching stream, so the host has enqueued all of it before the device starts; wall time of back-to-back replays
beside it (the host's enqueue rate).  Usage: python scripts/debug/graph_split_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.runtime import graph as G

dev = torch.device("cuda:0")
a = torch.zeros(1 << 14, device=dev)
big = torch.zeros(1 << 26, device=dev)
blocker = torch.zeros(1 << 28, device=dev)      # 1 GB: an add_ takes ~0.45 ms
s_side = torch.cuda.Stream()
s_run = torch.cuda.Stream()


def build(split, n_tiny, m_big, forks):
    """n_tiny kernels on the capture stream; `forks` times: fork, m_big long kernels on the side stream, join after n_tiny / forks tiny ones"""
    G.SPLIT_REPLAY = split
    g = G._Graph()
    with G._graph_capture(g):
        cur = torch.cuda.current_stream()
        per = n_tiny // max(forks, 1)
        for f in range(max(forks, 1)):
            if forks:
                s_side.wait_stream(cur)
                with torch.cuda.stream(s_side):
                    for _ in range(m_big):
                        big.add_(1.0)
            for _ in range(per):
                a.add_(1.0)
            if forks:
                cur.wait_stream(s_side)
    return g


def device_time(g, stream, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for _ in range(8):
                blocker.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def wall_time(g, stream, reps=20):
    with torch.cuda.stream(stream):
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for n_tiny, m_big, forks in [(400, 0, 0), (400, 1, 1), (400, 4, 1), (400, 1, 2), (400, 1, 4), (400, 1, 8)]:
    line = f"{n_tiny} tiny kernels, {forks} fork/join pair(s) x {m_big} long kernel(s):"
    for split in (False, True):
        g = build(split, n_tiny, m_big, forks)
        for name, stream in (("null", torch.cuda.default_stream()), ("own", s_run)):
            line += f"  {'split' if split else 'torch'} replay on the {name} stream: device {device_time(g, stream):7.1f} us wall {wall_time(g, stream):7.1f} us;"
    print(line, flush=True)
