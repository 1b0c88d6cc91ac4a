"""how fast a one-chain hipGraph of tiny kernels replays when another stream is busy: alone; with an event edge to an idle stream; beside a
plain kernel launch on another stream; beside another graph on another stream.  Times the main graph with events on its own stream."""
import sys, time
import torch

dev = torch.device("cuda:0")
n = 400
a = torch.zeros(1 << 14, device=dev)
big = torch.zeros(1 << 26, device=dev)
small_other = torch.zeros(1 << 14, device=dev)
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()

g_main = torch.cuda.CUDAGraph()
with torch.cuda.stream(s_main):
    with torch.cuda.graph(g_main, stream=s_main):
        for _ in range(n):
            a.add_(1.0)
g_side = torch.cuda.CUDAGraph()
with torch.cuda.stream(s_side):
    with torch.cuda.graph(g_side, stream=s_side):
        big.add_(1.0)
g_side_tiny = torch.cuda.CUDAGraph()
with torch.cuda.stream(s_side):
    with torch.cuda.graph(g_side_tiny, stream=s_side):
        small_other.add_(1.0)
torch.cuda.synchronize()


def run(kind, reps=10):
    ts = []
    for _ in range(reps + 2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if kind == "side kernel":
            with torch.cuda.stream(s_side):
                big.add_(1.0)
        elif kind == "side graph":
            with torch.cuda.stream(s_side):
                g_side.replay()
        elif kind == "tiny side graph":
            with torch.cuda.stream(s_side):
                g_side_tiny.replay()
        elif kind == "8 side kernels":
            with torch.cuda.stream(s_side):
                for _ in range(8):
                    big.add_(1.0)
        elif kind == "event edge to idle side":
            ev = torch.cuda.Event(); ev.record(s_main); s_side.wait_event(ev)
        elif kind == "fork event + side graph":
            ev = torch.cuda.Event(); ev.record(s_main); s_side.wait_event(ev)
            with torch.cuda.stream(s_side):
                g_side.replay()
        elif kind == "fork event + side kernel":
            ev = torch.cuda.Event(); ev.record(s_main); s_side.wait_event(ev)
            with torch.cuda.stream(s_side):
                big.add_(1.0)
        elif kind == "side graph, fork event after":
            with torch.cuda.stream(s_side):
                g_side.replay()
            ev = torch.cuda.Event(); ev.record(s_main); s_side.wait_event(ev)
        elif kind == "main kernel + fork event + side graph":
            with torch.cuda.stream(s_main):
                small_other.add_(1.0)
            ev = torch.cuda.Event(); ev.record(s_main); s_side.wait_event(ev)
            with torch.cuda.stream(s_side):
                g_side.replay()
        with torch.cuda.stream(s_main):
            e0.record(s_main)
            g_main.replay()
            e1.record(s_main)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


for kind in ["alone", "event edge to idle side", "tiny side graph", "side kernel", "side graph", "fork event + side graph", "fork event + side kernel", "side graph, fork event after",
             "main kernel + fork event + side graph", "8 side kernels", "alone"]:
    print(f"{n} tiny kernels as one graph, {kind:40s}: {run(kind):8.1f} us", flush=True)
