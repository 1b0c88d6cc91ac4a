"""does an EXTERNAL event recorded inside a captured hipGraph let another stream start work in the middle of a replay (ROCm 7 / torch 2.10)?
graph = [a = fill(1); record(ev); long spin; b = fill(2)]; after g.replay() a side stream waits for ev and copies a -> c."""
import sys, time
import torch
dev = "cuda"
a = torch.zeros(1 << 20, device=dev); b = torch.zeros(1 << 20, device=dev); c = torch.zeros(1 << 20, device=dev)
side = torch.cuda.Stream()
try:
    ev = torch.cuda.Event(external=True)
except TypeError as e:
    print("no external events in this torch:", e); sys.exit(0)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    a.fill_(0.5); torch.cuda._sleep(1000)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    a.add_(1.0)
    ev.record()
    torch.cuda._sleep(int(2.4e9 * 0.02))      # ~20 ms
    b.add_(2.0)
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    g.replay()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        c.copy_(a)
        done = torch.cuda.Event(); done.record()
    done.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"replay {it}: side copy finished after {1e3 * (t1 - t0):.2f} ms, graph after {1e3 * (t2 - t0):.2f} ms; c = {float(c[0])} (a = {float(a[0])}), b = {float(b[0])}")
