"""does relaunching the SAME hipGraph exec wait for its previous launch?  100 small kernels per graph, 30 replays back to back: one exec,
then two execs of the same work alternating.  Wall time per replay; run under rocprofv3 --kernel-trace to see the gaps."""
import sys, time
import torch
x = torch.zeros(1 << 20, device="cuda")
def work():
    for _ in range(100):
        x.add_(1.0)
def capture():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work()
    return g
work(); torch.cuda.synchronize()
g1, g2 = capture(), capture()
for name, seq in (("same exec", [g1] * 30), ("alternating", [g1, g2] * 15)):
    seq[0].replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for g in seq:
        g.replay()
    t_sub = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{name}: {t / 30 * 1e6:.1f} us per replay (host submit {t_sub / 30 * 1e6:.1f} us per replay)")
