"""Do two hipGraphs launched back to back on two streams run CONCURRENTLY, and can the second wait on the device for a flag the first sets
mid-way (miseg_flag_wait: a one-thread spin kernel - an ordering without a graph edge)?  Graph M = a chain of 400 small dependent kernels with a flag
set after the 150th; graph L = [flag wait] + 6 big streaming kernels.  Prints the device-clock stamps.  python scripts/debug/two_graph_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops, lib as L
lib = L.load()
dev = "cuda"
small = torch.randn(64, 64, device=dev); small2 = torch.zeros_like(small)
big = [torch.randn(96 ** 3, 48, device=dev).bfloat16() for _ in range(3)]
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
step = torch.zeros(1, dtype=torch.int64, device=dev)      # device step counter (graph M bumps it)
want = torch.zeros(1, dtype=torch.int64, device=dev)      # graph L's own copy of the step it belongs to
flag = torch.zeros(1, dtype=torch.int64, device=dev)
tout = torch.zeros(1, dtype=torch.int32, device=dev)
P = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def stamp(i): lib.miseg_debug_stamp(P(stamps, 8 * i), S())
def run_m():
    lib.miseg_counter_add(P(step), 1, S())
    stamp(0)
    for i in range(400):
        ops.add(small, small, out=small2)
        if i == 150:
            lib.miseg_counter_copy(P(flag), P(step), S()); stamp(1)
    stamp(2)
def run_l():
    lib.miseg_counter_add(P(want), 1, S())
    stamp(3)
    L.check(lib.miseg_flag_wait(P(flag), P(want), 500000, P(tout), S()), "flag_wait")
    stamp(4)
    for i in range(6):
        ops.add(big[0], big[1], out=big[2])
    stamp(5)
sm, sl = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sm): run_m()
torch.cuda.synchronize()
with torch.cuda.stream(sl): run_l()
torch.cuda.synchronize()
gm, gl = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(gm, stream=sm): run_m()
with torch.cuda.graph(gl, stream=sl): run_l()
torch.cuda.synchronize()
def both():
    with torch.cuda.stream(sm): gm.replay()
    with torch.cuda.stream(sl): gl.replay()
    torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter(); both(); dt = (time.perf_counter() - t0) * 1e6
    st = [v / 100.0 for v in stamps.cpu().tolist()]
    b = st[0]
    print(f"replay {rep}: wall {dt:7.0f} us | M begin 0, flag set {st[1]-b:6.0f}, M end {st[2]-b:6.0f} | L begin {st[3]-b:6.0f}, L past the wait {st[4]-b:6.0f}, L end {st[5]-b:6.0f} | "
          f"timed out {int(tout.item())} step {int(step.item())} want {int(want.item())}")
# each alone
with torch.cuda.stream(sm):
    t0 = time.perf_counter(); gm.replay(); torch.cuda.synchronize(); print(f"M alone {(time.perf_counter()-t0)*1e6:.0f} us")
with torch.cuda.stream(sl):
    t0 = time.perf_counter(); gl.replay(); torch.cuda.synchronize(); print(f"L alone (flag behind: must time out? no - want advanced past flag) {(time.perf_counter()-t0)*1e6:.0f} us, timed out {int(tout.item())}")
