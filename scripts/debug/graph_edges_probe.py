"""What a cross-stream edge costs inside a replayed hipGraph: two chains of n small (launch-bound) kernels captured on
two streams with one fork and one join, plus k extra event edges main -> side, side -> main, or both, spread evenly.  Prints the
replay time per variant; eager two-stream time beside it.  Env knobs of the HIP runtime worth sweeping: DEBUG_HIP_FORCE_GRAPH_QUEUES,
DEBUG_CLR_GRAPH_PACKET_CAPTURE, GPU_MAX_HW_QUEUES.   Usage: python scripts/debug/graph_edges_probe.py [n]"""
import sys, time
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
a = torch.zeros(1 << 14, device=dev)
b = torch.zeros(1 << 14, device=dev)
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()


def body(k_ms, k_sm):
    """n kernels on each chain; k_ms edges main -> side and k_sm edges side -> main"""
    s_side.wait_stream(s_main)
    ms_at = set(int((i + 1) * n / (k_ms + 1)) for i in range(k_ms))
    sm_at = set(int((i + 0.5) * n / (k_sm + 0.5)) for i in range(k_sm))
    for i in range(n):
        with torch.cuda.stream(s_main):
            a.add_(1.0)
            if i in ms_at:
                e = torch.cuda.Event(); e.record(s_main); s_side.wait_event(e)
        with torch.cuda.stream(s_side):
            b.add_(1.0)
            if i in sm_at:
                e = torch.cuda.Event(); e.record(s_side); s_main.wait_event(e)
    s_main.wait_stream(s_side)


def timed_graph(k_ms, k_sm, reps=20):
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s_main):
        with torch.cuda.graph(g, stream=s_main):
            body(k_ms, k_sm)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def timed_eager(k_ms, k_sm, reps=5):
    with torch.cuda.stream(s_main):
        body(k_ms, k_sm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(s_main):
            body(k_ms, k_sm)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def single_chain(reps=20):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s_main):
        with torch.cuda.graph(g, stream=s_main):
            for i in range(n):
                a.add_(1.0); b.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print(f"one chain of {2 * n} kernels: {single_chain():8.1f} us")
for k_ms, k_sm in [(0, 0), (1, 0), (0, 1), (1, 1), (3, 0), (0, 3), (3, 3), (8, 8)]:
    print(f"two chains of {n}, edges main->side {k_ms} side->main {k_sm}: graph {timed_graph(k_ms, k_sm):8.1f} us   eager {timed_eager(k_ms, k_sm):8.1f} us", flush=True)


# asymmetric: the step's shape - a long chain of tiny kernels on main, a few long kernels on the side stream
big = torch.zeros(1 << 26, device=dev)      # 256 MB: an add_ takes ~100+ us


def asym(m_big, reps=20):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s_main):
        with torch.cuda.graph(g, stream=s_main):
            s_side.wait_stream(s_main)
            with torch.cuda.stream(s_side):
                for _ in range(m_big):
                    big.add_(1.0)
            for i in range(2 * n):
                a.add_(1.0)
            s_main.wait_stream(s_side)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def big_alone(m_big, reps=20):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s_main):
        with torch.cuda.graph(g, stream=s_main):
            for _ in range(m_big):
                big.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for m in (0, 1, 2, 4, 8):
    print(f"main {2 * n} tiny kernels beside {m} big ones on the side stream: {asym(m):8.1f} us   (the big ones alone: {big_alone(m) if m else 0.0:8.1f} us)", flush=True)


# is the multi-stream cost per node additive for longer kernels?  main: 200 kernels of ~8 us (4 M floats) alone / beside one big side kernel
mid = torch.zeros(1 << 21, device=dev)


def mid_chain(with_side, reps=20, count=200):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s_main):
        with torch.cuda.graph(g, stream=s_main):
            if with_side:
                s_side.wait_stream(s_main)
                with torch.cuda.stream(s_side):
                    big.add_(1.0)
            for i in range(count):
                mid.add_(1.0)
            if with_side:
                s_main.wait_stream(s_side)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print(f"main 200 kernels of 2 M floats: alone {mid_chain(False):8.1f} us, beside one big side kernel {mid_chain(True):8.1f} us")
