"""variants of the split plan's launch sequence, written with torch streams / events, around two single-chain graphs (400 tiny kernels; one
long kernel): which step of the sequence costs the main chain its fast replay?  Device time of the main graph (events on its stream) and of
the whole sequence (events on the caller's stream), everything queued behind a blocker so the host is ahead."""
import sys, time
import torch

dev = torch.device("cuda:0")
n = 400
a = torch.zeros(1 << 14, device=dev)
big = torch.zeros(1 << 26, device=dev)
blocker = torch.zeros(1 << 28, device=dev)
s_a, s_b, s_c = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()

g_main = torch.cuda.CUDAGraph()
with torch.cuda.stream(s_a):
    with torch.cuda.graph(g_main, stream=s_a):
        for _ in range(n):
            a.add_(1.0)
g_side = torch.cuda.CUDAGraph()
with torch.cuda.stream(s_b):
    with torch.cuda.graph(g_side, stream=s_b):
        big.add_(1.0)
torch.cuda.synchronize()
E = lambda: torch.cuda.Event()


def seq(variant, caller):
    """returns (main graph us, whole sequence us)"""
    main = s_a if variant.startswith("own") else caller
    side = s_b
    t0, t1, m0, m1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(caller):
        for _ in range(8):
            blocker.add_(1.0)
        t0.record(caller)
        start = E(); start.record(caller)
        if main is not caller:
            main.wait_event(start)
        if "nofork" not in variant:
            side.wait_event(start)
        with torch.cuda.stream(side):
            g_side.replay()
        with torch.cuda.stream(main):
            if "timed" in variant:
                m0.record(main)
            g_main.replay()
            if "timed" in variant:
                m1.record(main)
        if "nojoin" not in variant:
            se = E(); se.record(side); main.wait_event(se)
            if main is not caller:
                en = E(); en.record(main); caller.wait_event(en)
        t1.record(caller)
    torch.cuda.synchronize()
    return (m0.elapsed_time(m1) * 1e3 if "timed" in variant else float("nan")), t0.elapsed_time(t1) * 1e3


for cname, caller in (("null", torch.cuda.default_stream()), ("s_c", s_c)):
    for variant in ["caller", "caller timed", "caller nojoin timed", "caller nofork timed", "caller nofork nojoin timed", "own", "own timed", "own nojoin timed", "own nofork timed"]:
        r = sorted(seq(variant, caller) for _ in range(7))[3]
        print(f"caller stream {cname:4s} variant {variant:28s}: main graph {r[0]:8.1f} us, sequence {r[1]:8.1f} us", flush=True)
