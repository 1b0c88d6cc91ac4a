"""the grouped conv weight-gradient launch of the headline step (the <= 48^3 layers of the main stream) against its layers one by one:
where do its ~450 us go?  Usage: python scripts/micro/wgrad_group_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops

LAYERS = [(48, 96, 48), (48, 48, 48), (24, 192, 96), (24, 96, 96), (12, 384, 192), (12, 192, 192), (6, 768, 384), (6, 384, 384), (3, 768, 768), (3, 768, 768),
          (24, 96, 96), (24, 96, 96), (12, 192, 192), (12, 192, 192)]


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / iters)
    return sorted(ts)[3] * 1e6


def main():
    dt = torch.bfloat16
    items = []
    for S, Cin, Cout in LAYERS:
        x = torch.randn(1, S, S, S, Cin, device="cuda").to(dt)
        dy = torch.randn(1, S, S, S, Cout, device="cuda").to(dt)
        dw = torch.zeros(Cout, Cin, 3, 3, 3, device="cuda")
        items.append((x, dy, dw, 2))
    tot = 0.0
    for (S, Cin, Cout), (x, dy, dw, _) in zip(LAYERS, items):
        d = timed(lambda: ops.conv3_wgrad(x, dy, dw=dw, accumulate=2))
        fl = 2.0 * S ** 3 * 27 * Cin * Cout
        tot += d
        print(f"{S}^3 {Cin}->{Cout}: {d:7.1f} us  {fl / d / 1e6:6.1f} TF/s   dw {dw.numel() * 4 / 1e6:.1f} MB", flush=True)
    print(f"one by one: {tot:.1f} us")
    def grouped(sel):
        q = [items[i] for i in sel]
        ops._flush_conv_wgrads(q)
        ops._WGRAD_KEEP.clear()
    d = timed(lambda: grouped(range(len(items))))
    print(f"grouped ({len(items)} layers): {d:.1f} us")
    for name, sel in (("all but 6^3 + 3^3", [0, 1, 2, 3, 4, 5, 10, 11, 12, 13]), ("all but 3^3", [0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13]), ("48^3 + 24^3", [0, 1, 2, 3, 10, 11]), ("12^3", [4, 5, 12, 13]), ("6^3 + 3^3", [6, 7, 8, 9]), ("3^3", [8, 9]), ("6^3", [6, 7])):
        print(f"grouped {name}: {timed(lambda: grouped(sel)):.1f} us")


if __name__ == "__main__":
    main()
