"""What the batched parameter cast of the headline net consists of, and what each kind of entry costs: the table of
runtime/arena.py::ParamArena._build_table grouped by (transpose, inner), each group timed as a launch of its own (params_version = None:
unconditional).  python scripts/debug/cast_table_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
import bench  # noqa: E402
from mi_seg_amd.hip import lib as L  # noqa: E402
from mi_seg_amd.hip import ops  # noqa: E402
from mi_seg_amd.runtime.arena import ParamArena  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def main():
    dtype = torch.bfloat16
    model = bench.build_model(dtype)
    params = [p for p in model.parameters() if p.requires_grad]
    arena = ParamArena(params, dtype)
    x = torch.randn(1, 1, 96, 96, 96, device="cuda")
    for _ in range(2):
        arena.begin_step()
        y = model(x, [0])
        y.float().sum().backward()
        arena.end_backward()
    torch.cuda.synchronize()
    ents = list(arena._req.items())
    groups = {}
    for key, ent in ents:
        p, dst = ent[0], ent[1]
        R = p.shape[0]
        Cc = p.numel() // R
        whole = R % 32 == 0 and Cc % 32 == 0
        groups.setdefault((int(key[1]), key[2], whole), []).append((p, dst, R, Cc, key))
    lib = L.load()
    dev = arena.flat.device

    def launch_of(items):
        descs = (L.CastDesc * len(items))()
        tile0 = 0
        for i, (p, dst, R, Cc, key) in enumerate(items):
            descs[i] = L.CastDesc(p.data_ptr(), dst.data_ptr(), R, Cc, int(key[1]), key[2], key[3], tile0)
            tile0 += ((R + 31) // 32) * ((Cc + 31) // 32)
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        n = len(items)

        def go():
            L.check(lib.miseg_param_cast_batch(raw.data_ptr(), n, tile0, L.BF16, None, None, ops._stream()), "param_cast_batch")
        return go, tile0, raw
    allitems = [it for g in groups.values() for it in g]
    go, tiles, keep = launch_of(allitems)
    ver = torch.tensor([1, -1, 0], dtype=torch.int64, device=dev)
    raw_all, n_all = keep, len(allitems)

    def go_versioned():
        L.check(lib.miseg_counter_add(C.c_void_p(ver.data_ptr()), 1, ops._stream()), "counter_add")
        L.check(lib.miseg_param_cast_batch(raw_all.data_ptr(), n_all, tiles, L.BF16, C.c_void_p(ver.data_ptr()), C.c_void_p(ver.data_ptr() + 8), ops._stream()), "param_cast_batch")

    def go_bump_only():
        L.check(lib.miseg_counter_add(C.c_void_p(ver.data_ptr()), 1, ops._stream()), "counter_add")
    print("versioned (live) launch incl. the version bump: %.1f us; the bump alone %.1f us" % (timed(go_versioned), timed(go_bump_only)))
    print("all: %d entries, %d tiles, %.1f M elements: %.1f us" % (len(allitems), tiles, sum(i[2] * i[3] for i in allitems) / 1e6, timed(go)))
    for k, items in sorted(groups.items()):
        go, tiles, keep = launch_of(items)
        el = sum(i[2] * i[3] for i in items)
        shapes = sorted({(i[2], i[3]) for i in items}, key=lambda s: -s[0] * s[1])[:6]
        print("transpose %d inner %d whole-tiles %s: %3d entries %6d tiles %6.2f M elements %7.1f us   largest %s" % (k[0], k[1], k[2], len(items), tiles, el / 1e6, timed(go), shapes))
    arena.detach()


if __name__ == "__main__":
    main()
