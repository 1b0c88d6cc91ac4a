"""Where a small instance-norm launch spends its time (in-kernel s_memtime stamps of thread 0 / workgroup 0, each behind a wait for that
wave's outstanding memory operations).  Needs the debug build:
    hipcc ... -DMISEG_NORM_STAMPS -c mi-seg_amd/csrc/norm.hip -o /tmp/norm_dbg.o ; link with the other objects -> scripts/micro/libmiseg_norm_dbg.so
    MISEG_HIP_LIB=scripts/micro/libmiseg_norm_dbg.so python scripts/debug/norm_stamps.py        (scripts/debug/build_norm_dbg.sh does both)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge

ge.load_package()
from mi_seg_amd.hip import lib as L, ops

lib = L.load()
lib.miseg_debug_norm_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
NAMES = ["rows + affine loads", "statistics gather", "mask + sums", "totals (shuffles + LDS)", "affine-gradient atomics", "gradient + stores"]
for S, Cc in [(216, 384), (1728, 192), (1728, 384), (27, 768)]:
    x = torch.randn(1, S, Cc, device="cuda").bfloat16()
    dy = torch.randn_like(x)
    styles = torch.zeros(1, dtype=torch.int32, device="cuda")
    gam, bet = [torch.ones(Cc, device="cuda")] * 2, [torch.zeros(Cc, device="cuda")] * 2
    dg, db = [torch.zeros(Cc, device="cuda")] * 2, [torch.zeros(Cc, device="cuda")] * 2
    y, stat = ops.instnorm_fwd(x, 1, S, styles, gam, bet, act=L.ACT_LEAKY)
    for _ in range(3):
        ops.instnorm_bwd(dy, None, x, 1, S, stat, styles, gam, dg, db, act=L.ACT_LEAKY, betas=bet)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.instnorm_bwd(dy, None, x, 1, S, stat, styles, gam, dg, db, act=L.ACT_LEAKY, betas=bet)
    e1.record()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.miseg_debug_norm_stamps(buf)
    t = [buf[i] for i in range(7)]
    # __builtin_readcyclecounter = s_memtime: shader-clock ticks (2.4 GHz)
    parts = "  ".join(f"{n} {(t[i + 1] - t[i]) / 2.4:.0f} ns" for i, n in enumerate(NAMES))
    print(f"S {S} C {Cc}: eager launch-to-launch {e0.elapsed_time(e1) * 50:.1f} us | in-kernel {(t[6] - t[0]) / 2.4:.0f} ns: {parts}")
