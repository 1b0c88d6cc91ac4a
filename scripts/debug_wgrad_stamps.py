"""where a conv3 weight-gradient workgroup spends its time (debug build with -DMISEG_WGRAD_STAMPS):
MISEG_HIP_LIB=scripts/micro/libmiseg_hip_dbg.so python scripts/debug_wgrad_stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops, lib as L
lib = L.load()
lib.miseg_debug_wgrad_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
dt = torch.bfloat16
for S, Cin, Cout in [(96, 48, 48), (96, 96, 48), (48, 48, 48), (24, 96, 96), (12, 192, 192), (6, 384, 384), (3, 768, 768)]:
    x = torch.randn(1, S, S, S, Cin, device="cuda").to(dt); dy = torch.randn(1, S, S, S, Cout, device="cuda").to(dt)
    for _ in range(2): ops.conv3_wgrad(x, dy)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    lib.miseg_debug_wgrad_stamps(buf)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); ops.conv3_wgrad(x, dy); t1.record(); torch.cuda.synchronize()
    lib.miseg_debug_wgrad_stamps(buf)
    stage, kloop, epi, bricks, gwait, lst, vmw = [buf[i] for i in range(7)]
    tot = stage + kloop + epi
    print(f"{S}^3 {Cin}->{Cout}: {t0.elapsed_time(t1)*1e3:7.1f} us (both kernels) | per brick: staging {10*stage/bricks:6.0f} ns (barrier {10*gwait/bricks:5.0f}, lds store {10*lst/bricks:5.0f} of which vmcnt wait {10*vmw/bricks:5.0f}, load issue {10*(stage-gwait-lst)/bricks:5.0f}) k-loop {10*kloop/bricks:6.0f} ns | "
          f"share staging {100*stage/tot:4.1f}% k-loop {100*kloop/tot:4.1f}% epilogue {100*epi/tot:4.1f}%  bricks {bricks}")

# the grouped launch of a C-Swin-UNETR step (every 3^3 conv of 48^3 and below)
layers = [(48, 48, 48)] * 3 + [(48, 96, 48)] + [(24, 96, 96)] * 3 + [(24, 192, 96)] + [(12, 192, 192)] * 3 + [(12, 384, 192)] + [(6, 384, 384), (6, 768, 384)] + [(3, 768, 768)] * 2
sets = {"all": layers, "48^3": layers[:4], "24^3": layers[4:8], "12^3": layers[8:12], "6^3": layers[12:14], "3^3": layers[14:]}
for name, ls in sets.items():
    items = []
    for S, Cin, Cout in ls:
        items.append((torch.randn(1, S, S, S, Cin, device="cuda").to(dt), torch.randn(1, S, S, S, Cout, device="cuda").to(dt), torch.zeros(Cout, Cin, 3, 3, 3, device="cuda")))
    for rep in range(2):
        ops.DEFAULT_QUEUES = ops.StepQueues()
        for x, dy, dw in items:
            ops.conv3_wgrad(x, dy, dw=dw, accumulate=True)
        torch.cuda.synchronize(); lib.miseg_debug_wgrad_stamps(buf)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); ops.DEFAULT_QUEUES.flush(); t1.record(); torch.cuda.synchronize()
        ops.DEFAULT_QUEUES = None
    lib.miseg_debug_wgrad_stamps(buf)
    stage, kloop, epi, bricks, gwait, lst, vmw = [buf[i] for i in range(7)]
    tot = stage + kloop + epi
    print(f"group {name:5s}: {t0.elapsed_time(t1)*1e3:7.1f} us (both kernels) | workgroup-time sum {10*tot/1e3:8.1f} us = {10*tot/1e3/256:6.1f} us per CU | staging {100*stage/tot:4.1f}% k-loop {100*kloop/tot:4.1f}% epilogue {100*epi/tot:4.1f}%  bricks {bricks}")
