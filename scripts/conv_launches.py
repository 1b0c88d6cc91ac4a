"""every 3x3x3 convolution launch (forward / data gradient) of one eager C-Swin-UNETR step, timed in place by the library's in-situ timing
(csrc/common.cpp::miseg_prof_arm): shape, microseconds, TFLOP/s.  Usage: python scripts/conv_launches.py [c3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
import bench
from mi_seg_amd.hip import ops
from mi_seg_amd.runtime.arena import ParamArena
from mi_seg_amd.testing import roofline

wl = "c3" if "c3" in sys.argv else "c2"
model = bench.build_model(torch.bfloat16, wl)
params = [p for p in model.parameters() if p.requires_grad]
arena = ParamArena(params, torch.bfloat16)
x = torch.rand(1, 1, 96, 96, 96, device="cuda")
cot = torch.randn(1, 6, 96, 96, 96, device="cuda")
def step():
    arena.begin_step(); model(x, [0]).backward(cot); arena.publish()
for _ in range(3): step()
torch.cuda.synchronize()
# shapes: wrap conv3_fwd to remember them in call order
shapes = []
orig = ops.conv3_fwd
def rec(x_, wpk, Cout, **kw):
    shapes.append((tuple(x_.shape[1:4]), x_.shape[-1], Cout, kw.get("res") is not None, bool(kw.get("want_stat")), bool(kw.get("defer"))))
    return orig(x_, wpk, Cout, **kw)
ops.conv3_fwd = rec
import mi_seg_amd.hip.functional as HF
prof = roofline.profile_step(step)
ops.conv3_fwd = orig
rows = []
for name, lst in prof.items():
    if name.startswith("conv3_fwd"):
        for t in lst: rows.append((name, t))
print(f"{len(shapes)} conv3_fwd calls; per kernel name: " + ", ".join(f"{n}: {len(l)}" for n, l in prof.items() if n.startswith("conv3")))
tot = 0.0
for name, lst in prof.items():
    if not name.startswith("conv3_fwd"): continue
    for (ms, fl, nb, allms) in lst:
        tot += allms
        print(f"{name:45s} {ms*1e3:8.1f} us (call {allms*1e3:8.1f})  {fl/1e9:8.2f} GF  {fl/ms/1e9:8.1f} TF/s  {nb/1e6:7.1f} MB alg")
print(f"total {tot*1e3:.1f} us")
from collections import Counter
print(Counter(shapes))
