"""every instance-norm family call of one eager C-Swin-UNETR step, timed in place (csrc/common.cpp::miseg_prof_arm), in call order:
algorithmic MB, microseconds of the call (all its kernels), GB/s.  Usage: python scripts/norm_launches.py [c3] [nobranch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "nobranch" in sys.argv:
    os.environ["MISEG_NO_BRANCH"] = "1"
import torch
import __graft_entry__ as ge
ge.load_package()
import bench
from mi_seg_amd.hip import ops
from mi_seg_amd.runtime.arena import ParamArena
from mi_seg_amd.testing import roofline

wl = "c3" if "c3" in sys.argv else "c2"
model = bench.build_model(torch.bfloat16, wl)
arena = ParamArena([p for p in model.parameters() if p.requires_grad], torch.bfloat16)
x = torch.rand(1, 1, 96, 96, 96, device="cuda")
cot = torch.randn(1, 6, 96, 96, 96, device="cuda")
def step():
    arena.begin_step(); model(x, [0]).backward(cot); arena.publish()
for _ in range(3): step()
torch.cuda.synchronize()
prof = roofline.profile_step(step)
for name, lst in prof.items():
    if not name.startswith("instnorm"): continue
    tot = sum(t[3] for t in lst)
    print(f"== {name}: {len(lst)} calls, {tot*1e3:.0f} us")
    for i, (ms, fl, nb, allms) in enumerate(lst):
        print(f"{i:3d} {nb/1e6:8.2f} MB  first {ms*1e3:7.1f} us  call {allms*1e3:7.1f} us  {nb/allms/1e6:7.0f} GB/s")
    big = [t for t in lst if t[2] > 20e6]
    print(f"   calls > 20 MB: {len(big)}, {sum(t[3] for t in big)*1e3:.0f} us, {sum(t[2] for t in big)/1e6:.0f} MB -> {sum(t[2] for t in big)/sum(t[3] for t in big)/1e6:.0f} GB/s")
    small = [t for t in lst if t[2] <= 20e6]
    print(f"   calls <= 20 MB: {len(small)}, {sum(t[3] for t in small)*1e3:.0f} us ({sum(t[3] for t in small)/max(len(small),1)*1e3:.1f} us each)")
