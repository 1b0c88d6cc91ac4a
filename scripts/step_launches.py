"""every hooked launch of one eager training step, timed in place (csrc/common.cpp::miseg_prof_arm), in call order per hooked name:
microseconds of the first kernel and of the whole call, GFLOP, algorithmic MB.  Usage: python scripts/step_launches.py [c2|c3] [name prefix ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
import bench
from mi_seg_amd.runtime.arena import ParamArena
from mi_seg_amd.testing import roofline

wl = "c3" if "c3" in sys.argv else "c2"
want = [a for a in sys.argv[1:] if a not in ("c2", "c3")]
model = bench.build_model(torch.bfloat16, wl)
arena = ParamArena([p for p in model.parameters() if p.requires_grad], torch.bfloat16)
x = torch.rand(1, 1, 96, 96, 96, device="cuda")
cot = torch.randn(1, 6, 96, 96, 96, device="cuda")
def step():
    arena.begin_step(); model(x, [0]).backward(cot); arena.publish()
for _ in range(3): step()
torch.cuda.synchronize()
prof = roofline.profile_step(step)
print("totals: " + ", ".join(f"{n}: {len(l)} calls {sum(t[3] for t in l) * 1e3:.0f} us" for n, l in sorted(prof.items(), key=lambda kv: -sum(t[3] for t in kv[1]))))
for name, lst in prof.items():
    if want and not any(name.startswith(w) for w in want): continue
    print(f"== {name}: {len(lst)} calls, {sum(t[3] for t in lst) * 1e3:.0f} us")
    for i, (ms, fl, nb, allms) in enumerate(lst):
        print(f"{i:3d}  first {ms * 1e3:7.1f} us  call {allms * 1e3:7.1f} us  {fl / 1e9:8.2f} GF {fl / allms / 1e9 if allms else 0:7.1f} TF/s  {nb / 1e6:8.2f} MB {nb / allms / 1e6 if allms else 0:6.0f} GB/s")
