"""every NT GEMM launch (linears, 1x1x1 convs, ConvTranspose GEMMs) of one eager C-Swin-UNETR step, timed in place: shape, microseconds, GB/s of
algorithmic bytes.  Usage: python scripts/gemm_launches.py [c3]"""
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
arena = ParamArena([p for p in model.parameters() if p.requires_grad], torch.bfloat16)
x = torch.rand(1, 1, 96, 96, 96, device="cuda")
cot = torch.randn(1, 6, 96, 96, 96, device="cuda")
def step():
    arena.begin_step(); model(x, [0]).backward(cot); arena.publish()
for _ in range(3): step()
torch.cuda.synchronize()
shapes = []
orig_call = ops._call
def rec(fn_name, params, prof=None, prof_params=None, extra=()):
    if fn_name == "miseg_gemm" and prof is not None and prof[0] == "gemm_nt":
        shapes.append((params.M, params.N, params.K, bool(params.res), bool(params.stat), params.act, getattr(params, "scat_d", 0)))
    return orig_call(fn_name, params, prof=prof, prof_params=prof_params, extra=extra)
ops._call = rec
prof = roofline.profile_step(step)
ops._call = orig_call
lst = prof.get("gemm_nt", [])
print(len(lst), "gemm_nt launches timed,", len(shapes), "recorded")
rows = []
for (ms, fl, nb, allms), sh in zip(lst, shapes):
    rows.append((ms, sh, nb))
tot = sum(r[0] for r in rows)
for ms, sh, nb in rows:
    if ms * 1e3 >= 9.0:
        print(f"M {sh[0]:7d} N {sh[1]:5d} K {sh[2]:5d} res {int(sh[3])} stat {int(sh[4])} act {sh[5]} : {ms*1e3:7.1f} us  {nb/1e6:7.1f} MB  {nb/ms/1e6:7.0f} GB/s")
print(f"total {tot*1e3:.1f} us over {len(rows)} launches; launches below 9 us: {sum(1 for r in rows if r[0]*1e3 < 9.0)} = {sum(r[0] for r in rows if r[0]*1e3 < 9.0)*1e3:.1f} us")
