import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops
x = torch.randn(1, 96, 96, 96, 48, device="cuda").to(torch.bfloat16)
w = torch.randn(48, 48, 3, 3, 3, device="cuda") / 36
fwdp, _ = ops.pack_conv3(w, torch.bfloat16)
out = torch.empty(1, 96, 96, 96, 48, device="cuda", dtype=torch.bfloat16)
g = torch.cuda.CUDAGraph()
ops.conv3_fwd(x, fwdp, 48, out=out); torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(50): ops.conv3_fwd(x, fwdp, 48, out=out)
print("start", flush=True)
t0 = time.time()
n = 0
while time.time() - t0 < 12:
    g.replay(); n += 50
    if n % 5000 == 0: torch.cuda.synchronize()
torch.cuda.synchronize()
print("avg us per conv", (time.time() - t0) / n * 1e6)
