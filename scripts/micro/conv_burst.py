"""one 96^3 48->48 convolution launch timed with device events: back to back, and with idle gaps in front of every launch"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops
x = torch.randn(1, 96, 96, 96, 48, device="cuda").to(torch.bfloat16)
w = torch.randn(48, 48, 3, 3, 3, device="cuda") / 36
fwdp, _ = ops.pack_conv3(w, torch.bfloat16)
out = torch.empty(1, 96, 96, 96, 48, device="cuda", dtype=torch.bfloat16)
for _ in range(5): ops.conv3_fwd(x, fwdp, 48, out=out)
torch.cuda.synchronize()
for gap_ms in (0, 0.2, 1, 5, 20):
    ts = []
    for _ in range(15):
        if gap_ms: torch.cuda.synchronize(); time.sleep(gap_ms / 1e3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv3_fwd(x, fwdp, 48, out=out); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
    print(f"idle gap {gap_ms:5.1f} ms: median {v[7]:7.1f} us  min {v[0]:7.1f}  max {v[-1]:7.1f}")
