import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops, lib as L
S, Cin, Cout = 96, int(sys.argv[1]) if len(sys.argv) > 1 else 48, 48
dt = torch.bfloat16
x = torch.randn(1, S, S, S, Cin, device="cuda").to(dt); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
fp, _ = ops.pack_conv3(w, dt)
for _ in range(3): ops.conv3_fwd(x, fp, Cout)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["MISEG_HIP_LIB"])
n = 16 * 3456
buf = (C.c_ulonglong * n)()
assert lib.miseg_debug_stamps(buf, n) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(-1, 16)[:, :8].astype(np.int64)
d = a[:, 1:] - a[:, :-1]
print("phase 6, wave 0, mean cycles between stamps [wload, frags0, step0, step1, step2, wstore, barrier]:", d.mean(0).round(0), "total", (a[:, 7] - a[:, 0]).mean())
print("median:", np.median(d, 0))
