"""Op-by-op dump of one forward + backward of the plain UNet (BASELINE configs[0], fixture c1_64, fp32) for a differential run: every
ops.conv3_fwd / conv3_wgrad / instnorm_* / prelu_* call's first input and output go to $DUMPDIR/NNN.pt, the call list to calls.json.
Run twice with the two settings to compare (e.g. MISEG_CONV3_NARROW=1 / 0), then diff the files call by call: the first call whose output
differs beyond rounding while its input does not is where the two runs part (round 4: one PReLU mask bit).
Usage: DUMPDIR=/tmp/d1 MISEG_CONV3_NARROW=1 python scripts/debug/unet_op_dump.py ; DUMPDIR=/tmp/d0 MISEG_CONV3_NARROW=0 python scripts/debug/unet_op_dump.py"""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import torch
import __graft_entry__ as ge
ge.load_package()
import conftest as C
from mi_seg_amd.networks.nets.unet import UNet
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.utils.detfill import det_input, fill_module_
from mi_seg_amd.hip import ops
T, R = C.Golden("unet_truth"), C.Golden("unet")
tag = "c1_64"
c = R.meta["cases"][tag]
norm = lambda n: parse_normalization(n, True, 4, 2)
m = UNet(3, 1, 6, channels=c["channels"], strides=c["strides"], num_res_units=c["num_res_units"], act="prelu", norm_down=norm(c.get("norm_down", "instance")), norm_up=norm("instance"),
         dropout=0.0, bias=True, adn_ordering="NDA")
fill_module_(m); m = m.to("cuda"); m.set_compute_dtype(torch.float32)
out_dir = os.environ["DUMPDIR"]; os.makedirs(out_dir, exist_ok=True)
calls = []
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        r = orig(*a, **k)
        t = r[0] if isinstance(r, tuple) else r
        x = a[0]
        i = len(calls)
        calls.append((name, tuple(x.shape), tuple(t.shape)))
        torch.save({"in": x.detach().float().cpu(), "out": t.detach().float().cpu()}, os.path.join(out_dir, f"{i:03d}.pt"))
        return r
    setattr(ops, name, f)
for n in ("conv3_fwd", "conv3_wgrad", "instnorm_fwd", "instnorm_bwd", "prelu_fwd", "prelu_bwd"):
    if hasattr(ops, n): wrap(n)
case = T.meta["cases"][tag]
y = m(det_input(1234, case["x"]).to("cuda"), case["modalities"])
y.backward(det_input(4321, tuple(y.shape)).to("cuda"))
torch.cuda.synchronize()
import json
json.dump(calls, open(os.path.join(out_dir, "calls.json"), "w"))
print(len(calls), "calls")
