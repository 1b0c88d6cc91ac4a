"""dump the C1 UNet's fp32 parameter gradients (tests/golden inputs) to a file: run once per library / environment setting and diff"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.networks.nets.unet import UNet
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.utils.detfill import det_input, fill_module_
inst = parse_normalization("instance", True, 4, 2)
m = UNet(3, 1, 6, channels=[32, 64, 128, 256], strides=[2, 2, 2], num_res_units=2, act="prelu", norm_down=inst, norm_up=parse_normalization("instance", True, 4, 2), dropout=0.0, bias=True, adn_ordering="NDA")
fill_module_(m); m = m.cuda().set_compute_dtype(torch.float32)
from mi_seg_amd.hip import functional as HF
REC = {}
_in, _pr = HF.instance_norm, HF.prelu
def rec_norm(x, *a, **k):
    y = _in(x, *a, **k)
    REC[f"norm{len(REC):02d}_{tuple(x.shape)}"] = y.detach().cpu()
    return y
def rec_prelu(x, w):
    y = _pr(x, w)
    def hook(g):
        REC[f"prelu_dy{len(REC):02d}_{tuple(x.shape)}"] = g.detach().cpu()
    y.register_hook(hook)
    return y
HF.instance_norm, HF.prelu = rec_norm, rec_prelu
y = m(det_input(1234, (1, 1, 64, 64, 64)).cuda(), None)
y.backward(det_input(4321, tuple(y.shape)).cuda())
torch.save({"y": y.detach().cpu(), **REC, **{k: p.grad.cpu() for k, p in m.named_parameters()}}, sys.argv[1])
if len(sys.argv) > 2:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[1])
    rows = sorted(((float((a[k] - b[k]).norm() / (a[k].norm() + 1e-30)), k) for k in a), reverse=True)
    rows = [r for r in rows if not r[1].endswith('conv.bias')]
    for e, k in rows[:14]:
        print(f"{e:.3e}  {k}  {tuple(a[k].shape)}")
