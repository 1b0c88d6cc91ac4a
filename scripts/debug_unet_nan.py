import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.networks.nets.unet import UNet
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.utils.detfill import fill_module_, det_input
from mi_seg_amd.hip import functional as HF
N = lambda n: parse_normalization(n, True, 4, 2)
m = UNet(3, 1, 6, channels=[32, 64, 128, 256], strides=[2, 2, 2], num_res_units=2, act="prelu", norm_down=N("instance"), norm_up=N("instance"), dropout=0.0, bias=True).cuda()
fill_module_(m); m.set_compute_dtype(torch.bfloat16)
for name in ["conv3", "conv3_thin", "conv1", "conv3_transposed_weight", "subsample2", "upsample2_zero", "rowbias", "prelu", "instance_norm", "add", "cat_channels", "to_ncdhw"]:
    f = getattr(HF, name)
    def wrap(f=f, name=name):
        def g(*a, **k):
            y = f(*a, **k)
            t = y if isinstance(y, torch.Tensor) else y[0]
            bad = bool(torch.isnan(t.float()).any()) or bool(torch.isinf(t.float()).any())
            shp = [tuple(v.shape) for v in a if isinstance(v, torch.Tensor)]
            print(f"{name:26s} in {shp} -> {tuple(t.shape)} absmax {float(t.float().abs().max()):.3e} {'NaN/Inf!' if bad else ''}", flush=True)
            return y
        return g
    setattr(HF, name, wrap())
x = det_input(1234, (1, 1, 64, 64, 64)).cuda()
with torch.no_grad():
    y = m(x)
