"""the C1 UNet's one-element (PReLU slope) gradients in bf16 against the float64 fixture and the reference's autocast run: one line per slope.
Run per environment setting (MISEG_CONV3_PAD_MIN=0 / 64 ...) to see how far a different summation order moves them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.networks.nets.unet import UNet
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.utils.detfill import det_input, fill_module_
T = np.load(os.path.join(ROOT, "tests/golden/unet_truth.npz"))
m = UNet(3, 1, 6, channels=[32, 64, 128, 256], strides=[2, 2, 2], num_res_units=2, act="prelu", norm_down=parse_normalization("instance", True, 4, 2),
         norm_up=parse_normalization("instance", True, 4, 2), dropout=0.0, bias=True, adn_ordering="NDA")
fill_module_(m); m = m.cuda().set_compute_dtype(torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] != "f32" else torch.float32)
y = m(det_input(1234, (1, 1, 64, 64, 64)).cuda(), None)
y.backward(det_input(4321, tuple(y.shape)).cuda())
vals = sorted(abs(float(T[k].reshape(-1)[0])) for k in T.files if k.startswith("c1_64/grad64:") and T[k].size == 1)
print("median |g| of the slopes", vals[len(vals) // 2])
for k, p in m.named_parameters():
    if p.numel() == 1:
        t, a = float(T[f"c1_64/grad64:{k}"].reshape(-1)[0]), float(T[f"c1_64/gradamp:{k}"].reshape(-1)[0])
        print(f"{k:55s} truth {t:10.3f}  autocast {a:10.3f}  here {float(p.grad):10.3f}")
