import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.runtime.arena import ParamArena
from mi_seg_amd.utils.detfill import fill_module_, det_input
from mi_seg_amd.hip import ops
dtype = torch.float32
N = lambda n: parse_normalization(n, True, 4, 2)
net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=N("instance_cond"), encoder_norm_name=N("instance_cond"), decoder_norm_name=N("instance")).cuda()
fill_module_(net); net.set_compute_dtype(dtype)
x = det_input(3, (2, 1, 64, 64, 64)).cuda(); cot = det_input(4, (2, 3, 64, 64, 64)).cuda()
names = [k for k, _ in net.named_parameters()]; params = [p for _, p in net.named_parameters()]
def plain(mods):
    for p in params: p.grad = None
    ops.begin_step(); y = net(x, mods); y.backward(cot)
    return y.detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in params]
def cmp(tag, ga, gb):
    rms = sorted(float(b.norm()) / b.numel() ** 0.5 for b in gb if b is not None)
    med = rms[len(rms) // 2]
    worst = sorted(((float((a - b).norm() / (b.norm() + 1e-20)), k) for k, a, b in zip(names, ga, gb)
                    if a is not None and b is not None and float(b.norm()) / b.numel() ** 0.5 > 1e-3 * med), reverse=True)[:4]
    print(tag, [(round(e, 6), k) for e, k in worst])
y1, g1 = plain([0, 0]); y2, g2 = plain([0, 0]); cmp("plain vs plain", g1, g2)
y3, g3 = plain([0, 1]); y4, g4 = plain([0, 1]); cmp("plain01 vs plain01", g3, g4)
arena = ParamArena(params, dtype)
for it, (mods, gr) in enumerate([([0, 0], g1), ([0, 0], g1), ([0, 1], g3), ([0, 1], g3)]):
    arena.begin_step(); y = net(x, mods); y.backward(cot); arena.publish()
    cmp(f"arena step {it} {mods}", [p.grad for p in params], gr)
