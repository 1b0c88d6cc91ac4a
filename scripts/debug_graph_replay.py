import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.runtime.arena import ParamArena
from mi_seg_amd.runtime.graph import GraphedStep
from mi_seg_amd.utils.detfill import fill_module_, det_input
N = lambda n: parse_normalization(n, True, 4, 2)
net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=N("instance_cond"), encoder_norm_name=N("instance_cond"), decoder_norm_name=N("instance")).cuda()
fill_module_(net); net.set_compute_dtype(torch.bfloat16)
params = [p for p in net.parameters() if p.requires_grad]
names = [k for k, _ in net.named_parameters()]
arena = ParamArena(params, torch.bfloat16)
x = det_input(3, (1, 1, 64, 64, 64)).cuda(); cot = det_input(4, (1, 3, 64, 64, 64)).cuda()
def eager(m):
    arena.begin_step(); y = net(x, [m]); y.backward(cot); arena.publish(); torch.cuda.synchronize()
    return y.detach().clone(), arena.flat.clone()
ye0, ge0 = eager(0); ye0, ge0 = eager(0); ye1, ge1 = eager(1)
gs = GraphedStep(net, (1, 1, 64, 64, 64), (1, 3, 64, 64, 64), arena=arena)
for it, m in enumerate([0, 1, 0, 1, 0]):
    y = gs(x, [m], cot); torch.cuda.synchronize()
    ref_y, ref_g = (ye0, ge0) if m == 0 else (ye1, ge1)
    ey = float((y - ref_y).norm() / ref_y.norm()); eg = float((arena.flat - ref_g).norm() / ref_g.norm())
    bad = [(names[i], float(v.norm())) for i, v in enumerate(arena.views) if not torch.isfinite(v).all() or float(v.norm()) > 1e8][:5]
    print(f"step {it} modality {m}: logits rel err {ey:.3e}  arena rel err {eg:.3e}  |flat| {float(arena.flat.norm()):.4e}  suspicious {bad}")

arena.detach()
g2 = GraphedStep(net, (1, 1, 64, 64, 64), (1, 3, 64, 64, 64))
for seq in ([0, 0, 0, 0], [1, 1, 0, 0, 1]):
    errs = []
    for m in seq:
        y = g2(x, [m], cot); torch.cuda.synchronize()
        ref_y = ye0 if m == 0 else ye1
        errs.append("%.2e" % float((y - ref_y).norm() / ref_y.norm()))
    print(seq, errs)
# plain eager repeated, same tensors
from mi_seg_amd.hip import ops
for it in range(3):
    ops.begin_step(); y = net(x, [0]); torch.cuda.synchronize()
    print("eager", it, "%.2e" % float((y - ye0).norm() / ye0.norm()))
