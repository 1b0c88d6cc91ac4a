import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.networks.norms.conditional_instance_norm import styles_to_device
from mi_seg_amd.utils.detfill import fill_module_, det_input
from mi_seg_amd.hip import ops
N = lambda n: parse_normalization(n, True, 4, 2)
net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=N("instance_cond"), encoder_norm_name=N("instance_cond"), decoder_norm_name=N("instance")).cuda()
fill_module_(net); net.set_compute_dtype(torch.bfloat16)
x = det_input(3, (1, 1, 64, 64, 64)).cuda()
styles = styles_to_device([0], x.device, 1)
dt = torch.bfloat16
def graphed(fn, tag):
    with torch.no_grad():
        ops.begin_step(); ref = fn(); ops.begin_step(); ref = fn()
        torch.cuda.synchronize()
        ref = [t.clone() for t in (ref if isinstance(ref, (list, tuple)) else [ref])]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ops.begin_step(); out = fn()
        out = list(out) if isinstance(out, (list, tuple)) else [out]
        res = []
        for rep in range(3):
            g.replay(); torch.cuda.synchronize()
            res.append(["%.1e" % float((o.float() - r.float()).norm() / (r.float().norm() + 1e-30)) for o, r in zip(out, ref)])
        print(tag, res, flush=True)
def partial(k):
    hs_ = net.swinViT(x, net.normalize, styles, dt)
    outs = [hs_[4]]
    if k >= 1: a0 = net.encoder1(None, styles, image=x, dtype=dt); outs.append(a0)
    if k >= 2: a1 = net.encoder2(hs_[0], styles); outs.append(a1)
    if k >= 3: a2 = net.encoder3(hs_[1], styles); a3 = net.encoder4(hs_[2], styles); outs += [a2, a3]
    if k >= 4: b4 = net.encoder10(hs_[4], styles); b3 = net.decoder5(b4, hs_[3], styles); outs += [b4, b3]
    if k >= 5: b2 = net.decoder4(b3, a3, styles); b1 = net.decoder3(b2, a2, styles); outs += [b2, b1]
    if k >= 6: b0 = net.decoder2(b1, a1, styles); oo = net.decoder1(b0, a0, styles); outs += [b0, oo]
    if k >= 7: outs.append(net.out(oo))
    return outs
for k in range(7, 0, -1):
    graphed(lambda: partial(k)[-1], "p%d last only" % k)
