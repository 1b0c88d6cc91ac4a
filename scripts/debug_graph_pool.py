import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
fn = lambda: net(x, styles)
mode = sys.argv[1]
with torch.no_grad():
    ops.begin_step(); ref = fn(); ops.begin_step(); ref = fn()
    torch.cuda.synchronize()
    T_eager = ops.STAT_POOL.off
    pe = ops.STAT_POOL.buf.cpu().numpy().copy()
    ref = ref.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        off0 = ops.STAT_POOL.off
        ops.begin_step(); out = fn()
    print("T eager", T_eager, "off before capture begin_step", off0, "off after capture", ops.STAT_POOL.off, "pool@%x" % ops.STAT_POOL.buf.data_ptr())
    pools = []
    for rep in range(3):
        g.replay(); torch.cuda.synchronize()
        if mode == "bad":
            print("err", float((out.float() - ref.float()).norm() / ref.float().norm()))
        pools.append(ops.STAT_POOL.buf.cpu().numpy().copy())
    print("final err", float((out.float() - ref.float()).norm() / ref.float().norm()))
    for tag, a, b in (("eager vs r1", pe, pools[0]), ("r1 vs r2", pools[0], pools[1]), ("r2 vs r3", pools[1], pools[2])):
        idx = np.nonzero(a != b)[0]
        print(tag, "differing", idx.size, "range", (idx[0], idx[-1]) if idx.size else None)
        for i in idx[:6]:
            print("   [%d] %r -> %r" % (i, a[i], b[i]))
