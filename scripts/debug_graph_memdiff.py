"""dump every tensor allocated during capture after replay 1 and after replay 2 and report which differ (graph replay debugging)."""
import os, sys, ctypes, traceback
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
hip = ctypes.CDLL("libamdhip64.so")
N = lambda n: parse_normalization(n, True, 4, 2)
net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=N("instance_cond"), encoder_norm_name=N("instance_cond"), decoder_norm_name=N("instance")).cuda()
fill_module_(net); net.set_compute_dtype(torch.bfloat16)
x = det_input(3, (1, 1, 64, 64, 64)).cuda()
styles = styles_to_device([0], x.device, 1)
fn = lambda: net(x, styles)
LOG = []
real_empty = torch.empty
def logging_empty(*a, **k):
    t = real_empty(*a, **k)
    if t.is_cuda:
        fr = [f for f in traceback.extract_stack(limit=6)[:-1]]
        LOG.append((len(LOG), t.data_ptr(), t.numel() * t.element_size(), str(t.dtype), " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr[-3:]))))
    return t
def dump():
    torch.cuda.synchronize()
    res = []
    for seq, ptr, nb, dt, where in LOG:
        h = np.empty(nb, dtype=np.uint8)
        rc = hip.hipMemcpy(ctypes.c_void_p(h.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nb), 2)
        assert rc == 0, rc
        res.append(h)
    pool = ops.STAT_POOL.buf.cpu().numpy().copy()
    return res, pool
with torch.no_grad():
    ops.begin_step(); ref = fn(); ops.begin_step(); ref = fn()
    torch.cuda.synchronize()
    ref = ref.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        torch.empty = logging_empty
        ops.begin_step(); out = fn()
        torch.empty = real_empty
    print("allocs in capture:", len(LOG), "bytes", sum(l[2] for l in LOG))
    mode = sys.argv[1]
    g.replay(); torch.cuda.synchronize()
    if mode == "bad":
        e1 = float((out.float() - ref.float()).norm() / ref.float().norm())
    g.replay(); torch.cuda.synchronize()
    d2, p2 = dump()
    e2 = float((out.float() - ref.float()).norm() / ref.float().norm())
    print(mode, "err after replay 2:", e2)
    import pickle
    pickle.dump((LOG, d2, p2), open("/tmp/dump_%s.pkl" % mode, "wb"), protocol=4)
