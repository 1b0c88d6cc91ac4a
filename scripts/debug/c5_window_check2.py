import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.runtime.arena import ParamArena
from mi_seg_amd.runtime.graph import GraphedForward
from mi_seg_amd.utils.detfill import fill_module_
from mi_seg_amd.training import inferer
cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
m = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst)
fill_module_(m); m = m.cuda().set_compute_dtype(torch.bfloat16).eval()
vol = torch.rand(1, 1, 512, 512, 363, generator=torch.Generator().manual_seed(77)).cuda()
grid = inferer.window_grid((512, 512, 363), (96, 96, 96), 0.5)
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
def win(i):
    d, h, w = grid[i]
    return vol[:, :, d:d + 96, h:h + 96, w:w + 96].contiguous()
from mi_seg_amd.hip import ops as _ops
if os.environ.get("BIGPOOL"):
    _ops.STAT_POOL.numel = 1 << 25
if os.environ.get("EAGER_FIRST"):
    with torch.no_grad():
        m(torch.cat([win(i) for i in (0, 1, 2, 3)], 0), [1] * 4)
arena = ParamArena(list(m.parameters()), torch.bfloat16)
pred = GraphedForward(m, (4, 1, 96, 96, 96), arena=arena)
mode = sys.argv[1] if len(sys.argv) > 1 else "direct"
with torch.no_grad():
    outs = []
    for k, ids in enumerate(([0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11])):
        xb = torch.cat([win(i) for i in ids], 0)
        outs.append(pred(xb, [1] * 4).clone())
        if mode == "alloc" and k == 0:
            big = torch.empty(700, 6, 96, 96, 96, device="cuda"); big2 = torch.empty(1, 6, 512, 512, 363, device="cuda")
    for k, ids in enumerate(([0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11])):
        xb = torch.cat([win(i) for i in ids], 0)
        ye = m(xb, [1] * 4)
        print(mode, ids, [round(rel(outs[k][j], ye[j]), 4) for j in range(4)], [round(float(outs[k][j].abs().max()), 2) for j in range(4)])
with torch.no_grad():
    xa = torch.cat([win(i) for i in (0, 1, 2, 3)], 0)
    r = [pred(xa, [1] * 4).clone() for _ in range(3)]
    ye = m(xa, [1] * 4)
    print(mode, "same input thrice: replay k vs eager", [round(rel(r[k], ye), 4) for k in range(3)], "replay1 == replay2", bool(torch.equal(r[1], r[2])))
