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
with torch.no_grad():
    for ids in ([0, 1, 2, 3], [4, 5, 6, 7]):
        xb = torch.cat([win(i) for i in ids], 0)
        yb = m(xb, [1] * 4).clone()
        ys = [m(win(i), [1])[0].clone() for i in ids]
        ys2 = [m(win(i), [1])[0].clone() for i in ids]
        print("batch vs single", [round(rel(yb[j], ys[j]), 4) for j in range(4)], "single vs single again", [rel(ys[j], ys2[j]) for j in range(4)], "absmax", [float(y.abs().max()) for y in ys], [float(yb[j].abs().max()) for j in range(4)])
    arena = ParamArena(list(m.parameters()), torch.bfloat16)
    pred = GraphedForward(m, (4, 1, 96, 96, 96), arena=arena)
    for ids in ([0, 1, 2, 3], [4, 5, 6, 7], [0, 1, 2, 3]):
        xb = torch.cat([win(i) for i in ids], 0)
        yg = pred(xb, [1] * 4).clone()
        ye = m(xb, [1] * 4).clone()
        print("graph vs eager batch", ids, [round(rel(yg[j], ye[j]), 4) for j in range(4)], [float(yg[j].abs().max()) for j in range(4)])
    # the flow of tests/test_hip_training.py::test_full_volume_sliding_window_of_the_headline_model
    seen, n = {}, [0]
    picks = {0, 6, 48, 349, 350, 693, 699, 343, 57}
    def predictor(x, mods):
        y = pred(x, mods)
        for j in range(x.shape[0]):
            if n[0] + j in picks:
                seen[n[0] + j] = y[j].clone()
        n[0] += x.shape[0]
        return y
    out = inferer.sliding_window_inference(vol, (96, 96, 96), 4, predictor, overlap=0.5, modalities=[1])
    for i in sorted(picks):
        ye = m(win(i), [1])[0]
        d, h, w = grid[i]
        print("window", i, grid[i], "graphed-in-loop vs eager single", round(rel(seen[i], ye), 4), "absmax", float(seen[i].abs().max()), float(ye.abs().max()),
              "input absmax", float(win(i).abs().max()))
