"""one eager step of the headline model: which launches ran on the branch stream, which in background form (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.runtime.arena import ParamArena
cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
m = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst).cuda()
m.set_compute_dtype(torch.bfloat16)
arena = ParamArena([p for p in m.parameters() if p.requires_grad], torch.bfloat16)
log = []
orig = ops._call
def call(fn_name, params, prof=None, prof_params=None):
    main = torch.cuda.current_stream() != ops._BRANCH_STREAM if ops._BRANCH_STREAM is not None else True
    log.append((fn_name, "M" if main else "B", getattr(params, "background", None), getattr(params, "max_workgroups", None), torch._C._current_graph_task_id()))
    return orig(fn_name, params, prof, prof_params)
ops._call = call
x = torch.randn(1, 1, 96, 96, 96, device="cuda")
for it in range(2):
    log.clear()
    arena.begin_step()
    y = m(x, [0])
    y.backward(torch.randn_like(y))
    arena.publish()
torch.cuda.synchronize()
for i, e in enumerate(log):
    if e[1] == "B":
        print(i, e)
print("calls", len(log))
