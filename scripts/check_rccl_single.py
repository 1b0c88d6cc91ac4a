"""single-rank RCCL check of the N>1 code path of bench.py (ParamArena.allreduce after a graphed step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import __graft_entry__ as ge
ge.load_package()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.runtime.arena import ParamArena
from mi_seg_amd.runtime.graph import GraphedStep
from mi_seg_amd.utils.detfill import fill_module_, det_input
N = lambda n: parse_normalization(n, True, 4, 2)
net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=N("instance_cond"), encoder_norm_name=N("instance_cond"), decoder_norm_name=N("instance")).cuda()
fill_module_(net); net.set_compute_dtype(torch.bfloat16)
params = [p for p in net.parameters() if p.requires_grad]
arena = ParamArena(params, torch.bfloat16)
gs = GraphedStep(net, (1, 1, 64, 64, 64), (1, 3, 64, 64, 64), arena=arena)
x = det_input(3, (1, 1, 64, 64, 64)).cuda(); cot = det_input(4, (1, 3, 64, 64, 64)).cuda()
for it in range(3):
    gs(x, [it % 2], cot)
    before = arena.flat.clone()
    arena.allreduce(1)
    torch.cuda.synchronize()
    assert torch.equal(before, arena.flat), "all-reduce over one rank must be the identity"
    none = sum(p.grad is None for p in params)
    print("step", it, "modality", it % 2, "params without grad:", none, "grad norm", float(arena.flat.norm()))
dist.destroy_process_group()
print("RCCL single-rank path OK")
