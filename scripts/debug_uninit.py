"""poison the caching allocator with NaNs between steps: any kernel that reads memory it (or a fill) did not write shows up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.utils.detfill import fill_module_, det_input
from mi_seg_amd.hip import ops
dtype = torch.float32 if len(sys.argv) < 2 or sys.argv[1] != "bf16" else torch.bfloat16
POISON = float(sys.argv[2]) if len(sys.argv) > 2 else float("nan")
N = lambda n: parse_normalization(n, True, 4, 2)
net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=N("instance_cond"), encoder_norm_name=N("instance_cond"), decoder_norm_name=N("instance")).cuda()
fill_module_(net); net.set_compute_dtype(dtype)
x = det_input(3, (2, 1, 64, 64, 64)).cuda(); cot = det_input(4, (2, 3, 64, 64, 64)).cuda()
names = [k for k, _ in net.named_parameters()]; params = [p for _, p in net.named_parameters()]
def poison():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    ts = [torch.full((1 << 28,), POISON, device="cuda") for _ in range(6)]   # 6 GiB of NaN
    torch.cuda.synchronize()
    del ts
def plain(mods):
    for p in params: p.grad = None
    ops.begin_step(); y = net(x, mods); y.backward(cot)
    return y.detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in params]
y1, g1 = plain([0, 0])
poison()
y2, g2 = plain([0, 0])
print("logits nan:", bool(torch.isnan(y2).any()), "rel diff", float((y2 - y1).norm() / y1.norm()))
bad = [k for k, g in zip(names, g2) if g is not None and bool(torch.isnan(g).any())]
print("params with NaN grads:", len(bad), bad[:12])
def cmp(tag, ga, gb):
    rms = sorted(float(b.norm()) / b.numel() ** 0.5 for b in gb if b is not None)
    med = rms[len(rms) // 2]
    worst = sorted(((float((a - b).norm() / (b.norm() + 1e-20)), k) for k, a, b in zip(names, ga, gb)
                    if a is not None and b is not None and float(b.norm()) / b.numel() ** 0.5 > 1e-3 * med), reverse=True)[:5]
    print(tag, [(round(e, 6), k) for e, k in worst])
cmp("run1 vs run2", g1, g2)
y3, g3 = plain([0, 0]); cmp("run2 vs run3", g2, g3)
y4, g4 = plain([0, 0]); cmp("run3 vs run4", g3, g4)
