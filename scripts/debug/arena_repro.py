"""arena mode vs plain autograd on small nets, repeated in one process in the order pytest runs test_param_arena_matches_plain_autograd:
the logits must be bit-identical (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.nets.unetr import UNETR
from mi_seg_amd.networks.nets.unet import UNet
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.runtime.arena import ParamArena
from mi_seg_amd.utils.detfill import fill_module_, det_input
cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for dtype in (torch.float32, torch.bfloat16):
        for kind in ("swin_unetr", "unetr", "unetr_conv", "unet"):
            torch.manual_seed(0)
            S = 64 if kind == "swin_unetr" else 32
            if kind == "swin_unetr":
                net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst).cuda()
            elif kind.startswith("unetr"):
                net = UNETR(1, 3, (32, 32, 32), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4, pos_embed="conv" if kind == "unetr_conv" else "perceptron",
                            vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst).cuda()
            else:
                net = UNet(3, 1, 3, channels=(8, 16, 32), strides=(2, 2), num_res_units=2, norm_down=cond, norm_up=inst).cuda()
            fill_module_(net)
            net.set_compute_dtype(dtype)
            x = det_input(3, (2, 1, S, S, S)).cuda()
            cot = det_input(4, (2, 3, S, S, S)).cuda()
            params = [p for p in net.parameters() if p.requires_grad]

            def plain(mods):
                for p in params:
                    p.grad = None
                ops.begin_step()
                y = net(x, mods)
                y.backward(cot)
                return y.detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in params]

            refs = {(0, 0): plain([0, 0]), (0, 1): plain([0, 1])}
            arena = ParamArena(params, dtype)
            try:
                for it, mods in enumerate([(0, 0), (0, 0), (0, 1)]):
                    arena.begin_step()
                    y = net(x, list(mods))
                    y.backward(cot)
                    arena.publish()
                    d = float((y.detach() - refs[mods][0]).abs().max())
                    host = [g.float().cpu() for g in refs[mods][1] if g is not None]      # host reads like compare_grads
                    got = [p.grad.float().cpu() for p in params if p.grad is not None]
                    if d != 0.0:
                        bad += 1
                        print(rep, kind, dtype, it, mods, "max |logit diff|", d, flush=True)
            finally:
                arena.detach()
print("reps done, mismatches:", bad)
