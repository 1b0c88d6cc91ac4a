"""__graft_entry__.smoke(): one small forward+backward of the hot path on cuda:0, checked against the CPU oracle."""
import os
import sys

import torch


def run():
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import nets as ON                      # checker only
    from oracle.functional import relative_position_index
    from ..networks.nets.swin_unetr import SwinUNETR
    from ..networks.norms.utils import parse_normalization
    from ..utils.detfill import det_input, fill_module_
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs cuda:0 (MI355X); the HIP path has no CPU fallback")
    cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
    m = SwinUNETR((64, 64, 64), 1, 6, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond,
                  decoder_norm_name=inst)
    fill_module_(m)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    m = m.to("cuda:0")
    x = det_input(7, (2, 1, 64, 64, 64))
    g = det_input(8, (2, 6, 64, 64, 64))
    mods = [1, 0]
    y = m(x.to("cuda:0"), mods)
    y.backward(g.to("cuda:0"))
    torch.cuda.synchronize()
    yo = ON.swin_unetr_forward(sd, x, mods, ON.swin_unetr_cfg(feature_size=12))
    yo.backward(g)
    err = float((y.cpu().double() - yo.double()).norm() / yo.double().norm())
    w = m.decoder1.conv_block.conv1.conv.weight.grad.cpu().double()
    wo = sd["decoder1.conv_block.conv1.conv.weight"].grad.double()
    gerr = float((w - wo).norm() / wo.norm())
    print(f"smoke: C-Swin-UNETR fs=12 64^3 batch 2 fwd+bwd on cuda:0 vs CPU oracle: logits rel err {err:.2e}, "
          f"decoder1 conv1 dW rel err {gerr:.2e}")
    if not (err < 1e-3 and gerr < 1e-3):
        raise AssertionError(f"smoke parity failed: {err} {gerr}")
