"""cpu_baseline leg of bench.py: the CPU oracle (oracle/, a PyTorch fp32 restatement of the reference's path, pinned
against the reference's golden vectors) timed on the GPU box's host cores.  This is the checker being timed as a
baseline -- it is never part of the product path."""
import os
import sys
import time

import torch


def cpu_baseline():
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import nets as ON                     # noqa: E402  (bench.py cpu_baseline leg only)
    from oracle.functional import relative_position_index
    from ..networks.nets.swin_unetr import SwinUNETR
    from ..networks.norms.utils import parse_normalization
    from ..utils.detfill import det_input, det_values
    cores = min(16, os.cpu_count() or 1)     # the GPU box's CPU share for one GPU
    torch.set_num_threads(cores)
    cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
    with torch.device("meta"):
        m = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond,
                      decoder_norm_name=inst)
    sd = {}
    for k, v in m.state_dict().items():
        sd[k] = relative_position_index() if k.endswith("relative_position_index") else \
            torch.from_numpy(det_values(k, v.shape)).requires_grad_(True)
    cfg = ON.swin_unetr_cfg(feature_size=48)
    x = det_input(1234, (1, 1, 96, 96, 96))
    g = det_input(4321, (1, 6, 96, 96, 96))
    t0 = time.perf_counter()
    y = ON.swin_unetr_forward(sd, x, [0], cfg)
    y.backward(g)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 x 96^3 patch forward+backward (first call, no warm-up), oracle/nets.py swin_unetr_forward fp32, "
                      f"{dt:.1f} s"}
