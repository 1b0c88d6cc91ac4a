"""Distance of the HIP path's C2 parameter gradients from the reference's float64 run, next to the reference's own fp32 / autocast-bf16
distances (tests/golden/swin_unetr_c2_truth.npz, swin_unetr_c2.npz).  Prints the distributions; the thresholds of
tests/test_hip_modules.py::test_swin_unetr_c2_vs_truth come from here.

    python scripts/c2_vs_truth.py [f32|bf16 ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from conftest import sample  # noqa: E402
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR  # noqa: E402
from mi_seg_amd.networks.norms.utils import parse_normalization  # noqa: E402
from mi_seg_amd.utils.detfill import ce_cotangent, det_input, fill_module_  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def main():
    T = np.load(os.path.join(ROOT, "tests/golden/swin_unetr_c2_truth.npz"))
    R = np.load(os.path.join(ROOT, "tests/golden/swin_unetr_c2.npz"))
    cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
    m = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst)
    fill_module_(m)
    m = m.cuda()
    x = det_input(1234, (1, 1, 96, 96, 96)).cuda()
    for mode in sys.argv[1:] or ["f32", "bf16"]:
        m.set_compute_dtype(torch.float32 if mode == "f32" else torch.bfloat16)
        for cot, (k32, k64, kamp) in {"noise": ("grad:", "grad64:", "gradamp:"), "ce": ("grad2:", "grad64_2:", "gradamp_2:")}.items():
            m.zero_grad(set_to_none=True)
            y = m(x, [0])
            y.backward(ce_cotangent(y) if cot == "ce" else det_input(4321, tuple(y.shape)).cuda())
            got = {k: sample(p.grad).numpy() for k, p in m.named_parameters() if p.grad is not None}
            keys = [k[len("c2_m0/" + k64):] for k in T.files if k.startswith("c2_m0/" + k64)]
            rms = {k: np.linalg.norm(T["c2_m0/" + k64 + k].astype(np.float64)) / np.sqrt(T["c2_m0/" + k64 + k].size) for k in keys}
            med = sorted(rms.values())[len(rms) // 2]
            live = [k for k in keys if rms[k] > 1e-3 * med]
            rows = []
            for k in live:
                t = T["c2_m0/" + k64 + k]
                rows.append((rel(got[k], t), rel(R["c2_m0/" + k32 + k], t), rel(T["c2_m0/" + kamp + k], t), rel(got[k], R["c2_m0/" + k32 + k]), k))
            print(f"== {mode} / {cot}: logits vs fp64 {rel(sample(y).numpy(), T['c2_m0/logits64_samples']):.2e} (ref32 {rel(R['c2_m0/logits_samples'], T['c2_m0/logits64_samples']):.2e}, "
                  f"refamp {rel(T['c2_m0/logitsamp_samples'], T['c2_m0/logits64_samples']):.2e})")
            for name, col in (("hip vs fp64", 0), ("ref32 vs fp64", 1), ("refamp vs fp64", 2), ("hip vs ref32", 3)):
                v = sorted(r_[col] for r_ in rows)
                print(f"   {name:15s} median {v[len(v) // 2]:.2e}  p90 {v[int(len(v) * .9)]:.2e}  max {v[-1]:.2e}")
            ref = 1 if mode == "f32" else 2
            ratio = sorted(((r_[0] / (r_[ref] + 1e-12)), r_[0], r_[ref], r_[4]) for r_ in rows)
            print("   worst ratios hip/ref at equal precision:")
            for q in ratio[-8:]:
                print(f"      {q[0]:.2f}  hip {q[1]:.2e}  ref {q[2]:.2e}  {q[3]}")
            dead = [k for k in keys if k not in live]
            dmax = max(float(np.linalg.norm(got[k]) / np.sqrt(got[k].size)) for k in dead) if dead else 0.0
            print(f"   vanishing-gradient parameters: {len(dead)}, worst rms {dmax:.2e} against the median live rms {med:.2e}")
            sys.stdout.flush()


if __name__ == "__main__":
    main()
