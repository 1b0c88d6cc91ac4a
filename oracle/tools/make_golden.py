#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU.

Run in the build container only (needs /root/reference, which never travels):

    python oracle/tools/make_golden.py [--only NAME ...] [--skip-full]

What is imported from where
  * /root/reference/networks/...   the reference modules, by path, at run time (never copied)
  * oracle/tools/monai_standin     the build's own stand-in for the MONAI helper symbols the reference
                                   imports (MONAI is not installed here)
  * mi-seg_amd/utils/detfill.py    name-keyed deterministic weights, shared with oracle + product

Fixtures hold inputs, expected outputs and gradients (data only).  Arithmetic that lives in the
stand-in rather than in the reference (MLPBlock, SABlock, DropPath) is listed under the
``unpinned`` key of each fixture's metadata.
"""
import argparse
import hashlib
import json
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MISEG_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "monai_standin"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from mi_seg_amd.utils.detfill import ce_cotangent, det_input, fill_module_  # noqa: E402

from networks.blocks.dynunet_block import UnetResBlock  # noqa: E402  (reference)
from networks.blocks.patch_merging import PatchMerging, PatchMergingV2  # noqa: E402
from networks.blocks.swin_transformer_block import SwinTransformerBlock  # noqa: E402
from networks.blocks.transformer_block import TransformerBlock  # noqa: E402
from networks.blocks.unetr_block import UnetrPrUpBlock, UnetrUpBlock  # noqa: E402
from networks.blocks.window_attention import WindowAttention  # noqa: E402
from networks.nets.swin_unetr import SwinUNETR  # noqa: E402
from networks.nets.unet import UNet  # noqa: E402
from networks.nets.unetr import UNETR  # noqa: E402
from networks.norms.conditional_instance_norm import ConditionalInstanceNorm1d, ConditionalInstanceNorm3d  # noqa: E402
from networks.norms.utils import parse_normalization  # noqa: E402
from networks.utils.swin_utils import compute_mask, get_window_size  # noqa: E402

# ---- torch CPU bug workaround (generator process only; the reference's files are untouched) ----------
# On torch 2.10 CPU, F.instance_norm's backward returns WRONG input/weight gradients whenever grad_output
# arrives non-contiguous (torch.autograd.gradcheck fails on
#     F.instance_norm(x).permute(0, 2, 3, 4, 1)
# in float64).  The reference reaches exactly that state around every non-LayerNorm norm in its Swin /
# ViT blocks (`rearrange` right after the norm, swin_transformer_block.py:107-112), so un-patched
# fixtures would pin a torch bug, not the reference's mathematics (its forward values are unaffected,
# and the authors' CUDA stack computes the correct gradient).  The wrapper below only makes the incoming
# gradient contiguous; `--check-torch-bug` prints the gradcheck evidence with and without it.
_orig_instance_norm = torch.nn.functional.instance_norm


class _ContiguousGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def _instance_norm_contiguous_grad(input, *args, **kwargs):
    return _ContiguousGrad.apply(_orig_instance_norm(input, *args, **kwargs))


torch.nn.functional.instance_norm = _instance_norm_contiguous_grad


def check_torch_bug():
    m = ConditionalInstanceNorm3d(2, 4).double()
    x = torch.randn(2, 3, 4, 5, 4, dtype=torch.float64, requires_grad=True)
    f = lambda t: m(t.permute(0, 4, 1, 2, 3), [1, 0]).permute(0, 2, 3, 4, 1)  # noqa: E731
    torch.nn.functional.instance_norm = _orig_instance_norm
    print("reference cond-norm on a rearranged tensor, stock torch : gradcheck =",
          torch.autograd.gradcheck(f, (x,), raise_exception=False))
    torch.nn.functional.instance_norm = _instance_norm_contiguous_grad
    print("same, incoming gradient made contiguous               : gradcheck =",
          torch.autograd.gradcheck(f, (x,), raise_exception=False))


OUT = os.path.join(ROOT, "tests", "golden")
NSAMP = 4096
torch.set_num_threads(8)


def np32(t):
    return t.detach().to(torch.float32).cpu().numpy()


def sample(t, n=NSAMP):
    """Deterministic strided sample of a tensor (all of it when small)."""
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return np32(f)
    idx = torch.linspace(0, f.numel() - 1, n).round().long()
    return np32(f[idx])


def digest(t):
    return hashlib.sha256(np.ascontiguousarray(np32(t)).tobytes()).hexdigest()


def grads_of(module, full=True):
    """name -> grad (full or sampled); plus the list of parameters whose grad is None."""
    out, none = {}, []
    for k, p in module.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            out["grad:" + k] = np32(p.grad) if full else sample(p.grad)
            if not full:
                out["gnorm:" + k] = np.float64(p.grad.double().norm().item())
    return out, none


def save(name, arrays, meta):
    os.makedirs(OUT, exist_ok=True)
    arrays = dict(arrays)
    arrays["__meta__"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


COND = parse_normalization("instance_cond", True, 4, 2)
INST = parse_normalization("instance", True, 4, 2)
LAYER = parse_normalization("layer", True, 4, 2)


# ------------------------------------------------------------------------------------------
def gen_cond_instnorm():
    arrays, meta = {}, {"unpinned": [], "cases": {}}
    for tag, cls, shape, styles in (
        ("3d_mixed", ConditionalInstanceNorm3d, (2, 6, 5, 6, 7), [1, 0]),
        ("3d_same", ConditionalInstanceNorm3d, (2, 6, 5, 6, 7), [0, 0]),
        ("3d_tensor_styles", ConditionalInstanceNorm3d, (3, 4, 3, 4, 5), torch.tensor([1, 1, 0])),
        ("1d_mixed", ConditionalInstanceNorm1d, (2, 8, 10), [0, 1]),
    ):
        m = cls(2, shape[1])
        fill_module_(m)
        x = det_input(11, shape).requires_grad_(True)
        g = det_input(12, shape)
        y = m(x, styles)
        y.backward(g)
        gr, none = grads_of(m)
        arrays.update({f"{tag}/x": np32(x), f"{tag}/g": np32(g), f"{tag}/y": np32(y), f"{tag}/dx": np32(x.grad)})
        arrays.update({f"{tag}/{k}": v for k, v in gr.items()})
        meta["cases"][tag] = {"shape": list(shape), "styles": [int(s) for s in styles], "grad_none": none}
    save("cond_instnorm", arrays, meta)


def gen_window_attention():
    arrays, meta = {}, {"unpinned": [], "cases": {}}
    dim, heads = 12, 3
    m = WindowAttention(dim, heads, (7, 7, 7), qkv_bias=True)
    fill_module_(m)
    mask = compute_mask([14, 7, 7], (7, 7, 7), (3, 3, 3), "cpu")          # [2, 343, 343]
    for tag, n, b, msk in (("n343_nomask", 343, 2, None), ("n343_mask", 343, 4, mask), ("n216_sliced", 216, 2, None)):
        m.zero_grad()
        x = det_input(21, (b, n, dim)).requires_grad_(True)
        g = det_input(22, (b, n, dim))
        y = m(x, msk)
        y.backward(g)
        gr, none = grads_of(m)
        arrays.update({f"{tag}/x": np32(x), f"{tag}/g": np32(g), f"{tag}/y": np32(y), f"{tag}/dx": np32(x.grad)})
        arrays.update({f"{tag}/{k}": v for k, v in gr.items()})
        meta["cases"][tag] = {"n": n, "b": b, "dim": dim, "heads": heads, "mask": msk is not None}
    arrays["mask_14_7_7"] = np32(mask)
    arrays["relative_position_index"] = m.relative_position_index.numpy().astype(np.int64)
    save("window_attention", arrays, meta)


def gen_swin_block():
    arrays, meta = {}, {"unpinned": ["mlp (MONAI MLPBlock stand-in)"], "cases": {}}
    for tag, dhw, shift, norm in (
        ("pad_noshift", (10, 10, 10), (0, 0, 0), COND),
        ("pad_shift", (10, 10, 10), (3, 3, 3), COND),
        ("clamped6", (6, 6, 6), (3, 3, 3), COND),
        ("layer_shift", (8, 9, 10), (3, 3, 3), LAYER),
        ("inst_noshift", (7, 7, 14), (0, 0, 0), INST),
    ):
        dim, heads = 12, 3
        import copy
        m = SwinTransformerBlock(dim, heads, (7, 7, 7), shift, norm_type=copy.deepcopy(norm))
        fill_module_(m)
        ws, ss = get_window_size(dhw, (7, 7, 7), (3, 3, 3))
        pads = [int(np.ceil(d / w)) * w for d, w in zip(dhw, ws)]
        mask = compute_mask(pads, ws, ss, "cpu")
        x = det_input(31, (2,) + dhw + (dim,)).requires_grad_(True)
        g = det_input(32, (2,) + dhw + (dim,))
        mod = [1, 0]
        y = m(x, mask, modalities=mod)
        y.backward(g)
        gr, none = grads_of(m)
        arrays.update({f"{tag}/x": np32(x), f"{tag}/g": np32(g), f"{tag}/y": np32(y), f"{tag}/dx": np32(x.grad)})
        arrays.update({f"{tag}/{k}": v for k, v in gr.items()})
        meta["cases"][tag] = {"dhw": list(dhw), "shift": list(shift), "norm": norm[0], "dim": dim, "heads": heads,
                              "modalities": mod, "grad_none": none}
    save("swin_block", arrays, meta)


def gen_patch_merging():
    arrays, meta = {}, {"unpinned": [], "cases": {}}
    import copy
    for tag, cls, dhw, norm in (("merging", PatchMerging, (6, 6, 6), COND), ("mergingv2", PatchMergingV2, (6, 6, 6), COND),
                                ("merging_odd_layer", PatchMerging, (5, 6, 7), LAYER)):
        m = cls(4, norm_type=copy.deepcopy(norm), spatial_dims=3)
        fill_module_(m)
        x = det_input(41, (2,) + dhw + (4,)).requires_grad_(True)
        y = m(x, modalities=[0, 1])
        g = det_input(42, tuple(y.shape))
        y.backward(g)
        gr, none = grads_of(m)
        arrays.update({f"{tag}/x": np32(x), f"{tag}/g": np32(g), f"{tag}/y": np32(y), f"{tag}/dx": np32(x.grad)})
        arrays.update({f"{tag}/{k}": v for k, v in gr.items()})
        meta["cases"][tag] = {"dhw": list(dhw), "norm": norm[0], "modalities": [0, 1], "grad_none": none}
    save("patch_merging", arrays, meta)


def gen_unetr_blocks():
    arrays, meta = {}, {"unpinned": [], "cases": {}}
    cases = (
        ("res_8_to_12_cond", lambda: UnetResBlock(3, 8, 12, 3, 1, COND), (2, 8, 8, 8, 8), None),
        ("res_8_to_8_cond", lambda: UnetResBlock(3, 8, 8, 3, 1, COND), (2, 8, 8, 8, 8), None),
        ("res_1_to_8_inst", lambda: UnetResBlock(3, 1, 8, 3, 1, INST), (2, 1, 8, 9, 10), None),
        ("up_16_to_8_inst", lambda: UnetrUpBlock(3, 16, 8, 3, 2, INST, res_block=True), (2, 16, 4, 4, 4), (2, 8, 8, 8, 8)),
        ("prup_16_to_8_cond", lambda: UnetrPrUpBlock(3, 16, 8, 1, 3, 1, 2, COND, conv_block=True, res_block=True),
         (2, 16, 3, 3, 3), None),
    )
    for tag, make, xs, ss in cases:
        m = make()
        fill_module_(m)
        x = det_input(51, xs).requires_grad_(True)
        mod = [1, 0]
        if ss is not None:
            skip = det_input(53, ss).requires_grad_(True)
            y = m(x, skip, mod)
        else:
            skip = None
            y = m(x, mod)
        g = det_input(52, tuple(y.shape))
        y.backward(g)
        gr, none = grads_of(m)
        arrays.update({f"{tag}/x": np32(x), f"{tag}/g": np32(g), f"{tag}/y": np32(y), f"{tag}/dx": np32(x.grad)})
        if skip is not None:
            arrays.update({f"{tag}/skip": np32(skip), f"{tag}/dskip": np32(skip.grad)})
        arrays.update({f"{tag}/{k}": v for k, v in gr.items()})
        meta["cases"][tag] = {"x": list(xs), "modalities": mod, "grad_none": none}
    save("unetr_blocks", arrays, meta)


def gen_transformer_block():
    arrays, meta = {}, {"unpinned": ["attn (MONAI SABlock stand-in)", "mlp (MONAI MLPBlock stand-in)"], "cases": {}}
    import copy
    for tag, norm, qkvb in (("cond", COND, False), ("layer_bias", LAYER, True)):
        m = TransformerBlock(32, 64, 4, 0.0, qkvb, norm_type=copy.deepcopy(norm))
        fill_module_(m)
        x = det_input(61, (2, 27, 32)).requires_grad_(True)
        g = det_input(62, (2, 27, 32))
        y = m(x, [0, 1])
        y.backward(g)
        gr, none = grads_of(m)
        arrays.update({f"{tag}/x": np32(x), f"{tag}/g": np32(g), f"{tag}/y": np32(y), f"{tag}/dx": np32(x.grad)})
        arrays.update({f"{tag}/{k}": v for k, v in gr.items()})
        meta["cases"][tag] = {"hidden": 32, "mlp": 64, "heads": 4, "qkv_bias": qkvb, "norm": norm[0],
                              "modalities": [0, 1], "grad_none": none}
    save("transformer_block", arrays, meta)


def whole_net(tag, model, xshape, modalities, meta_extra, full_grads, unpinned, arrays_out, meta_out, second=False):
    fill_module_(model)
    x = det_input(1234, xshape)
    t0 = time.time()
    y = model(x, modalities)
    t1 = time.time()
    g = det_input(4321, tuple(y.shape))
    y.backward(g, retain_graph=second)
    t2 = time.time()
    gr, none = grads_of(model, full=full_grads)
    if second:   # same forward, spatially coherent cotangent (mean cross-entropy against block labels)
        model.zero_grad(set_to_none=True)
        y.backward(ce_cotangent(y))
        gr2, _ = grads_of(model, full=False)
        arrays_out.update({f"{tag}/{k.replace('grad:', 'grad2:').replace('gnorm:', 'gnorm2:')}": v for k, v in gr2.items()})
    arrays_out.update({f"{tag}/logits_samples": sample(y), f"{tag}/logits_l2": np.float64(y.double().norm().item())})
    if y.numel() <= 200_000:
        arrays_out[f"{tag}/logits"] = np32(y)
    arrays_out.update({f"{tag}/{k}": v for k, v in gr.items()})
    meta_out["cases"][tag] = dict(meta_extra, x=list(xshape), modalities=None if modalities is None else
                                  [int(m) for m in modalities], grad_none=none, logits_sha256=digest(y),
                                  n_params=sum(p.numel() for p in model.parameters()),
                                  n_state=len(model.state_dict()), cpu_fwd_s=round(t1 - t0, 3),
                                  cpu_bwd_s=round(t2 - t1, 3), state_keys=list(model.state_dict().keys()),
                                  state_shapes=[list(v.shape) for v in model.state_dict().values()])
    meta_out["unpinned"] = unpinned
    print(f"    {tag}: fwd {t1 - t0:.2f}s bwd {t2 - t1:.2f}s params {meta_out['cases'][tag]['n_params']}")


def gen_swin_unetr_small():
    import copy
    arrays, meta = {}, {"cases": {}}
    for tag, mods, ds, vit, enc in (("fs12_64_m10", [1, 0], "merging", COND, COND),
                                    ("fs12_64_v2_layer", [0, 0], "mergingv2", LAYER, INST)):
        m = SwinUNETR((64, 64, 64), 1, 6, feature_size=12, num_heads=(3, 6, 12, 24), downsample=ds,
                      vit_norm_name=copy.deepcopy(vit), encoder_norm_name=copy.deepcopy(enc),
                      decoder_norm_name=copy.deepcopy(INST))
        whole_net(tag, m, (2, 1, 64, 64, 64), mods, {"feature_size": 12, "downsample": ds, "vit_norm": vit[0],
                                                      "encoder_norm": enc[0], "decoder_norm": "instance"},
                  False, ["swinViT.*.mlp (MONAI MLPBlock stand-in)"], arrays, meta)
    save("swin_unetr_small", arrays, meta)


def gen_swin_unetr_c2():
    import copy
    arrays, meta = {}, {"cases": {}}
    for tag, mods in (("c2_m0", [0]), ("c2_m1", [1])):
        m = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24),
                      vit_norm_name=copy.deepcopy(COND), encoder_norm_name=copy.deepcopy(COND),
                      decoder_norm_name=copy.deepcopy(INST))
        whole_net(tag, m, (1, 1, 96, 96, 96), mods, {"feature_size": 48, "downsample": "merging",
                                                      "vit_norm": "instance_cond", "encoder_norm": "instance_cond",
                                                      "decoder_norm": "instance"},
                  False, ["swinViT.*.mlp (MONAI MLPBlock stand-in)"], arrays, meta, second=(tag == "c2_m0"))
    save("swin_unetr_c2", arrays, meta)


def gen_swin_unetr_c2_truth():
    """The conditioning of the C2 gradient comparison, measured on the REFERENCE itself: the same modules, weights, input and cotangents
    as `swin_unetr_c2` case c2_m0, run (a) in float64 = the "truth" both fp32 implementations approximate, and (b) under
    torch.autocast(cpu, bfloat16) = the reference's own mixed-precision mode (its default: utils/trainer.py:34 autocast(enabled=amp),
    tune.py:267 amp = not no_amp; fp16 on CUDA there, bf16 is what this CPU offers).  The parameter gradients of this net are a
    discontinuous function of the forward pass (LeakyReLU(0.01) sign flips of instance-normalised pre-activations), so a forward
    deviation eps moves them by ~sqrt(eps): tests bound the HIP path's distance from the float64 run by the reference's OWN distance
    from it at the same precision (tests/test_hip_modules.py::test_swin_unetr_c2_vs_truth)."""
    import copy
    arrays, meta = {}, {"cases": {}}
    m = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=copy.deepcopy(COND),
                  encoder_norm_name=copy.deepcopy(COND), decoder_norm_name=copy.deepcopy(INST))
    fill_module_(m)
    x = det_input(1234, (1, 1, 96, 96, 96))
    g = det_input(4321, (1, 6, 96, 96, 96))
    tag = "c2_m0"

    def grab(model, key):
        for k, p in model.named_parameters():
            if p.grad is not None:
                arrays[f"{tag}/{key}:{k}"] = sample(p.grad)
        model.zero_grad(set_to_none=True)

    t0 = time.time()
    md = copy.deepcopy(m).double()
    yd = md(x.double(), [0])
    yd.backward(g.double(), retain_graph=True)
    grab(md, "grad64")
    yd.backward(ce_cotangent(yd).double())
    grab(md, "grad64_2")
    arrays[f"{tag}/logits64_samples"] = sample(yd)
    t1 = time.time()
    del md
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ya = m(x, [0])
    ya.backward(g.to(ya.dtype), retain_graph=True)
    grab(m, "gradamp")
    ya.backward(ce_cotangent(ya.float()).to(ya.dtype))
    grab(m, "gradamp_2")
    arrays[f"{tag}/logitsamp_samples"] = sample(ya.float())
    t2 = time.time()
    meta["cases"][tag] = {"x": [1, 1, 96, 96, 96], "modalities": [0], "fp64_s": round(t1 - t0, 1), "autocast_bf16_s": round(t2 - t1, 1),
                          "autocast_logits_dtype": str(ya.dtype)}
    meta["unpinned"] = ["swinViT.*.mlp (MONAI MLPBlock stand-in)"]
    print(f"    fp64 {t1 - t0:.0f}s autocast {t2 - t1:.0f}s")
    save("swin_unetr_c2_truth", arrays, meta)


def truth_case(arrays, meta, tag, build, xshape, modalities, unpinned):
    """float64 and autocast-bf16 runs of the reference modules `build()` yields (same weights / input / white-noise cotangent as the
    fp32 case `tag` of the companion fixture): the conditioning baseline of tests/test_hip_modules.py::_vs_truth"""
    m = build()
    fill_module_(m)
    x = det_input(1234, xshape)
    t0 = time.time()
    md = build()
    fill_module_(md)
    md = md.double()
    yd = md(x.double(), modalities)
    g = det_input(4321, tuple(yd.shape))
    yd.backward(g.double())
    for k, p in md.named_parameters():
        if p.grad is not None:
            arrays[f"{tag}/grad64:{k}"] = sample(p.grad)
    arrays[f"{tag}/logits64_samples"] = sample(yd)
    t1 = time.time()
    del md
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ya = m(x, modalities)
    ya.backward(g.to(ya.dtype))
    for k, p in m.named_parameters():
        if p.grad is not None:
            arrays[f"{tag}/gradamp:{k}"] = sample(p.grad)
    arrays[f"{tag}/logitsamp_samples"] = sample(ya.float())
    t2 = time.time()
    meta["cases"][tag] = {"x": list(xshape), "modalities": None if modalities is None else [int(v) for v in modalities], "fp64_s": round(t1 - t0, 1),
                          "autocast_bf16_s": round(t2 - t1, 1), "autocast_logits_dtype": str(ya.dtype)}
    meta["unpinned"] = unpinned
    print(f"    {tag}: fp64 {t1 - t0:.0f}s autocast {t2 - t1:.0f}s")


def gen_unetr_c3_truth():
    """C3 (and the small UNETR) in float64 and under autocast-bf16: see gen_swin_unetr_c2_truth"""
    import copy
    arrays, meta = {}, {"cases": {}}
    up = ["vit.blocks.*.attn (MONAI SABlock stand-in)", "vit.blocks.*.mlp (MONAI MLPBlock stand-in)"]
    truth_case(arrays, meta, "small_32", lambda: UNETR(1, 6, (32, 32, 32), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4, pos_embed="perceptron",
                                                        vit_norm_name=copy.deepcopy(COND), encoder_norm_name=copy.deepcopy(COND),
                                                        decoder_norm_name=copy.deepcopy(INST)), (2, 1, 32, 32, 32), [0, 1], up)
    truth_case(arrays, meta, "c3_m1", lambda: UNETR(1, 6, (96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12, pos_embed="perceptron",
                                                     vit_norm_name=copy.deepcopy(COND), encoder_norm_name=copy.deepcopy(COND),
                                                     decoder_norm_name=copy.deepcopy(INST)), (1, 1, 96, 96, 96), [1], up)
    save("unetr_c3_truth", arrays, meta)


def gen_unet_truth():
    """C1 (the plain UNet, BASELINE configs[0]) and its conditional variant in float64 and under autocast-bf16"""
    import copy
    arrays, meta = {}, {"cases": {}}
    truth_case(arrays, meta, "c1_64", lambda: UNet(3, 1, 6, channels=[32, 64, 128, 256], strides=[2, 2, 2], num_res_units=2, act="prelu",
                                                   norm_down=copy.deepcopy(INST), norm_up=copy.deepcopy(INST), dropout=0.0, bias=True, adn_ordering="NDA"),
               (1, 1, 64, 64, 64), None, [])
    truth_case(arrays, meta, "cond_32", lambda: UNet(3, 1, 6, channels=[8, 16, 32], strides=[2, 2], num_res_units=2, act="prelu",
                                                     norm_down=copy.deepcopy(COND), norm_up=copy.deepcopy(INST), dropout=0.0, bias=True, adn_ordering="NDA"),
               (2, 1, 32, 32, 32), [1, 0], [])
    save("unet_truth", arrays, meta)


def gen_unetr():
    import copy
    arrays, meta = {}, {"cases": {}}
    m = UNETR(1, 6, (32, 32, 32), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4, pos_embed="perceptron",
              vit_norm_name=copy.deepcopy(COND), encoder_norm_name=copy.deepcopy(COND),
              decoder_norm_name=copy.deepcopy(INST))
    up = ["vit.blocks.*.attn (MONAI SABlock stand-in)", "vit.blocks.*.mlp (MONAI MLPBlock stand-in)"]
    whole_net("small_32", m, (2, 1, 32, 32, 32), [0, 1], {"feature_size": 8, "hidden_size": 48, "mlp_dim": 96,
                                                           "num_heads": 4, "pos_embed": "perceptron"},
              False, up, arrays, meta)
    save("unetr_small", arrays, meta)


def gen_unetr_c3():
    import copy
    arrays, meta = {}, {"cases": {}}
    m = UNETR(1, 6, (96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12, pos_embed="perceptron",
              vit_norm_name=copy.deepcopy(COND), encoder_norm_name=copy.deepcopy(COND),
              decoder_norm_name=copy.deepcopy(INST))
    up = ["vit.blocks.*.attn (MONAI SABlock stand-in)", "vit.blocks.*.mlp (MONAI MLPBlock stand-in)"]
    whole_net("c3_m1", m, (1, 1, 96, 96, 96), [1], {"feature_size": 16, "hidden_size": 768, "mlp_dim": 3072,
                                                     "num_heads": 12, "pos_embed": "perceptron"},
              False, up, arrays, meta)
    save("unetr_c3", arrays, meta)


def gen_unet():
    import copy
    arrays, meta = {}, {"cases": {}}
    m = UNet(3, 1, 6, channels=[32, 64, 128, 256], strides=[2, 2, 2], num_res_units=2, act="prelu",
             norm_down=copy.deepcopy(INST), norm_up=copy.deepcopy(INST), dropout=0.0, bias=True, adn_ordering="NDA")
    whole_net("c1_64", m, (1, 1, 64, 64, 64), None, {"channels": [32, 64, 128, 256], "strides": [2, 2, 2],
                                                      "num_res_units": 2}, False, [], arrays, meta)
    m = UNet(3, 1, 6, channels=[8, 16, 32], strides=[2, 2], num_res_units=2, act="prelu",
             norm_down=copy.deepcopy(COND), norm_up=copy.deepcopy(INST), dropout=0.0, bias=True, adn_ordering="NDA")
    whole_net("cond_32", m, (2, 1, 32, 32, 32), [1, 0], {"channels": [8, 16, 32], "strides": [2, 2],
                                                          "num_res_units": 2, "norm_down": "instance_cond"},
              False, [], arrays, meta)
    save("unet", arrays, meta)



SEEDS = (1234, 2345, 3456)
NSEED_SAMP = 1024


def gen_seeds_truth():
    """More draws of the whole-net conditioning comparison ("one draw against one draw is no parity check").
    For every whole-net case of the *_truth fixtures and every extra input seed s (the first draw, seed 1234, stays where it is): the
    REFERENCE's modules on x = det_input(s), cotangent det_input(s + 3087) - in float64 (the truth: 1024 strided samples of every parameter
    gradient are stored), in fp32 and under torch.autocast(cpu, bfloat16) (stored: only each parameter's DISTANCE from the float64 run over
    the same samples, which is all tests/test_hip_modules.py::test_*_vs_truth_over_seeds needs of them; one-element parameters keep their
    values, they are judged on the scale of their peers).  The gradients of these nets are a discontinuous function of the forward pass
    (activation sign flips): a bar on the median over seeds does not hinge on one lucky or unlucky flip of one draw."""
    import copy
    up_swin = ["swinViT.*.mlp (MONAI MLPBlock stand-in)"]
    up_vit = ["vit.blocks.*.attn (MONAI SABlock stand-in)", "vit.blocks.*.mlp (MONAI MLPBlock stand-in)"]
    cases = [
        ("c2_m0", lambda: SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=copy.deepcopy(COND),
                                     encoder_norm_name=copy.deepcopy(COND), decoder_norm_name=copy.deepcopy(INST)), (1, 1, 96, 96, 96), [0], up_swin),
        ("c3_m1", lambda: UNETR(1, 6, (96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12, pos_embed="perceptron",
                                vit_norm_name=copy.deepcopy(COND), encoder_norm_name=copy.deepcopy(COND), decoder_norm_name=copy.deepcopy(INST)),
         (1, 1, 96, 96, 96), [1], up_vit),
        ("small_32", lambda: UNETR(1, 6, (32, 32, 32), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4, pos_embed="perceptron",
                                   vit_norm_name=copy.deepcopy(COND), encoder_norm_name=copy.deepcopy(COND), decoder_norm_name=copy.deepcopy(INST)),
         (2, 1, 32, 32, 32), [0, 1], up_vit),
        ("c1_64", lambda: UNet(3, 1, 6, channels=[32, 64, 128, 256], strides=[2, 2, 2], num_res_units=2, act="prelu",
                               norm_down=copy.deepcopy(INST), norm_up=copy.deepcopy(INST), dropout=0.0, bias=True, adn_ordering="NDA"),
         (1, 1, 64, 64, 64), None, []),
        ("cond_32", lambda: UNet(3, 1, 6, channels=[8, 16, 32], strides=[2, 2], num_res_units=2, act="prelu",
                                 norm_down=copy.deepcopy(COND), norm_up=copy.deepcopy(INST), dropout=0.0, bias=True, adn_ordering="NDA"),
         (2, 1, 32, 32, 32), [1, 0], []),
    ]
    only = os.environ.get("MISEG_SEED_CASES")
    arrays, meta = {}, {"cases": {}, "seeds": list(SEEDS), "samples": NSEED_SAMP, "cotangent_seed_offset": 3087}

    def rel(a, b):
        a, b = torch.from_numpy(a).double(), torch.from_numpy(b).double()
        return float((a - b).norm() / (b.norm() + 1e-30))

    for tag, build, xshape, mods, unp in cases:
        if only and tag not in only.split(","):
            continue
        meta["cases"][tag] = {"x": list(xshape), "modalities": None if mods is None else [int(v) for v in mods], "unpinned": unp, "keys": None, "secs": {}}
        for seed in SEEDS:
            t0 = time.time()
            x = det_input(seed, xshape)
            md = build()
            fill_module_(md)
            md = md.double()
            yd = md(x.double(), mods)
            g = det_input(seed + 3087, tuple(yd.shape))
            yd.backward(g.double())
            truth = {k: sample(p.grad, NSEED_SAMP) for k, p in md.named_parameters() if p.grad is not None}
            l64 = sample(yd, NSEED_SAMP)
            del md, yd
            keys = sorted(truth)
            if meta["cases"][tag]["keys"] is None:
                meta["cases"][tag]["keys"] = keys
            assert meta["cases"][tag]["keys"] == keys
            pre = f"{tag}/s{seed}/"
            for k in keys:
                arrays[pre + "grad64:" + k] = truth[k]
            arrays[pre + "logits64_samples"] = l64
            for mode in ("fp32", "amp"):
                m = build()
                fill_module_(m)
                if mode == "amp":
                    with torch.autocast("cpu", dtype=torch.bfloat16):
                        y = m(x, mods)
                    y.backward(g.to(y.dtype))
                else:
                    y = m(x, mods)
                    y.backward(g)
                got = {k: sample(p.grad.float(), NSEED_SAMP) for k, p in m.named_parameters() if p.grad is not None}
                assert sorted(got) == keys
                arrays[pre + f"e_{mode}"] = np.asarray([rel(got[k], truth[k]) for k in keys], dtype=np.float64)
                arrays[pre + f"elogits_{mode}"] = np.float64(rel(sample(y.float(), NSEED_SAMP), l64))
                for k in keys:
                    if truth[k].size == 1:
                        arrays[pre + f"v_{mode}:" + k] = got[k]
                del m, y
            meta["cases"][tag]["secs"][str(seed)] = round(time.time() - t0, 1)
            print(f"    {tag} seed {seed}: {time.time() - t0:.0f}s, median distance fp32 {np.median(arrays[pre + 'e_fp32']):.2e} amp {np.median(arrays[pre + 'e_amp']):.2e}", flush=True)
    save("seeds_truth" if not only else "seeds_truth_" + only.replace(",", "_"), arrays, meta)


GENS = {
    "cond_instnorm": gen_cond_instnorm, "window_attention": gen_window_attention, "swin_block": gen_swin_block,
    "patch_merging": gen_patch_merging, "unetr_blocks": gen_unetr_blocks, "transformer_block": gen_transformer_block,
    "swin_unetr_small": gen_swin_unetr_small, "unetr_small": gen_unetr, "unet": gen_unet,
    "swin_unetr_c2": gen_swin_unetr_c2, "unetr_c3": gen_unetr_c3, "swin_unetr_c2_truth": gen_swin_unetr_c2_truth,
    "unetr_c3_truth": gen_unetr_c3_truth, "unet_truth": gen_unet_truth, "seeds_truth": gen_seeds_truth,
}
FULL = ("swin_unetr_c2", "unetr_c3", "swin_unetr_c2_truth", "unetr_c3_truth", "seeds_truth")

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--check-torch-bug", action="store_true")
    a = ap.parse_args()
    if a.check_torch_bug:
        check_torch_bug()
        sys.exit(0)
    for name, fn in GENS.items():
        if a.only and name not in a.only:
            continue
        if a.skip_full and name in FULL:
            continue
        print(name)
        fn()
