"""GPU parity tests, module level: the drop-in modules (HIP kernels through the C ABI) against the golden vectors
captured from the reference's own modules, and against the CPU oracle.  fp32 tolerance 1e-3 (north_star)."""
import copy
import os

import pytest
import torch

from conftest import compare_grads, evidence, rel_err, sample, scalar_scale, state_from_meta

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3          # north_star: within 1e-3 relative fp32
TOL_BF16 = 4e-2
ZERO_GRAD = ("encoder1.layer.conv3.conv.weight",)      # d/dw == 0 analytically (a per-channel scale in front of an instance norm): both sides hold rounding noise


def _fill(m):
    from mi_seg_amd.utils.detfill import fill_module_
    fill_module_(m)
    return m.to(DEV)


def _styles(mods, B):
    from mi_seg_amd.networks.norms.conditional_instance_norm import styles_to_device
    return styles_to_device(mods, DEV, B)


def _norm(name):
    from mi_seg_amd.networks.norms.utils import parse_normalization
    return parse_normalization(name, True, 4, 2)


def _check_param_grads(module, want, none_list, tol):
    named = dict(module.named_parameters())
    compare_grads({k: p.grad for k, p in named.items()}, want, tol)
    for k in none_list:
        assert named[k].grad is None, f"{k} must have no gradient"


def test_conditional_instance_norm_module(golden):
    from mi_seg_amd.networks.norms.conditional_instance_norm import ConditionalInstanceNorm1d, ConditionalInstanceNorm3d
    G = golden("cond_instnorm")
    for tag, cls in (("3d_mixed", ConditionalInstanceNorm3d), ("3d_same", ConditionalInstanceNorm3d),
                     ("3d_tensor_styles", ConditionalInstanceNorm3d), ("1d_mixed", ConditionalInstanceNorm1d)):
        case = G.meta["cases"][tag]
        m = _fill(cls(2, case["shape"][1]))
        x = G.t(f"{tag}/x").to(DEV).requires_grad_(True)
        styles = case["styles"] if tag != "3d_tensor_styles" else torch.tensor(case["styles"], device=DEV)
        y = m(x, styles)
        y.backward(G.t(f"{tag}/g").to(DEV))
        assert rel_err(y, G.t(f"{tag}/y")) < TOL
        assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
        _check_param_grads(m, G.grads(tag), case["grad_none"], TOL)
    m = _fill(ConditionalInstanceNorm3d(2, 3))
    with pytest.raises(ValueError):
        m(torch.zeros(2, 3, 4, 4, 4, device=DEV), [0])
    with pytest.raises(ValueError):
        m(torch.zeros(2, 3, 4, 4, device=DEV), [0, 1])


@pytest.mark.parametrize("tag", ["n343_nomask", "n343_mask", "n216_sliced"])
def test_window_attention(golden, tag):
    from mi_seg_amd.networks.blocks.window_attention import WindowAttention
    from oracle.functional import window_partition, window_reverse
    G = golden("window_attention")
    case = G.meta["cases"][tag]
    m = _fill(WindowAttention(case["dim"], case["heads"], (7, 7, 7), qkv_bias=True))
    xw, gw, yw, dxw = (G.t(f"{tag}/{k}") for k in ("x", "g", "y", "dx"))
    C = case["dim"]
    if tag == "n343_nomask":
        dims, ws, ss = (2, 7, 7, 7), (7, 7, 7), (0, 0, 0)
    elif tag == "n216_sliced":
        dims, ws, ss = (2, 6, 6, 6), (6, 6, 6), (0, 0, 0)
    else:
        dims, ws, ss = (2, 14, 7, 7), (7, 7, 7), (3, 3, 3)

    def to_grid(t):     # windows of the rolled grid -> un-rolled grid
        g = window_reverse(t, ws, dims)
        return torch.roll(g, shifts=ss, dims=(1, 2, 3)) if any(ss) else g

    def to_win(t):
        g = torch.roll(t, shifts=tuple(-s for s in ss), dims=(1, 2, 3)) if any(ss) else t
        return window_partition(g, ws)

    x = to_grid(xw).contiguous().to(DEV).requires_grad_(True)
    y = m(x, ws, ss)
    y.backward(to_grid(gw).contiguous().to(DEV))
    assert rel_err(to_win(y.detach().cpu()), yw) < TOL
    assert rel_err(to_win(x.grad.cpu()), dxw) < TOL
    _check_param_grads(m, G.grads(tag), [], TOL)


@pytest.mark.parametrize("tag", ["pad_noshift", "pad_shift", "clamped6", "layer_shift", "inst_noshift"])
def test_swin_block(golden, tag):
    from mi_seg_amd.networks.blocks.swin_transformer_block import SwinTransformerBlock
    G = golden("swin_block")
    case = G.meta["cases"][tag]
    m = _fill(SwinTransformerBlock(case["dim"], case["heads"], (7, 7, 7), tuple(case["shift"]), norm_type=_norm(case["norm"])))
    x = G.t(f"{tag}/x").to(DEV).requires_grad_(True)
    y = m(x, _styles(case["modalities"], x.shape[0]))
    y.backward(G.t(f"{tag}/g").to(DEV))
    assert rel_err(y, G.t(f"{tag}/y")) < TOL
    assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
    _check_param_grads(m, G.grads(tag), case["grad_none"], TOL)


@pytest.mark.parametrize("dim,heads,grid,shift,norm", [(48, 3, (16, 16, 16), (0, 0, 0), "instance_cond"), (96, 6, (16, 18, 16), (3, 3, 3), "instance_cond"),
                                                  (48, 3, (16, 16, 17), (3, 3, 3), "instance"), (192, 12, (12, 12, 12), (3, 3, 3), "instance_cond"),
                                                  (384, 24, (6, 6, 6), (0, 0, 0), "instance_cond")])
def test_swin_block_with_its_norms_folded_into_the_gemms(dim, heads, grid, shift, norm):
    """round 5 (stages 1 - 2 of the headline net: one sample, bf16, >= 4096 tokens): norm1's apply pass inside the qkv GEMM's operand load,
    norm2's inside the MLP's first product, the norms' backward sums in the data-gradient epilogues (HF._NormLinear / _NormMlp) - against the
    same block with every norm pass a launch of its own (ops.FOLD_NORMS = False): same output bit for bit (same fma, same rounding), gradients
    to the summation order of the backward sums; and the launch count drops by the folded passes."""
    from mi_seg_amd.hip import ops
    from mi_seg_amd.networks.blocks.swin_transformer_block import SwinTransformerBlock
    from mi_seg_amd.testing import roofline
    m = _fill(SwinTransformerBlock(dim, heads, (7, 7, 7), shift, norm_type=_norm(norm)))
    m.to(DEV)
    st = _styles([1], 1)
    g = torch.Generator().manual_seed(7)
    x0 = (torch.randn((1,) + grid + (dim,), generator=g) * 1.5 + 0.3).to(DEV).to(torch.bfloat16)
    cot = torch.randn((1,) + grid + (dim,), generator=g).to(DEV).to(torch.bfloat16)
    res = {}
    for fold in (True, False):
        ops.FOLD_NORMS = fold
        try:
            m.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            y = m(x, st)
            y.backward(cot)
            res[fold] = (y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                         sorted(k for k, p in m.named_parameters() if p.grad is None))
        finally:
            ops.FOLD_NORMS = True
    (y1, dx1, g1, n1), (y0, dx0, g0, n0) = res[True], res[False]
    if x0.numel() // dim >= 4096:
        assert torch.equal(y1, y0), rel_err(y1, y0)
    else:      # the deep stages: the unfolded form is the register-resident one-launch norm kernel, whose statistics are summed in another order
        assert rel_err(y1, y0) < 4e-3, rel_err(y1, y0)
    assert n1 == n0 and sorted(g1) == sorted(g0)
    assert rel_err(dx1, dx0) < 4e-3, rel_err(dx1, dx0)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 4e-3, (k, rel_err(g1[k], g0[k]))
    if norm == "instance_cond":
        assert any(k.endswith("norms.0.weight") for k in n1)          # the absent style's rows keep grad None in both forms


@pytest.mark.parametrize("tag", ["merging", "mergingv2", "merging_odd_layer"])
def test_patch_merging(golden, tag):
    from mi_seg_amd.networks.blocks.patch_merging import PatchMerging, PatchMergingV2
    G = golden("patch_merging")
    case = G.meta["cases"][tag]
    cls = PatchMergingV2 if tag == "mergingv2" else PatchMerging
    m = _fill(cls(4, norm_type=_norm(case["norm"]), spatial_dims=3))
    x = G.t(f"{tag}/x").to(DEV).requires_grad_(True)
    y = m(x, _styles(case["modalities"], 2))
    y.backward(G.t(f"{tag}/g").to(DEV))
    assert rel_err(y, G.t(f"{tag}/y")) < TOL
    assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
    _check_param_grads(m, G.grads(tag), case["grad_none"], TOL)


def _cl(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


@pytest.mark.parametrize("tag", ["res_8_to_12_cond", "res_8_to_8_cond", "res_1_to_8_inst", "up_16_to_8_inst", "prup_16_to_8_cond"])
def test_unetr_blocks(golden, tag):
    from mi_seg_amd.networks.blocks.dynunet_block import UnetResBlock
    from mi_seg_amd.networks.blocks.unetr_block import UnetrPrUpBlock, UnetrUpBlock
    G = golden("unetr_blocks")
    case = G.meta["cases"][tag]
    cond, inst = _norm("instance_cond"), _norm("instance")
    st = _styles(case["modalities"], 2)
    x_nc = G.t(f"{tag}/x").to(DEV)
    if tag == "res_1_to_8_inst":
        m = _fill(UnetResBlock(3, 1, 8, 3, 1, inst))
        y = m(None, st, image=x_nc, dtype=torch.float32)
        x = None
    else:
        x = _cl(x_nc).requires_grad_(True)
        if tag == "res_8_to_12_cond":
            m = _fill(UnetResBlock(3, 8, 12, 3, 1, cond))
            y = m(x, st)
        elif tag == "res_8_to_8_cond":
            m = _fill(UnetResBlock(3, 8, 8, 3, 1, cond))
            y = m(x, st)
        elif tag == "up_16_to_8_inst":
            m = _fill(UnetrUpBlock(3, 16, 8, 3, 2, inst, res_block=True))
            skip = _cl(G.t(f"{tag}/skip").to(DEV)).requires_grad_(True)
            y = m(x, skip, st)
        else:
            m = _fill(UnetrPrUpBlock(3, 16, 8, 1, 3, 1, 2, cond, conv_block=True, res_block=True))
            y = m(x, st)
    y.backward(_cl(G.t(f"{tag}/g").to(DEV)))
    assert rel_err(y.permute(0, 4, 1, 2, 3), G.t(f"{tag}/y")) < TOL
    if x is not None:
        assert rel_err(x.grad.permute(0, 4, 1, 2, 3), G.t(f"{tag}/dx")) < TOL
    if tag == "up_16_to_8_inst":
        assert rel_err(skip.grad.permute(0, 4, 1, 2, 3), G.t(f"{tag}/dskip")) < TOL
    _check_param_grads(m, G.grads(tag), case["grad_none"], TOL)


def _whole(G, tag, model, tol, dtype=torch.float32, ce=False):
    from mi_seg_amd.utils.detfill import ce_cotangent, det_input
    case = G.meta["cases"][tag]
    model = _fill(model)
    assert list(model.state_dict().keys()) == case["state_keys"]
    model.set_compute_dtype(dtype)
    x = det_input(1234, case["x"]).to(DEV)
    y = model(x, case["modalities"])
    assert y.dtype == torch.float32 and list(y.shape[:2]) == [case["x"][0], 6]
    e = rel_err(sample(y), G.t(f"{tag}/logits_samples"))
    assert e < tol, ("logits", e)
    y.backward(ce_cotangent(y) if ce else det_input(4321, tuple(y.shape)).to(DEV))
    named = dict(model.named_parameters())
    # 10 x tol: the reference's own fp32 run sits 1e-3 .. 5e-3 from its float64 run on these gradients (test_swin_unetr_c2_vs_truth).
    # bf16: the gradients are judged per parameter against the float64 fixtures (test_*_vs_truth) - the fp32 fixture is no bar for them
    worst = None
    if dtype == torch.float32:
        worst = compare_grads({k: p.grad for k, p in named.items()}, G.grads2(tag) if ce else G.grads(tag), 10 * tol, sampled=True, vanish_tol=1e-2)
    none = [k for k, p in named.items() if p.grad is None]
    assert sorted(none) == sorted(case["grad_none"])
    return worst


@pytest.mark.parametrize("tag", ["fs12_64_m10", "fs12_64_v2_layer"])
def test_swin_unetr_small(golden, tag):
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    G = golden("swin_unetr_small")
    c = G.meta["cases"][tag]
    m = SwinUNETR((64, 64, 64), 1, 6, feature_size=12, num_heads=(3, 6, 12, 24), downsample=c["downsample"], vit_norm_name=_norm(c["vit_norm"]),
                  encoder_norm_name=_norm(c["encoder_norm"]), decoder_norm_name=_norm(c["decoder_norm"]))
    _whole(G, tag, m, TOL)


@pytest.mark.parametrize("tag,dtype,tol,ce", [("c2_m0", torch.float32, TOL, False), ("c2_m1", torch.float32, TOL, False),
                                              ("c2_m0", torch.float32, TOL, True)])
def test_swin_unetr_c2_headline(golden, tag, dtype, tol, ce):
    """BASELINE configs[1]: C-Swin-UNETR fs=48, 96^3, 6 classes -- fwd + bwd on the same seeded patch as the reference.
    ce=False: white-noise cotangent on the logits (every parameter gradient is then a sqrt(N)-cancelling random sum: the
    fp32 path still agrees to ~3e-3); ce=True: gradient of the mean voxel cross-entropy against block labels.  Logits at 1e-3, every
    parameter gradient at 1e-2 against the reference's fp32 run with no escape for small errors; the bf16 mode and the sharper per-parameter
    bar against the reference's float64 run are in test_swin_unetr_c2_vs_truth."""
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    G = golden("swin_unetr_c2")
    m = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"),
                  encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance"))
    _whole(G, tag, m, tol, dtype, ce)


@pytest.mark.parametrize("dtype,cot", [(torch.float32, "noise"), (torch.float32, "ce"), (torch.bfloat16, "noise"), (torch.bfloat16, "ce")])
def test_swin_unetr_c2_vs_truth(golden, dtype, cot):
    """The gradient bar of the benchmarked configuration, per parameter, no pooling, no escape hatch.

    The parameter gradients of this net are a DISCONTINUOUS function of the forward pass (LeakyReLU(0.01) on instance-normalised
    pre-activations: a forward deviation eps flips ~0.4 eps of the signs), so they move by ~sqrt(eps): the reference's OWN fp32 run is
    1.2e-3 (white-noise cotangent) / 3.4e-3 (cross-entropy cotangent) median away from its float64 run, its own autocast-bf16 run 24 %
    (tests/golden/swin_unetr_c2_truth.npz, made by oracle/tools/make_golden.py from the reference's modules).  "Within 1e-3 of the fp32
    reference" is therefore below the reference's own rounding noise at this test point; what CAN be asked of an implementation is that it
    is no further from the float64 run than the reference itself is at the same precision:
        fp32 mode:  |hip - f64| <= 3.5 x |ref_fp32 - f64| + 1e-5 per parameter,  median over parameters <= 1.5 x the reference's median,
                    logits within 8e-7 of the float64 run (the reference's fp32 run: 7.6e-7)
        bf16 mode:  |hip - f64| <= 2 x |ref_autocast - f64| per parameter,      median over parameters <= 1.25 x the reference's median
    relative L2 over the 4096-element sample of each tensor.  Both sides of each inequality are single draws of rounding noise (any change
    of a summation order re-rolls them).  Round 3: the parity mode's 3x3x3 convolution sums in blocks (csrc/conv3d.hip) - logits 1.3e-6 ->
    4.6e-7 from the float64 run; two builds of equal kernel accuracy (scripts/debug/norm_bwd_accuracy.py: the norm backward kernels of both sit
    1e-7 from float64) drew gradient medians 0.79 x / 1.29 x the reference's (white noise) and 1.01 x (cross-entropy), worst per-parameter
    ratios 1.2 / 3.3 - which sign flips a run catches is a draw, hence 3.5 x per parameter (round 4; 4 x in round 3; rounds 1-2, one running fp32 sum over K = 27 Cin:
    medians 1.65 x, worst 2.7 - 3.9); bf16: worst 1.2 - 1.5, medians 0.90 - 0.93.  Parameters whose true gradient is zero (a bias in front of an
    instance norm) are listed by the float64 run itself and must be ~0."""
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.utils.detfill import ce_cotangent, det_input
    T, R = golden("swin_unetr_c2_truth"), golden("swin_unetr_c2")
    m = _fill(SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"),
                        encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")))
    m.set_compute_dtype(dtype)
    y = m(det_input(1234, (1, 1, 96, 96, 96)).to(DEV), [0])
    k64, kref = ("grad64:", "grad:" if dtype == torch.float32 else "gradamp:") if cot == "noise" else ("grad64_2:", "grad2:" if dtype == torch.float32 else "gradamp_2:")
    src = R if dtype == torch.float32 else T
    e_logits = rel_err(sample(y), T.t("c2_m0/logits64_samples"))
    e_ref = rel_err(R.t("c2_m0/logits_samples") if dtype == torch.float32 else T.t("c2_m0/logitsamp_samples"), T.t("c2_m0/logits64_samples"))
    assert e_logits <= (8e-7 if dtype == torch.float32 else 3.0 * e_ref), ("logits", e_logits, e_ref)
    y.backward(ce_cotangent(y) if cot == "ce" else det_input(4321, tuple(y.shape)).to(DEV))
    truth = {k[len("c2_m0/" + k64):]: T.t(k) for k in T.z.files if k.startswith("c2_m0/" + k64)}
    rms = {k: float(g.double().norm()) / g.numel() ** 0.5 for k, g in truth.items()}
    med = sorted(rms.values())[len(rms) // 2]
    named = dict(m.named_parameters())
    factor, slack, med_factor = (3.5, 1e-5, 1.5) if dtype == torch.float32 else (2.0, 0.0, 1.25)
    worst = (0.0, "")
    all_hip, all_ref = [], []
    for k, t in truth.items():
        got = sample(named[k].grad)
        if rms[k] < 1e-3 * med:                                   # analytically zero gradient
            assert float(got.double().norm()) / got.numel() ** 0.5 < (1e-4 if dtype == torch.float32 else 0.1) * med, (k, "should vanish")
            continue
        e_hip, e_r = rel_err(got, t), rel_err(src.t(f"c2_m0/{kref}{k}"), t)
        worst = max(worst, (e_hip / (e_r + 1e-12), k))
        all_hip.append(e_hip)
        all_ref.append(e_r)
        assert e_hip <= factor * e_r + slack, (k, e_hip, e_r)
    m_hip, m_ref = sorted(all_hip)[len(all_hip) // 2], sorted(all_ref)[len(all_ref) // 2]
    assert m_hip <= med_factor * m_ref, ("median over parameters", m_hip, m_ref)
    evidence(f"c2 vs truth {dtype} {cot}: logits {e_logits:.2e} (reference at this precision {e_ref:.2e}); gradient medians {m_hip:.2e} vs {m_ref:.2e}; "
             f"worst ratio {worst[0]:.2f} ({worst[1]}); bar: per parameter {factor} x + {slack}, median {med_factor} x")
    assert sorted(k for k, p in named.items() if p.grad is None) == sorted(R.meta["cases"]["c2_m0"]["grad_none"])


def test_swin_unetr_c2_full_tensors_vs_oracle():
    """the fixtures hold 4096 strided samples per tensor; a localized error (one brick, one window) could hide between them.  Here the CPU
    oracle runs the C2 patch once (fp32, ~6 s on 16 threads) and EVERY element of the logits and of every parameter gradient is compared:
    logits relative L2 < 1e-5 and max |diff| < 1e-4 of the largest logit; gradients relative L2 < 1e-2 over the full tensors."""
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.utils.detfill import det_input
    from oracle import nets as ON
    m = _fill(SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"),
                        encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")))
    sd = {k: (v.detach().cpu().clone().requires_grad_(True) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in m.state_dict().items()}
    x, g = det_input(1234, (1, 1, 96, 96, 96)), det_input(4321, (1, 6, 96, 96, 96))
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16))
    yo = ON.swin_unetr_forward(sd, x, [1], ON.swin_unetr_cfg(feature_size=48))
    yo.backward(g)
    torch.set_num_threads(threads)
    y = m(x.to(DEV), [1])
    y.backward(g.to(DEV))
    d = (y.detach().cpu() - yo.detach())
    assert rel_err(y, yo) < 1e-5
    assert float(d.abs().max()) < 1e-4 * float(yo.detach().abs().max())
    want = {k: v.grad for k, v in sd.items() if v.is_floating_point() and v.grad is not None}
    got = {k: p.grad for k, p in m.named_parameters()}
    assert sorted(k for k, p in m.named_parameters() if p.grad is None) == sorted(k for k, v in sd.items() if v.is_floating_point() and v.grad is None)
    compare_grads(got, want, 1e-2, skip=ZERO_GRAD)


def test_forward_and_data_gradients_are_bitwise_reproducible():
    """two identical eager steps: logits, the gradient w.r.t. every activation on the dX chain and hence every dX-derived quantity are
    bit-identical (no atomics on that path: statistics are fp64 sums whose rounding to fp32 is order-independent in practice, split
    reductions are summed in a fixed order).  The weight-gradient reductions (split-K partial tiles / fp32 atomics of small outputs) are
    reproducible to rounding only: checked at 1e-5."""
    from mi_seg_amd.hip import ops
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.utils.detfill import det_input
    m = _fill(SwinUNETR((64, 64, 64), 1, 6, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"),
                        encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")))
    x = det_input(1, (2, 1, 64, 64, 64)).to(DEV)
    cot = det_input(2, (2, 6, 64, 64, 64)).to(DEV)
    runs = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        ops.begin_step()
        y = m(x, [1, 0])
        y.backward(cot)
        runs.append((y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0]), "logits differ between two identical runs"
    # (analytically zero gradients hold rounding noise: judged absolutely; the stem's 1x1x1 shortcut weight - one input channel in front of
    # an instance norm, d/dw == 0 - is noise of ordinary magnitude and is left out)
    compare_grads(runs[1][1], {k: v.float().cpu() for k, v in runs[0][1].items()}, 1e-5, skip=ZERO_GRAD)


@pytest.mark.parametrize("tag,dtype", [("c1_64", torch.float32), ("cond_32", torch.float32), ("c1_64", torch.bfloat16)])
def test_unet(golden, tag, dtype):
    """BASELINE configs[0] on the HIP path: MONAI residual UNet (strided 3^3 convs, ConvTranspose k3 s2, PReLU, biases) fwd + bwd
    against the fixtures of the reference's own modules."""
    from mi_seg_amd.networks.nets.unet import UNet
    G = golden("unet")
    c = G.meta["cases"][tag]
    m = UNet(3, 1, 6, channels=c["channels"], strides=c["strides"], num_res_units=c["num_res_units"], act="prelu",
             norm_down=_norm(c.get("norm_down", "instance")), norm_up=_norm("instance"), dropout=0.0, bias=True, adn_ordering="NDA")
    _whole(G, tag, m, TOL if dtype == torch.float32 else TOL_BF16, dtype)


@pytest.mark.parametrize("tag", ["cond", "layer_bias"])
def test_transformer_block(golden, tag):
    from mi_seg_amd.networks.blocks.transformer_block import TransformerBlock
    G = golden("transformer_block")
    case = G.meta["cases"][tag]
    m = _fill(TransformerBlock(case["hidden"], case["mlp"], case["heads"], 0.0, case["qkv_bias"], norm_type=_norm(case["norm"])))
    x = G.t(f"{tag}/x").to(DEV).requires_grad_(True)
    y = m(x, _styles(case["modalities"], 2), (3, 3, 3))
    y.backward(G.t(f"{tag}/g").to(DEV))
    assert rel_err(y, G.t(f"{tag}/y")) < TOL
    assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
    _check_param_grads(m, G.grads(tag), case["grad_none"], TOL)


def test_unetr_small(golden):
    from mi_seg_amd.networks.nets.unetr import UNETR
    G = golden("unetr_small")
    m = UNETR(1, 6, (32, 32, 32), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4, pos_embed="perceptron", vit_norm_name=_norm("instance_cond"),
              encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance"))
    _whole(G, "small_32", m, TOL)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL), (torch.bfloat16, TOL_BF16)])
def test_unetr_c3(golden, dtype, tol):
    """BASELINE configs[2]: C-UNETR (ViT-B/16 encoder, instance_cond), 96^3."""
    from mi_seg_amd.networks.nets.unetr import UNETR
    G = golden("unetr_c3")
    m = UNETR(1, 6, (96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12, pos_embed="perceptron",
              vit_norm_name=_norm("instance_cond"), encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance"))
    _whole(G, "c3_m1", m, tol, dtype)


SMALL_NET_BAR = dict(fp32=(3.0, 1e-3, 1.5), bf16=(4.0, 2e-2, 1.5), bf16_scalar_family=True)
"""(factor, slack, median factor) of _vs_truth for the 32^3 / 64^3 nets.  Their gradients jump with every single activation-sign flip
(LeakyReLU / PReLU on normalised pre-activations): with ~1e6 pre-activations and a forward error of 3e-7 the expected number of flips is
below one, so the reference's fp32 run happens to sit 8e-7 from its float64 run while ONE flip near the output puts every upstream
gradient 1e-4 .. 7e-3 away (measured: UNETR 32^3, 205 of 276 parameters at 3.6e-3 .. 6.6e-3 together; UNet 64^3, the six parameters in front of
the first PReLU at 1.1e-4) - the fp32 slack was that jump (1e-2) through round 4.  Round 5: 1e-3 - the fp32 convolutions keep ONE summation
plan and the fp32 GEMMs sum in blocks, neither small net catches a flip on this input any more (gradient medians 5.4e-7 / 1.1e-6); the
robust form of this check is test_vs_truth_over_seeds (three inputs, medians over seeds, no slack at all).  bf16: torch.autocast keeps norm outputs / activations in fp32 and only runs convolutions
and linears in bf16, this path STORES every activation in bf16 - sums with heavy cancellation (the one-element PReLU slope gradients: a
sum over ~2 M voxels) carry that storage rounding, and so do autocast's: against the float64 run the 13 slopes of the C1 UNet (|g| 1.5 .. 1166,
median 174) sit 2 .. 86 away under autocast (rms 36 = 0.21 of the median) and 3 .. 103 away here (rms 33 / 41 / 41 with the 96-byte-chunk
convolution path padded from 0 / 64 / 128 bytes: every change of a summation order in front of them re-draws all 13, the largest error
moves from slope to slope - scripts/debug/unet_scalar_bf16.py).  One draw against one draw is no bar for them: `bf16_scalar_family` compares
the FAMILY (rms of error / scale no more than 1.5 x autocast's + 0.05) and holds every single slope within 3 x autocast's rms + 0.1.
The full-size nets (C2, C3) need none of this.  Round 4: the bf16 bar is relative to autocast's own distance like C2's - 4 x per parameter
with an absolute slack of 2e-2 (was 3 x + 1e-1: a 10 % slack is no parity check); measured worst 3.79 x on one 1x1x1 weight of the 32^3 UNETR,
every other parameter of the three small nets within 3 x (profiles/r04_vs_truth.txt holds the printed lines of the final build)."""


def _vs_truth(T, R, tag, model, dtype, fp32=(4.0, 1e-5, 2.0), bf16=(2.0, 0.0, 1.25), bf16_scalar_family=False):
    """the bar of test_swin_unetr_c2_vs_truth for any net: per parameter no further from the reference's float64 run (fixture T) than the
    reference's own run at the same precision (fp32: fixture R, autocast-bf16: T) times `factor` (+ slack), medians within `med_factor` (+ slack)"""
    from mi_seg_amd.utils.detfill import det_input
    case = T.meta["cases"][tag]
    m = _fill(model)
    m.set_compute_dtype(dtype)
    y = m(det_input(1234, case["x"]).to(DEV), case["modalities"])
    is32 = dtype == torch.float32
    e_logits = rel_err(sample(y), T.t(f"{tag}/logits64_samples"))
    e_ref = rel_err(R.t(f"{tag}/logits_samples") if is32 else T.t(f"{tag}/logitsamp_samples"), T.t(f"{tag}/logits64_samples"))
    assert e_logits <= 2.0 * e_ref + (1e-6 if is32 else 0.0), ("logits", e_logits, e_ref)
    y.backward(det_input(4321, tuple(y.shape)).to(DEV))
    k64 = f"{tag}/grad64:"
    truth = {k[len(k64):]: T.t(k) for k in T.z.files if k.startswith(k64)}
    rms = {k: float(g.double().norm()) / g.numel() ** 0.5 for k, g in truth.items()}
    med = sorted(rms.values())[len(rms) // 2]
    named = dict(m.named_parameters())
    factor, slack, med_factor = fp32 if is32 else bf16
    all_hip, all_ref, worst, bad, fam = [], [], (0.0, ""), [], []
    need = 0.0          # the absolute slack the worst parameter needs on top of `factor` x the reference's distance (reported as evidence)
    scal = scalar_scale(truth)
    for k, t in truth.items():
        got = sample(named[k].grad)
        if rms[k] < 1e-3 * med:                                   # analytically zero gradient
            assert float(got.double().norm()) / got.numel() ** 0.5 < (1e-4 if is32 else 0.1) * med, (k, "should vanish")
            continue
        e_hip = rel_err(got, t)
        e_r = rel_err(R.t(f"{tag}/grad:{k}") if is32 else T.t(f"{tag}/gradamp:{k}"), t)
        if t.numel() == 1:      # an ill-conditioned one-element sum: judged on the scale of its peers (conftest.scalar_scale)
            scale = max(float(t.double().abs().max()), scal)
            ref_v = R.t(f"{tag}/grad:{k}") if is32 else T.t(f"{tag}/gradamp:{k}")
            e_hip, e_r = float((got.double() - t.double()).abs().max()) / scale, float((ref_v.double() - t.double()).abs().max()) / scale
        worst = max(worst, (e_hip / (e_r + 1e-12), k))
        all_hip.append(e_hip)
        all_ref.append(e_r)
        if t.numel() == 1 and not is32 and bf16_scalar_family:      # judged as a family below (SMALL_NET_BAR)
            fam.append((e_hip, e_r, k))
            continue
        need = max(need, e_hip - factor * e_r)
        if e_hip > factor * e_r + slack:
            bad.append((round(e_hip / (e_r + 1e-12), 2), k, f"{e_hip:.2e}", f"{e_r:.2e}"))
    if fam:
        rms_hip, rms_ref = (sum(f[0] ** 2 for f in fam) / len(fam)) ** 0.5, (sum(f[1] ** 2 for f in fam) / len(fam)) ** 0.5
        assert rms_hip <= 1.5 * rms_ref + 0.05, ("one-element parameters as a family", rms_hip, rms_ref)
        bad += [(round(f[0] / (rms_ref + 1e-12), 2), f[2], f"{f[0]:.2e}", f"family rms {rms_ref:.2e}") for f in fam if f[0] > 3.0 * rms_ref + 0.1]
    assert not bad, (f"{len(bad)} of {len(all_hip)} parameters further from the float64 run than {factor} x the reference at this precision", sorted(bad)[-12:])
    m_hip, m_ref = sorted(all_hip)[len(all_hip) // 2], sorted(all_ref)[len(all_ref) // 2]
    assert m_hip <= med_factor * m_ref + slack, ("median over parameters", m_hip, m_ref)
    evidence(f"{tag} vs truth {dtype}: logits {e_logits:.2e} (reference at this precision {e_ref:.2e}); gradient medians {m_hip:.2e} vs {m_ref:.2e}; "
             f"worst ratio {worst[0]:.2f} ({worst[1]}); slack needed at {factor} x: {need:.2e}; bar: per parameter {factor} x + {slack}, median {med_factor} x + {slack}")
    assert sorted(k for k, p in named.items() if p.grad is None) == sorted(R.meta["cases"][tag]["grad_none"])


@pytest.mark.parametrize("tag", ["small_32", "c3_m1"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unetr_vs_truth(golden, tag, dtype):
    """BASELINE configs[2] (C-UNETR, networks/nets/unetr.py:254-276) and the small UNETR against the reference's float64 run: the per-parameter
    bar of test_swin_unetr_c2_vs_truth, no pooling of small parameters, no escape for small errors"""
    from mi_seg_amd.networks.nets.unetr import UNETR
    T, R = golden("unetr_c3_truth"), golden("unetr_small" if tag == "small_32" else "unetr_c3")
    kw = dict(feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4) if tag == "small_32" else dict(feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12)
    m = UNETR(1, 6, tuple(T.meta["cases"][tag]["x"][2:]), pos_embed="perceptron", vit_norm_name=_norm("instance_cond"), encoder_norm_name=_norm("instance_cond"),
              decoder_norm_name=_norm("instance"), **kw)
    _vs_truth(T, R, tag, m, dtype, **(SMALL_NET_BAR if tag == "small_32" else {}))


@pytest.mark.parametrize("tag", ["c1_64", "cond_32"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unet_vs_truth(golden, tag, dtype):
    """BASELINE configs[0] (the plain UNet, networks/nets/unet.py:351-353) and its conditional variant against the reference's float64 run"""
    from mi_seg_amd.networks.nets.unet import UNet
    T, R = golden("unet_truth"), golden("unet")
    c = R.meta["cases"][tag]
    m = UNet(3, 1, 6, channels=c["channels"], strides=c["strides"], num_res_units=c["num_res_units"], act="prelu",
             norm_down=_norm(c.get("norm_down", "instance")), norm_up=_norm("instance"), dropout=0.0, bias=True, adn_ordering="NDA")
    _vs_truth(T, R, tag, m, dtype, **SMALL_NET_BAR)


def _seed_case_model(tag):
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.networks.nets.unet import UNet
    from mi_seg_amd.networks.nets.unetr import UNETR
    cond, inst = (lambda: _norm("instance_cond")), (lambda: _norm("instance"))
    if tag == "c2_m0":
        return SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=cond(), encoder_norm_name=cond(), decoder_norm_name=inst())
    if tag == "c3_m1":
        return UNETR(1, 6, (96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12, pos_embed="perceptron", vit_norm_name=cond(),
                     encoder_norm_name=cond(), decoder_norm_name=inst())
    if tag == "small_32":
        return UNETR(1, 6, (32, 32, 32), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4, pos_embed="perceptron", vit_norm_name=cond(),
                     encoder_norm_name=cond(), decoder_norm_name=inst())
    if tag == "c1_64":
        return UNet(3, 1, 6, channels=[32, 64, 128, 256], strides=[2, 2, 2], num_res_units=2, act="prelu", norm_down=inst(), norm_up=inst(), dropout=0.0,
                    bias=True, adn_ordering="NDA")
    return UNet(3, 1, 6, channels=[8, 16, 32], strides=[2, 2], num_res_units=2, act="prelu", norm_down=cond(), norm_up=inst(), dropout=0.0, bias=True,
                adn_ordering="NDA")


SEED_BARS = {torch.float32: dict(per_param=2.5, family=1.5, logits=1.25), torch.bfloat16: dict(per_param=1.6, family=1.25, logits=1.25)}
"""test_vs_truth_over_seeds, no absolute slack anywhere (the 1e-9 below only keeps 0 / 0 apart):
  * per parameter: the MEDIAN OVER THE THREE INPUT SEEDS of (distance of the HIP gradient from the reference's float64 run) / (distance of the
    reference's own run at the same precision from it) <= `per_param`;
  * per net: the median over seeds of (median over parameters of the HIP distances / median of the reference's) <= `family`;
  * the logits of EVERY seed within `logits` x the reference's distance;
  * one-element parameters in bf16 (PReLU slopes: sums over ~2 M voxels with heavy cancellation, SMALL_NET_BAR) as a family: the median over
    seeds of rms(HIP errors) / rms(reference errors) <= 1.5.
One draw against one draw (the single-seed tests above) hinges on which activation-sign flips the two runs happen to catch: the reference's
own fp32 run of the 32^3 UNETR sits 1.7e-6, 6.7e-3 and 1.2e-4 from its float64 run on the three seeds, the plain UNet's 1.2e-6, 1.2e-6,
2.5e-4 (oracle/tools/make_golden.py::gen_seeds_truth prints them); a median over seeds does not.
Measured on the round-5 build (profiles/r05_vs_truth_seeds.txt): fp32 worst per-parameter ratio 0.85 (C2) / 1.07 (C3) / 1.12 / 1.66 / 2.13 (the last
two: one-element PReLU slopes of the small UNets), family ratios 0.41 ... 1.06; bf16 worst 1.05 / 1.23 / 1.46, families 0.91 ... 1.07.
This test found a real gap on its first run: C-UNETR's fp32 logits sat 1.5 - 1.9 x further from float64 than the reference's on every seed
(one running fp32 sum over K = 4096 in the patch-embedding GEMM); the blocked accumulation of csrc/gemm.hip put them at 0.5 x."""


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag", ["c2_m0", "c3_m1", "small_32", "c1_64", "cond_32"])
def test_vs_truth_over_seeds(golden, tag, dtype):
    """every whole-net configuration of BASELINE.json on THREE inputs (tests/golden/seeds_truth.npz: the reference's
    float64 gradients and the distances of its own fp32 / autocast-bf16 runs from them, 1024 samples per parameter)."""
    from mi_seg_amd.utils.detfill import det_input
    S = golden("seeds_truth")
    case, seeds, n = S.meta["cases"][tag], S.meta["seeds"], S.meta["samples"]
    keys = case["keys"]
    bars = SEED_BARS[dtype]
    is32 = dtype == torch.float32
    mode = "fp32" if is32 else "amp"
    m = _fill(_seed_case_model(tag))
    m.set_compute_dtype(dtype)
    named = dict(m.named_parameters())
    ratios = {k: [] for k in keys}
    lines, fam_ratio, scal_ratio = [], [], []
    measure = bool(os.environ.get("MISEG_SEEDS_MEASURE"))      # print every line, assert nothing: how the bars were read off (profiles/r05_vs_truth_seeds.txt)
    for seed in seeds:
        pre = f"{tag}/s{seed}/"
        m.zero_grad(set_to_none=True)
        y = m(det_input(seed, tuple(case["x"])).to(DEV), case["modalities"])
        e_log, e_log_ref = rel_err(sample(y, n), S.t(pre + "logits64_samples")), float(S.z[pre + f"elogits_{mode}"])
        assert measure or e_log <= bars["logits"] * e_log_ref, (tag, seed, "logits", e_log, e_log_ref)
        y.backward(det_input(seed + S.meta["cotangent_seed_offset"], tuple(y.shape)).to(DEV))
        truth = {k: S.t(pre + "grad64:" + k) for k in keys}
        e_ref = dict(zip(keys, [float(v) for v in S.z[pre + f"e_{mode}"]]))
        rms = {k: float(g.double().norm()) / g.numel() ** 0.5 for k, g in truth.items()}
        med = sorted(rms.values())[len(rms) // 2]
        scal = scalar_scale(truth)
        hip, ref, sh, sr = [], [], [], []
        for k in keys:
            got, t = sample(named[k].grad, n), truth[k]
            if rms[k] < 1e-3 * med:                                   # analytically zero gradient: must vanish here too
                assert float(got.double().norm()) / got.numel() ** 0.5 < (1e-4 if is32 else 0.1) * med, (k, seed, "should vanish")
                continue
            if t.numel() == 1:      # an ill-conditioned one-element sum (PReLU slopes): judged on the scale of its peers (conftest.scalar_scale)
                scale = max(float(t.double().abs().max()), scal)
                e_h = float((got.double() - t.double()).abs().max()) / scale
                e_r = float((S.t(pre + f"v_{mode}:" + k).double() - t.double()).abs().max()) / scale
                if not is32:        # bf16: as a family (see SEED_BARS)
                    sh.append(e_h)
                    sr.append(e_r)
                    continue
            else:
                e_h, e_r = rel_err(got, t), e_ref[k]
            ratios[k].append(e_h / (e_r + 1e-9))
            hip.append(e_h)
            ref.append(e_r)
        mh, mr = sorted(hip)[len(hip) // 2], sorted(ref)[len(ref) // 2]
        fam_ratio.append(mh / (mr + 1e-12))
        if sh:
            scal_ratio.append((sum(v * v for v in sh) / len(sh)) ** 0.5 / ((sum(v * v for v in sr) / len(sr)) ** 0.5 + 1e-12))
        lines.append(f"seed {seed}: logits {e_log:.2e} (ref {e_log_ref:.2e}), gradient medians {mh:.2e} vs {mr:.2e}")
    meds = {k: sorted(v)[len(v) // 2] for k, v in ratios.items() if v}
    worst = max(meds.items(), key=lambda kv: kv[1])
    over = sorted(((round(v, 2), k) for k, v in meds.items() if v > bars["per_param"]), reverse=True)
    fam = sorted(fam_ratio)[len(fam_ratio) // 2]
    sfam = sorted(scal_ratio)[len(scal_ratio) // 2] if scal_ratio else None
    evidence(f"{tag} over seeds {seeds} {dtype}: " + "; ".join(lines) + f"; per-parameter median-over-seeds ratio: worst {worst[1]:.2f} ({worst[0]}), "
             f"median {sorted(meds.values())[len(meds) // 2]:.2f}; family ratio per seed {[round(v, 2) for v in fam_ratio]} (median {fam:.2f})"
             + (f"; one-element parameters as a family, rms ratio per seed {[round(v, 2) for v in scal_ratio]}" if scal_ratio else "")
             + f"; bars: per parameter {bars['per_param']} x, family {bars['family']} x, logits {bars['logits']} x, no slack")
    assert measure or fam <= bars["family"], (tag, "median over seeds of the family ratio", fam_ratio)
    assert measure or sfam is None or sfam <= 1.5, (tag, "one-element parameters as a family", scal_ratio)
    assert measure or not over, (f"{len(over)} of {len(meds)} parameters: median over seeds of the distance ratio above {bars['per_param']}", over[:12])


@pytest.mark.gpu
@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_activation_checkpointing_changes_memory_not_results(drop):
    """use_checkpoint=True (reference swin_transformer_block.py:241-252): the Swin blocks drop their activations after the forward pass and
    run again in the backward pass.  Same logits bit for bit, same gradients (up to the order of the weight-gradient sums), less memory held
    between the passes; with dropout on (attention probabilities, MLP / projection outputs, stochastic depth) the second run must draw the
    masks of the first - the gradients then still match the uncheckpointed net's."""
    from mi_seg_amd.hip import ops
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.utils.detfill import det_input
    x = det_input(5, (1, 1, 64, 64, 64)).to(DEV)
    cot = det_input(6, (1, 6, 64, 64, 64)).to(DEV)
    res = {}
    for ck in (False, True):
        torch.manual_seed(1234)
        ops.DROP.seed, ops.DROP.step_dev = None, None      # the same dropout keys for both nets
        m = _fill(SwinUNETR((64, 64, 64), 1, 6, feature_size=24, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"), encoder_norm_name=_norm("instance_cond"),
                            decoder_norm_name=_norm("instance"), use_checkpoint=ck, drop_rate=drop, attn_drop_rate=drop, dropout_path_rate=drop))
        m.set_compute_dtype(torch.bfloat16).train()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        ops.begin_step()
        y = m(x, [1])
        held = torch.cuda.memory_allocated() - base          # what the tape keeps alive between the passes
        y.backward(cot)
        torch.cuda.synchronize()
        res[ck] = (y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, held)
        del m, y
    assert torch.equal(res[True][0], res[False][0]), "checkpointing must not change the forward pass"
    assert sorted(res[True][1]) == sorted(res[False][1])
    compare_grads(res[True][1], {k: v.float().cpu() for k, v in res[False][1].items()}, 2e-3 if drop else 5e-4, skip=ZERO_GRAD)      # (order of the weight-gradient sums: measured up to 1.1e-4)
    assert res[True][2] < 0.8 * res[False][2], (res[True][2], res[False][2])
    if drop == 0.0:      # ... and the checkpointed step records into a hipGraph and replays like the eager one (arena gradients)
        from mi_seg_amd.runtime.arena import ParamArena
        from mi_seg_amd.runtime.graph import GraphedStep
        m = _fill(SwinUNETR((64, 64, 64), 1, 6, feature_size=24, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"), encoder_norm_name=_norm("instance_cond"),
                            decoder_norm_name=_norm("instance"), use_checkpoint=True))
        m.set_compute_dtype(torch.bfloat16)
        arena = ParamArena([p for p in m.parameters() if p.requires_grad], torch.bfloat16)
        try:
            step = GraphedStep(m, x.shape, cot.shape, arena=arena)
            for _ in range(2):
                yg = step(x, [1], cot)
            torch.cuda.synchronize()
            assert torch.equal(yg.detach(), res[False][0])
            got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
            compare_grads(got, {k: v.float().cpu() for k, v in res[False][1].items()}, 5e-4, skip=ZERO_GRAD)
        finally:
            arena.detach()


@pytest.mark.gpu
def test_two_models_in_one_process_keep_their_own_step_state():
    """VERDICT round 2 (process-global launch state): two models with an arena each - a training net and, say, its EMA / validation twin -
    whose steps INTERLEAVE (forward A, forward B, backward A, backward B).  Each arena owns its step queues, its branch-deferral queue and
    (round 3) its statistics pool; both steps must give exactly what each model gives alone."""
    from mi_seg_amd.networks.nets.unetr import UNETR
    from mi_seg_amd.networks.nets.unet import UNet
    from mi_seg_amd.runtime.arena import ParamArena
    from mi_seg_amd.utils.detfill import fill_module_, det_input
    from mi_seg_amd.hip import ops
    a = UNETR(1, 3, (32, 32, 32), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4, pos_embed="perceptron", vit_norm_name=_norm("instance_cond"),
              encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")).cuda()
    b = UNet(3, 1, 3, channels=(8, 16, 32), strides=(2, 2), num_res_units=2, norm_down=_norm("instance_cond"), norm_up=_norm("instance")).cuda()
    for m in (a, b):
        fill_module_(m)
        m.set_compute_dtype(torch.bfloat16)
    x = det_input(3, (2, 1, 32, 32, 32)).cuda()
    cot = det_input(4, (2, 3, 32, 32, 32)).cuda()
    arenas = [ParamArena([p for p in m.parameters() if p.requires_grad], torch.bfloat16) for m in (a, b)]
    assert arenas[0].pool is not arenas[1].pool and arenas[0].pool is not ops.DEFAULT_POOL

    def alone(m, ar):
        out = None
        for _ in range(2):          # (the first step of an arena registers its weight re-layouts)
            ar.begin_step()
            y = m(x, [0, 1])
            y.backward(cot)
            ar.publish()
            out = y.detach().clone(), ar.flat.clone()
        return out
    ref = [alone(m, ar) for m, ar in zip((a, b), arenas)]
    arenas[0].begin_step()
    ya = a(x, [0, 1])
    arenas[1].begin_step()
    yb = b(x, [0, 1])
    ya.backward(cot)
    yb.backward(cot)
    arenas[0].publish()
    arenas[1].publish()
    for y, ar, (yr, gr) in zip((ya, yb), arenas, ref):
        assert torch.equal(y.detach(), yr)
        assert rel_err(ar.flat, gr) < 1e-5
    ops.use_pool(None)


@pytest.mark.gpu
def test_arena_fill_leaves_out_the_slots_a_kernel_overwrites():
    """round 5: the per-step zero fill of the gradient arena skips the slots whose weight-gradient kernel stores every element without
    reading it (the tiny-volume conv weights of encoder10 / decoder5: 70 % of the headline net's 249 MB).  Poisoned with NaN before every
    step, the arena must still come out equal to the fully filled one; a slot the fill left out and nothing wrote must read zero at the
    end of the step; another modality (other conditional-norm rows used) changes nothing about it."""
    from mi_seg_amd.hip import ops
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.runtime import arena as A
    from mi_seg_amd.utils.detfill import det_input, fill_module_
    net = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"), encoder_norm_name=_norm("instance_cond"),
                    decoder_norm_name=_norm("instance")).cuda()
    fill_module_(net)
    net.set_compute_dtype(torch.bfloat16)
    x, cot = det_input(3, (1, 1, 96, 96, 96)).cuda(), det_input(4, (1, 6, 96, 96, 96)).cuda()
    params = [p for p in net.parameters() if p.requires_grad]
    arena = A.ParamArena(params, torch.bfloat16)
    names = [k for k, p in net.named_parameters() if p.requires_grad]

    def step(mod, poison):
        if poison:
            arena.flat.fill_(float("nan"))          # whatever the previous step left behind must not matter
        arena.begin_step()
        net(x, [mod]).backward(cot)
        arena.publish()
        return arena.flat.clone()
    try:
        assert A.SKIP_OVERWRITTEN_FILL
        full = step(0, False)                        # first step: nothing is known yet, the whole arena is filled
        skipped = sorted(names[i] for i in arena._overwritten)
        assert skipped == sorted(["encoder10.layer.conv1.conv.weight", "encoder10.layer.conv2.conv.weight", "decoder5.conv_block.conv1.conv.weight",
                                  "decoder5.conv_block.conv2.conv.weight"]), skipped
        lean = step(0, True)
        assert bool(torch.isfinite(lean).all())
        assert rel_err(lean, full) < 1e-5            # (weight-gradient sums are order-dependent to <= 1e-5; the left-out slots are bit-identical)
        for i in arena._overwritten:
            assert torch.equal(arena.views[i], full[arena._offs[i]:arena._offs[i] + params[i].numel()].view(params[i].shape))
        other = step(1, True)                        # the other modality's norm rows: their slots are in the filled ranges
        assert bool(torch.isfinite(other).all())
        # a left-out slot that nothing writes: zero at the end of the step (the all-reduce of a data-parallel step reads it)
        i0 = arena._overwritten[0]
        arena.flat.fill_(float("nan"))
        arena.begin_step()
        assert arena.views[i0].data_ptr() in arena.queues.unzeroed and bool(torch.isnan(arena.views[i0]).all())
        j0 = next(j for j in range(len(params)) if j not in arena._overwritten and params[j].numel() > 1000)
        assert bool((arena.views[j0] == 0).all())
        arena.end_backward()
        assert bool((arena.flat == 0).all())
        assert arena._overwritten == []              # nothing was overwritten this step: the next fill is whole again
    finally:
        arena.detach()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["swin_unetr", "unetr", "unetr_conv", "unet"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_param_arena_matches_plain_autograd(dtype, kind):
    """arena mode (flat gradient buffer the kernels accumulate into, batched per-step weight re-layouts) must give the
    gradients of the plain path: same kernels, different destination; also across two steps and a skipped style.  All three nets:
    a parameter whose gradient autograd writes itself (round 2: UNETR's position table went through a `.to(dtype)` copy) is dropped by
    arena.publish() - the None pattern and every gradient are compared."""
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.networks.nets.unetr import UNETR
    from mi_seg_amd.networks.nets.unet import UNet
    from mi_seg_amd.runtime.arena import ParamArena
    from mi_seg_amd.utils.detfill import fill_module_, det_input
    from mi_seg_amd.hip import ops
    torch.manual_seed(0)
    S = 64 if kind == "swin_unetr" else 32
    if kind == "swin_unetr":
        net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"),
                        encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")).cuda()
    elif kind.startswith("unetr"):
        net = UNETR(1, 3, (32, 32, 32), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4, pos_embed="conv" if kind == "unetr_conv" else "perceptron",
                    vit_norm_name=_norm("instance_cond"),
                    encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")).cuda()
    else:
        net = UNet(3, 1, 3, channels=(8, 16, 32), strides=(2, 2), num_res_units=2, norm_down=_norm("instance_cond"), norm_up=_norm("instance")).cuda()
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

    y_ref, g_ref = plain([0, 0])
    y_ref2, g_ref2 = plain([0, 1])
    arena = ParamArena(params, dtype)
    try:
        for it, (mods, yr, gr) in enumerate([([0, 0], y_ref, g_ref), ([0, 0], y_ref, g_ref), ([0, 1], y_ref2, g_ref2)]):
            arena.begin_step()
            y = net(x, mods)
            y.backward(cot)
            arena.publish()
            if not torch.equal(y.detach(), yr):      # the forward is bit-reproducible: ANY difference fails (round 2 downgraded a non-reproducing
                # one to a warning; round 3: 60 fresh processes - scripts/debug/fresh_process_stress.py - and every fresh-box run of the suite
                # were clean).  The message carries what a diagnosis needs: where, how much, which side moved.
                d = (y.detach() - yr).abs()
                bad = d > 0
                where = bad.nonzero()
                msg = (f"logits differ in arena mode (step {it}, {kind}, {dtype}): {int(bad.sum())} of {d.numel()} elements, max {float(d.max()):.3e}, "
                       f"first at {where[0].tolist()}, last at {where[-1].tolist()}, per sample {[int(b_.sum()) for b_ in bad]}")
                arena.begin_step()
                y2 = net(x, mods).detach().clone()
                arena.publish()
                arena.detach()
                y3, _ = plain(mods)
                arena = ParamArena(params, dtype)
                msg += (f"; recomputed: arena again == arena first {bool(torch.equal(y2, y.detach()))}, arena again == plain first {bool(torch.equal(y2, yr))}, "
                        f"plain again == plain first {bool(torch.equal(y3, yr))}")
                raise AssertionError(msg)
            names = [k for k, _ in net.named_parameters()]
            assert [k for k, g in zip(names, gr) if g is None] == [k for k, p in zip(names, params) if p.grad is None]
            want = {k: g.float().cpu() for k, g in zip(names, gr) if g is not None}
            got = {k: p.grad for k, p in zip(names, params) if p.grad is not None}
            # same kernels, same forward bits: only the order of the weight-gradient reductions differs (bf16: a bias gradient of the small
            # UNet, a column sum of bf16 values, read 1.07e-3 once in 96 repetitions)
            compare_grads(got, want, 1e-3 if dtype == torch.float32 else 2e-3, skip=ZERO_GRAD)
        assert arena._table is not None and arena._table[1] > 0
    finally:
        arena.detach()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eager", "arena", "graph"])
def test_side_branch_matches_the_single_stream_step(mode):
    """SwinUNETR.side_branch (encoder1 + encoder2 taped in front of decoder2 on a branch stream, their backward beside the deeper blocks'
    with the convolution kernels in background form) against the same step on one stream: bit-identical logits, the same None pattern,
    gradients equal up to the order of the weight-gradient sums (the throttled weight gradient splits its rows differently)."""
    from mi_seg_amd.hip import ops
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.runtime.arena import ParamArena
    from mi_seg_amd.runtime.graph import GraphedStep
    from mi_seg_amd.utils.detfill import fill_module_, det_input
    net = SwinUNETR((96, 96, 96), 1, 3, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"),
                    encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")).cuda()
    fill_module_(net)
    net.set_compute_dtype(torch.bfloat16)
    x = det_input(3, (1, 1, 96, 96, 96)).cuda()
    cot = det_input(4, (1, 3, 96, 96, 96)).cuda()
    params = [p for p in net.parameters() if p.requires_grad]
    names = [k for k, _ in net.named_parameters()]

    def run(branch):
        net.side_branch = branch
        for p in params:
            p.grad = None
        arena = ParamArena(params, torch.bfloat16) if mode != "eager" else None
        bg0 = ops.BACKGROUND_LAUNCHES
        try:
            if mode == "graph":
                step = GraphedStep(net, x.shape, cot.shape, arena=arena)
                y = step(x, [1], cot)
                y = step(x, [1], cot).detach().clone()
            else:
                (arena.begin_step if arena is not None else ops.begin_step)()
                y = net(x, [1])
                y.backward(cot)
                if arena is not None:
                    arena.publish()
                y = y.detach().clone()
            torch.cuda.synchronize()
            return y, {k: (None if p.grad is None else p.grad.detach().float().cpu().clone()) for k, p in zip(names, params)}, ops.BACKGROUND_LAUNCHES - bg0
        finally:
            if arena is not None:
                arena.detach()

    y0, g0, n0 = run(False)
    y1, g1, n1 = run(True)
    # per eager step / capture: 3 data-gradient convolutions + the branch's 96^3 weight gradient; with an arena also decoder1's two deferred ones
    assert n0 == 0 and n1 >= (4 if mode == "eager" else 6), (n0, n1)
    assert torch.equal(y0, y1)
    assert [k for k in names if g0[k] is None] == [k for k in names if g1[k] is None]
    compare_grads({k: v for k, v in g1.items() if v is not None}, {k: v for k, v in g0.items() if v is not None}, 2e-3, skip=ZERO_GRAD)


def test_forked_norm_leaves_its_statistics_on_the_skip_branch():
    """round 5: the affine-less norm of a Swin stage's returned feature map (swin_transformer.py:135-136,150-158) and norm1 of the stage's first
    block (swin_transformer_block.py:241) read the SAME tensor - `instance_norm(x, fork=True)` hangs the sums of its own statistics pass on the
    skip branch and `norm_linear` folds norm1 into the qkv GEMM with them, no second statistics launch: same sums as a fresh pass, the same
    bits out of the GEMM, the same gradients (to the order of their atomic sums)."""
    from mi_seg_amd.hip import functional as HF
    from mi_seg_amd.hip import ops
    g = torch.Generator().manual_seed(11)
    x0 = (torch.randn(1, 16, 16, 24, 48, generator=g) * 1.5 + 0.3).to(DEV).to(torch.bfloat16)
    w = (torch.randn(144, 48, generator=g) / 7).to(DEV).requires_grad_(True)
    b = (torch.randn(144, generator=g) / 4).to(DEV).requires_grad_(True)
    gam = [(torch.randn(48, generator=g) * 0.2 + 1).to(DEV).requires_grad_(True) for _ in range(2)]
    bet = [(torch.randn(48, generator=g) * 0.3).to(DEV).requires_grad_(True) for _ in range(2)]
    styles = torch.tensor([1], dtype=torch.int32, device=DEV)
    cot_a, cot_q = torch.randn(1, 16, 16, 24, 48, generator=g).to(DEV).to(torch.bfloat16), torch.randn(1, 16, 16, 24, 144, generator=g).to(DEV).to(torch.bfloat16)
    outs = []
    for carry in (True, False):
        ops.begin_step()
        x = x0.clone().requires_grad_(True)
        for t in (w, b, *gam, *bet):
            t.grad = None
        a, xs = HF.instance_norm(x, None, fork=True)
        st = getattr(xs, "_miseg_stat", None)
        assert st is not None and torch.equal(st.sum(0), ops.instnorm_stats(x.detach(), 1, 16 * 16 * 24).sum(0))
        if not carry:
            del xs._miseg_stat
        r = HF.norm_linear(xs, list(zip(gam, bet)), styles, (1,), 1e-5, w, b, fork=True)
        assert r is not None
        q, xs2 = r
        torch.autograd.backward([a, q, xs2], [cot_a, cot_q, cot_a])
        outs.append([a.detach(), q.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone(), gam[1].grad.clone(), bet[1].grad.clone()])
        assert gam[0].grad is None
    for i, (u, v) in enumerate(zip(*outs)):      # the forward bits are the same; the gradients' sums run through float atomics (order varies run to run)
        assert torch.equal(u, v) if i < 2 else rel_err(u, v) < (1e-4 if i == 2 else 1e-6), i


@pytest.mark.gpu
@pytest.mark.parametrize("fork_at", ["s0", "s1", "s2"])
def test_side_branch_forked_beside_the_swin_stages_on_an_inner_tape(fork_at):
    """round 5 (SwinUNETR.fork_at, HF.inner_tape_join): the branch's FORWARD forked as soon as Swin feature map 0 / 1 / 2 is queued, its
    backward taped where it always was - against the default placement in a replayed hipGraph: bit-identical logits, the same None pattern,
    the same gradients (the kernels and their operands are the same; only where the host issues them differs)."""
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.runtime.arena import ParamArena
    from mi_seg_amd.runtime.graph import GraphedStep
    from mi_seg_amd.utils.detfill import fill_module_, det_input
    net = SwinUNETR((64, 64, 64), 1, 3, feature_size=24, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"),
                    encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")).cuda()
    fill_module_(net)
    net.set_compute_dtype(torch.bfloat16)
    x = det_input(3, (1, 1, 64, 64, 64)).cuda()
    cot = det_input(4, (1, 3, 64, 64, 64)).cuda()
    params = [p for p in net.parameters() if p.requires_grad]
    names = [k for k, _ in net.named_parameters()]

    def run(where):
        net.fork_at = where
        for p in params:
            p.grad = None
        arena = ParamArena(params, torch.bfloat16)
        try:
            step = GraphedStep(net, x.shape, cot.shape, arena=arena)
            step(x, [1], cot)
            y = step(x, [1], cot).detach().clone()
            torch.cuda.synchronize()
            return y, {k: (None if p.grad is None else p.grad.detach().float().cpu().clone()) for k, p in zip(names, params)}
        finally:
            arena.detach()

    y0, g0 = run("e10")
    y1, g1 = run(fork_at)
    assert torch.equal(y0, y1)
    assert [k for k in names if g0[k] is None] == [k for k in names if g1[k] is None]
    compare_grads({k: v for k, v in g1.items() if v is not None}, {k: v for k, v in g0.items() if v is not None}, 2e-3, skip=ZERO_GRAD)


@pytest.mark.gpu
@pytest.mark.parametrize("use_arena", [False, True, "split"])
def test_graphed_step_replays_match_eager(use_arena):
    """every replay of the captured step - not only the first - must reproduce the eager step, for both modalities and with
    host activity (allocations, host reads) between replays.  Regression: zero fills recorded as memset nodes were replayed
    with a stale argument block, so from the second replay on the statistics pool was not zero (logits off by 7e-3,
    some gradients inf)."""
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.runtime.arena import ParamArena
    from mi_seg_amd.runtime.graph import GraphedStep
    from mi_seg_amd.utils.detfill import fill_module_, det_input
    from mi_seg_amd.hip import ops
    net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"),
                    encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")).cuda()
    fill_module_(net)
    net.set_compute_dtype(torch.bfloat16)
    x = det_input(3, (1, 1, 64, 64, 64)).cuda()
    cot = det_input(4, (1, 3, 64, 64, 64)).cuda()
    params = [p for p in net.parameters() if p.requires_grad]
    names = [k for k, _ in net.named_parameters()]

    def eager(m):
        for p in params:
            p.grad = None
        ops.begin_step()
        y = net(x, [m])
        y.backward(cot)
        torch.cuda.synchronize()
        return y.detach().clone(), {k: p.grad.float().cpu() for k, p in zip(names, params) if p.grad is not None}

    ref = {m: eager(m) for m in (0, 1)}
    arena = ParamArena(params, torch.bfloat16) if use_arena else None
    try:
        # "split": the data-parallel variant - two graphs (forward + decoder-side backward, then the encoder side) with a hook in
        # between where bench.py starts the all-reduce of the decoder-side gradients; here the hook checks that those are final
        split = use_arena == "split"
        step = GraphedStep(net, x.shape, cot.shape, arena=arena, split=split)
        late, seen = None, []
        if split:
            tail = arena.tail_offset(net.late_backward_parameters())
            assert 0 < tail < arena.flat.numel()
            late = lambda: seen.append(arena.flat[tail:].clone())
        for it, m in enumerate([0, 0, 1, 0, 1, 1]):
            y = step(x, [m], cot, between=late) if split else step(x, [m], cot)
            if split:
                torch.cuda.synchronize()
                assert torch.equal(seen[-1], arena.flat[tail:]), "decoder-side gradients changed after the first graph"
            torch.cuda.synchronize()
            y_ref, g_ref = ref[m]
            # the inline host-side check is part of the regression: it allocates and launches between two replays
            assert torch.equal(y.detach(), y_ref), f"replay {it} (modality {m}): logits differ from eager"
            got = {k: p.grad for k, p in zip(names, params) if p.grad is not None}
            assert set(got) == set(g_ref), f"replay {it}: set of parameters with a gradient differs"
            assert all(bool(torch.isfinite(g).all()) for g in got.values()), f"replay {it}: non-finite gradient"
            compare_grads(got, g_ref, 1e-3, skip=ZERO_GRAD)
    finally:
        if arena is not None:
            arena.detach()


@pytest.mark.gpu
@pytest.mark.parametrize("defer", ["early", "late"])
def test_split_step_as_bench_times_it_at_n_gt_1(defer):
    """The step `bench.py --gpus N` (N > 1) times on the headline net (fs=48, 96^3, bf16): two hipGraphs, the side branch in the second one.
    defer = "late": decoder1's two deferred weight gradients are a HOLE in the decoder-side tail (`split_defers = True`, MISEG_SPLIT_DEFER=late):
    at the hook between the graphs - where bench.py starts RCCL on the early ranges - those ranges must be final and the hole must not be.
    defer = "early" (bench.py's default since round 3): the two launches run on the idle branch stream INSIDE the first half - the whole tail
    is final at the hook, one range.  After the second graph the whole arena must equal the single-graph step's (reference: tune.py:103-109,
    one DDP exchange per step over the same gradients)."""
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.runtime.arena import ParamArena
    from mi_seg_amd.runtime.graph import GraphedStep
    from mi_seg_amd.utils.detfill import fill_module_, det_input
    net = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=_norm("instance_cond"),
                    encoder_norm_name=_norm("instance_cond"), decoder_norm_name=_norm("instance")).cuda()
    fill_module_(net)
    net.set_compute_dtype(torch.bfloat16)
    assert net.side_branch
    x = det_input(3, (1, 1, 96, 96, 96)).cuda()
    cot = det_input(4, (1, 6, 96, 96, 96)).cuda()
    params = [p for p in net.parameters() if p.requires_grad]
    arena = ParamArena(params, torch.bfloat16)
    try:
        single = GraphedStep(net, x.shape, cot.shape, arena=arena)
        ref = {}
        for m in (0, 1):
            y = single(x, [m], cot)
            torch.cuda.synchronize()
            ref[m] = (y.detach().clone(), arena.flat.clone(), [p.grad is None for p in params])
        # bench.py's ranges: the decoder-side tail minus the hole goes out at the hook, the head and the hole after the second graph
        n = arena.flat.numel()
        tail = arena.tail_offset(net.late_backward_parameters())
        hole = arena.param_range(net.deferred_backward_parameters())
        assert 0 < tail <= hole[0] < hole[1] <= n
        early = [(hole[1], n), (tail, hole[0])] if defer == "late" else [(tail, n)]
        net.split_defers = True if defer == "late" else "early"
        split = GraphedStep(net, x.shape, cot.shape, arena=arena, split=True)
        seen = []
        hook = lambda: seen.append(([arena.flat[lo:hi].clone() for lo, hi in early], arena.flat[hole[0]:hole[1]].clone(), arena.flat[:tail].clone()))
        for it, m in enumerate([1, 0, 0, 1]):
            y = split(x, [m], cot, between=hook)
            torch.cuda.synchronize()
            at_hook, hole_at_hook, head_at_hook = seen[-1]
            for (lo, hi), t in zip(early, at_hook):
                assert torch.equal(t, arena.flat[lo:hi]), f"replay {it}: arena[{lo}:{hi}] changed after the hook - it is not final there"
                assert float(t.abs().max()) > 0
            # the deferred weight gradients and the encoder / Swin side are produced by the second graph
            if defer == "late":
                assert float(hole_at_hook.abs().max()) == 0.0 and float(arena.flat[hole[0]:hole[1]].abs().max()) > 0, "the hole must be written by the second graph"
            else:
                assert torch.equal(hole_at_hook, arena.flat[hole[0]:hole[1]]) and float(hole_at_hook.abs().max()) > 0, "decoder1's gradients must be final at the hook"
            assert float(head_at_hook.abs().max()) == 0.0 and float(arena.flat[:tail].abs().max()) > 0
            y_ref, g_ref, none_ref = ref[m]
            assert torch.equal(y.detach(), y_ref), f"replay {it}: logits differ from the single-graph step"
            assert [p.grad is None for p in params] == none_ref
            assert bool(torch.isfinite(arena.flat).all())
            # decoder side (everything bench.py sends at the hook + the hole): the same kernels on the same inputs - equal up to the order
            # of the weight-gradient sums (<= 1e-5).  Encoder / Swin side: the cut tensors' gradients reach it as ONE
            # bf16 tensor instead of joining the fan-out sums one by one, i.e. a different bf16 rounding of the same sum - the tolerance of
            # the other graph-vs-eager tests
            names = [k for k, _ in net.named_parameters()]
            for k, p, v in zip(names, params, arena.views):
                if p.grad is not None and k not in ZERO_GRAD:
                    lo = v.storage_offset()
                    a, b = arena.flat[lo:lo + p.numel()], g_ref[lo:lo + p.numel()]
                    tol = 2e-5 if lo >= tail else 2e-3
                    err = float((a - b).norm()) / (float(b.norm()) + 1e-30)
                    assert err <= tol, f"replay {it}: gradient of {k} differs from the single graph by {err:.2e} (arena offset {lo}, tail starts at {tail})"
    finally:
        net.split_defers = False
        arena.detach()
