"""Pins the CPU oracle (oracle/) against golden vectors captured from the reference's own modules
(oracle/tools/make_golden.py).  CPU only."""
import pytest
import torch

from conftest import rel_err, sample, state_from_meta
from oracle import functional as OF
from oracle import nets as ON

TOL = 2e-5  # same fp32 math, different op order


def _leaf(t):
    return t.clone().requires_grad_(True)


def _check_grads(sd, prefix, want, none_list, tol=TOL):
    for k, g in want.items():
        got = sd[prefix + k].grad
        assert got is not None, k
        assert rel_err(got, g) < tol, (k, rel_err(got, g))
    for k in none_list:
        assert sd[prefix + k].grad is None, f"{k} should have no gradient"


@pytest.mark.parametrize("tag", ["3d_mixed", "3d_same", "3d_tensor_styles", "1d_mixed"])
def test_cond_instnorm(golden, tag):
    from mi_seg_amd.utils.detfill import det_values
    G = golden("cond_instnorm")
    case = G.meta["cases"][tag]
    C = case["shape"][1]
    sd = {f"norms.{s}.{p}": torch.from_numpy(det_values(f"norms.{s}.{p}", (C,))).requires_grad_(True)
          for s in range(2) for p in ("weight", "bias")}
    x = _leaf(G.t(f"{tag}/x"))
    y = OF.norm_channels_first(sd, "", "instance_cond", x, case["styles"])
    y.backward(G.t(f"{tag}/g"))
    assert rel_err(y, G.t(f"{tag}/y")) < TOL
    assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
    _check_grads(sd, "", G.grads(tag), case["grad_none"])


def test_cond_instnorm_errors():
    x = torch.zeros(2, 3, 4, 4, 4)
    with pytest.raises(ValueError):
        OF.cond_instance_norm(x, [0], [torch.ones(3)] * 2, [torch.zeros(3)] * 2)
    with pytest.raises(ValueError):
        OF.norm_channels_first({}, "", "instance_cond", x, None)


def test_mask_and_index(golden):
    G = golden("window_attention")
    assert torch.equal(OF.relative_position_index(), G.t("relative_position_index", torch.int64))
    assert torch.equal(OF.compute_mask((14, 7, 7), (7, 7, 7), (3, 3, 3)), G.t("mask_14_7_7"))
    idx = OF.relative_position_index()
    assert int(idx[0, 0]) == 1098 and int(idx.max()) == 2196


@pytest.mark.parametrize("tag", ["n343_nomask", "n343_mask", "n216_sliced"])
def test_window_attention(golden, tag):
    from mi_seg_amd.utils.detfill import det_values
    G = golden("window_attention")
    case = G.meta["cases"][tag]
    shapes = {"relative_position_bias_table": (2197, 3), "qkv.weight": (36, 12), "qkv.bias": (36,),
              "proj.weight": (12, 12), "proj.bias": (12,)}
    sd = {k: torch.from_numpy(det_values(k, s)).requires_grad_(True) for k, s in shapes.items()}
    sd["relative_position_index"] = OF.relative_position_index()
    x = _leaf(G.t(f"{tag}/x"))
    mask = G.t("mask_14_7_7") if case["mask"] else None
    y = OF.window_attention(sd, "", x, mask, case["heads"])
    y.backward(G.t(f"{tag}/g"))
    assert rel_err(y, G.t(f"{tag}/y")) < TOL
    assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
    _check_grads(sd, "", G.grads(tag), [])


def _block_sd(G, tag):
    from mi_seg_amd.utils.detfill import det_values
    sd = {}
    for k, g in G.grads(tag).items():
        sd[k] = torch.from_numpy(det_values(k, g.shape)).requires_grad_(True)
    for k in G.meta["cases"][tag].get("grad_none", []):
        sd[k] = None  # shape unknown here; filled by caller when needed
    return sd


@pytest.mark.parametrize("tag", ["pad_noshift", "pad_shift", "clamped6", "layer_shift", "inst_noshift"])
def test_swin_block(golden, tag):
    G = golden("swin_block")
    case = G.meta["cases"][tag]
    sd = _block_sd(G, tag)
    sd["attn.relative_position_index"] = OF.relative_position_index()
    x = _leaf(G.t(f"{tag}/x"))
    d, h, w = case["dhw"]
    ws, ss = OF.get_window_size((d, h, w), (7, 7, 7), (3, 3, 3))
    pads = [-(-a // b) * b for a, b in zip((d, h, w), ws)]
    mask = OF.compute_mask(pads, ws, ss)
    y = OF.swin_block(sd, "", x, mask, case["modalities"], case["heads"], (7, 7, 7), tuple(case["shift"]), case["norm"])
    y.backward(G.t(f"{tag}/g"))
    assert rel_err(y, G.t(f"{tag}/y")) < TOL
    assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
    _check_grads(sd, "", G.grads(tag), [])


@pytest.mark.parametrize("tag", ["merging", "mergingv2", "merging_odd_layer"])
def test_patch_merging(golden, tag):
    G = golden("patch_merging")
    case = G.meta["cases"][tag]
    sd = _block_sd(G, tag)
    x = _leaf(G.t(f"{tag}/x"))
    y = OF.patch_merging(sd, "", x, case["modalities"], case["norm"], "mergingv2" if tag == "mergingv2" else "merging")
    y.backward(G.t(f"{tag}/g"))
    assert rel_err(y, G.t(f"{tag}/y")) < TOL
    assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
    _check_grads(sd, "", G.grads(tag), [])
    if tag == "merging":   # duplicated slices: offsets (1,1,0) and (0,1,1) never reach the output
        assert float(x.grad[:, 1::2, 1::2, 0::2].abs().max()) == 0.0
        assert float(x.grad[:, 0::2, 1::2, 1::2].abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["res_8_to_12_cond", "res_8_to_8_cond", "res_1_to_8_inst", "up_16_to_8_inst",
                                 "prup_16_to_8_cond"])
def test_unetr_blocks(golden, tag):
    G = golden("unetr_blocks")
    case = G.meta["cases"][tag]
    sd = _block_sd(G, tag)
    x = _leaf(G.t(f"{tag}/x"))
    mod = case["modalities"]
    kind = "instance_cond" if tag.endswith("cond") else "instance"
    if tag.startswith("res"):
        y = OF.unet_res_block(sd, "", x, mod, kind)
    elif tag.startswith("up"):
        skip = _leaf(G.t(f"{tag}/skip"))
        y = OF.unetr_up_block(sd, "", x, skip, mod, kind)
    else:
        y = OF.unetr_pr_up_block(sd, "", x, mod, kind, 1)
    y.backward(G.t(f"{tag}/g"))
    assert rel_err(y, G.t(f"{tag}/y")) < TOL
    assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
    if tag.startswith("up"):
        assert rel_err(skip.grad, G.t(f"{tag}/dskip")) < TOL
    _check_grads(sd, "", G.grads(tag), [])


@pytest.mark.parametrize("tag", ["cond", "layer_bias"])
def test_transformer_block(golden, tag):
    G = golden("transformer_block")
    case = G.meta["cases"][tag]
    sd = _block_sd(G, tag)
    x = _leaf(G.t(f"{tag}/x"))
    y = OF.transformer_block(sd, "", x, case["modalities"], case["heads"], case["norm"])
    y.backward(G.t(f"{tag}/g"))
    assert rel_err(y, G.t(f"{tag}/y")) < TOL
    assert rel_err(x.grad, G.t(f"{tag}/dx")) < TOL
    _check_grads(sd, "", G.grads(tag), [])


def _whole(G, tag, fwd, cfg, tol=5e-5):
    from mi_seg_amd.utils.detfill import det_input
    case = G.meta["cases"][tag]
    sd = state_from_meta(case)
    x = det_input(1234, case["x"])
    y = fwd(sd, x, case["modalities"], cfg)
    assert rel_err(sample(y), G.t(f"{tag}/logits_samples")) < tol
    assert abs(float(y.double().norm()) / float(G.z[f"{tag}/logits_l2"]) - 1) < tol
    y.backward(det_input(4321, tuple(y.shape)))
    gn = G.gnorms(tag)
    worst = 0.0
    for k, g in G.grads(tag).items():
        got = sd[k].grad
        assert got is not None, k
        e = rel_err(sample(got), g)
        worst = max(worst, e)
        assert e < 20 * tol, (k, e)
        assert abs(float(got.double().norm()) / (gn[k] + 1e-30) - 1) < 20 * tol, k
    none = [k for k, v in sd.items() if torch.is_floating_point(v) and v.grad is None]
    assert sorted(none) == sorted(case["grad_none"])
    return worst


@pytest.mark.parametrize("tag", ["fs12_64_m10", "fs12_64_v2_layer"])
def test_swin_unetr_small(golden, tag):
    G = golden("swin_unetr_small")
    c = G.meta["cases"][tag]
    cfg = ON.swin_unetr_cfg(feature_size=12, downsample=c["downsample"], vit_norm=c["vit_norm"],
                            encoder_norm=c["encoder_norm"], decoder_norm=c["decoder_norm"])
    _whole(G, tag, ON.swin_unetr_forward, cfg)


def test_unetr_small(golden):
    G = golden("unetr_small")
    cfg = ON.unetr_cfg(img_size=(32, 32, 32), feature_size=8, hidden_size=48, mlp_dim=96, num_heads=4)
    _whole(G, "small_32", ON.unetr_forward, cfg)


@pytest.mark.parametrize("tag", ["c1_64", "cond_32"])
def test_unet(golden, tag):
    G = golden("unet")
    c = G.meta["cases"][tag]
    cfg = ON.unet_cfg(channels=c["channels"], strides=c["strides"], num_res_units=c["num_res_units"],
                      norm_down=c.get("norm_down", "instance"))
    _whole(G, tag, ON.unet_forward, cfg)


def test_c2_state_layout(golden):
    """Checkpoint drop-in: 273 entries / 62,218,200 parameters for the headline config."""
    G = golden("swin_unetr_c2")
    c = G.meta["cases"]["c2_m0"]
    assert c["n_state"] == 273 and c["n_params"] == 62218200
    assert len(c["grad_none"]) == 62          # all norms.1.* when the batch holds modality 0 only (A6)
    assert all(".norms.1." in k for k in c["grad_none"])


@pytest.mark.parametrize("tag", ["c2_m0"])
def test_swin_unetr_c2_full(golden, tag):
    """Headline config at full size on the CPU oracle (about 40 s): both cotangents of the fixture."""
    from mi_seg_amd.utils.detfill import ce_cotangent, det_input
    G = golden("swin_unetr_c2")
    case = G.meta["cases"][tag]
    sd = state_from_meta(case)
    y = ON.swin_unetr_forward(sd, det_input(1234, case["x"]), case["modalities"], ON.swin_unetr_cfg(feature_size=48))
    assert rel_err(sample(y), G.t(f"{tag}/logits_samples")) < 5e-5
    y.backward(det_input(4321, tuple(y.shape)), retain_graph=True)
    for k, g in G.grads(tag).items():
        assert rel_err(sample(sd[k].grad), g) < 1e-3, k
    for v in sd.values():
        v.grad = None
    y.backward(ce_cotangent(y))
    for k, g in G.grads2(tag).items():
        assert rel_err(sample(sd[k].grad), g) < 1e-3, k
