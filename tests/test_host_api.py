"""CPU tests of the host-side drop-in surface: parser defaults, model factory, state_dict layouts, error conventions, C-ABI symbol
table, rank sharding, sliding-window geometry.  No GPU compute is launched."""
import argparse
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(extra=()):
    from mi_seg_amd.utils.parser import add_data_argparse_args, add_model_argparse_args, add_tune_argparse_args
    p = argparse.ArgumentParser()
    add_tune_argparse_args(add_data_argparse_args(add_model_argparse_args(p)))
    return p.parse_args(list(extra))


def test_parser_defaults_match_reference():
    a = _args()
    assert (a.model_name, a.roi_x, a.feature_size, a.hidden_size, a.mlp_dim, a.num_heads, a.pos_embed) == ("unetr", 96, [16], 768, 3072, 12, "perceptron")
    assert (a.vit_norm_name, a.encoder_norm_name, a.decoder_norm_name, a.num_styles, a.depth_swin_block, a.downsample) == \
        ("layer", "instance", "instance", 2, [2], "merging")
    assert (a.num_layers, a.strides, a.num_res_units, a.activation, a.criterion, a.optim_name, a.lr, a.reg_weight) == \
        (4, [2, 2, 2], 2, "prelu", "dice_focal", "adamw", 1e-4, 1e-5)
    assert (a.infer_overlap, a.sw_batch_size, a.scheduler, a.iters_to_accumulate) == (0.5, 1, "reduce_on_plateau", 1)


def test_model_factory_headline_config(golden):
    """README.md:170-173 recipe -> SwinUNETR with the reference's 273-entry state_dict."""
    from mi_seg_amd.networks.utils.utils import model_from_argparse_args
    a = _args(["--model_name=swin_unetr", "--out_channels=6", "--feature_size=48", "--num_heads=3", "--encoder_norm_name=instance_cond",
               "--vit_norm_name=instance_cond"])
    m = model_from_argparse_args(a)
    case = golden("swin_unetr_c2").meta["cases"]["c2_m0"]
    assert list(m.state_dict().keys()) == case["state_keys"]
    assert [list(v.shape) for v in m.state_dict().values()] == case["state_shapes"]
    assert sum(p.numel() for p in m.parameters()) == 62218200
    with pytest.raises(ValueError):
        model_from_argparse_args(_args(["--model_name=nope"]))


def test_unetr_and_unet_layouts(golden):
    from mi_seg_amd.networks.utils.utils import model_from_argparse_args
    m = model_from_argparse_args(_args(["--model_name=unetr", "--out_channels=6", "--encoder_norm_name=instance_cond", "--vit_norm_name=instance_cond"]))
    case = golden("unetr_c3").meta["cases"]["c3_m1"]
    assert list(m.state_dict().keys()) == case["state_keys"] and sum(p.numel() for p in m.parameters()) == 92824966
    u = model_from_argparse_args(_args(["--model_name=unet", "--out_channels=6"]))
    case = golden("unet").meta["cases"]["c1_64"]
    assert list(u.state_dict().keys()) == case["state_keys"] and sum(p.numel() for p in u.parameters()) == 4749969
    assert u.channels == [32, 64, 128, 256]                 # fs * 2**i, i = 1..num_layers (reference unet.py:218-219)
    with pytest.raises(RuntimeError):                        # no CPU fallback: the HIP path refuses host tensors
        u(torch.zeros(1, 1, 8, 8, 8))


def test_constructor_errors():
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.networks.norms.utils import parse_normalization
    with pytest.raises(ValueError):
        SwinUNETR((96, 96, 96), 1, 6, feature_size=50)
    with pytest.raises(ValueError):
        SwinUNETR((100, 96, 96), 1, 6, feature_size=48)
    with pytest.raises(ValueError):
        SwinUNETR((96, 96, 96), 1, 6, feature_size=48, encoder_norm_name="layer")
    with pytest.raises(ValueError):
        parse_normalization("nope", True)
    assert parse_normalization("instance_cond", True, 4, 2) == ("instance_cond", {"num_styles": 2, "affine": True})


def test_use_checkpoint_reaches_every_swin_block():
    """--use_checkpoint (reference utils/parser.py:24, swin_transformer_block.py:241-252): every Swin block is built with the flag (round 4:
    applied - the block runs again in the backward pass; tests/test_hip_modules.py::test_activation_checkpointing_changes_memory_not_results)"""
    import warnings
    from mi_seg_amd.networks.blocks.swin_transformer_block import SwinTransformerBlock
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m = SwinUNETR((32, 32, 32), 1, 2, feature_size=12, use_checkpoint=True)
    blocks = [b for b in m.modules() if isinstance(b, SwinTransformerBlock)]
    assert len(blocks) == 8 and all(b.use_checkpoint for b in blocks)
    assert not any(b.use_checkpoint for b in SwinUNETR((32, 32, 32), 1, 2, feature_size=12).modules() if isinstance(b, SwinTransformerBlock))


def test_no_cpu_fallback():
    """the product path fails loudly without a HIP device instead of computing on the CPU."""
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    m = SwinUNETR((64, 64, 64), 1, 6, feature_size=12)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 64, 64, 64), None)
    from mi_seg_amd.hip import lib, ops
    with pytest.raises(lib.MisegHipError):
        ops.rows(torch.zeros(4, 4))


def test_c_abi_exports_every_declared_symbol():
    """every function prototype of include/miseg_hip.h is exported by the shared object and bound in hip/lib.py."""
    from mi_seg_amd.hip import lib
    hdr = open(os.path.join(ROOT, "include", "miseg_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(miseg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    assert os.path.exists(lib.LIB_PATH), "run `python __graft_entry__.py` first"
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(so, name), f"{name} declared in include/miseg_hip.h but not exported"
        assert name in lib.PROTOS, f"{name} has no ctypes prototype in hip/lib.py"
    assert set(lib.PROTOS) <= declared
    so.miseg_abi_version.restype = ctypes.c_int
    assert so.miseg_abi_version() == lib.ABI_VERSION


def test_rank_sharding_matches_distributed_sampler():
    from torch.utils.data import DistributedSampler

    from mi_seg_amd.data.sampler import concat_modalities, rank_indices
    n = 16 + 16
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            s = DistributedSampler(list(range(n)), num_replicas=world, rank=r, shuffle=True, seed=0)
            s.set_epoch(3)
            mine = rank_indices(n, world, r, epoch=3, seed=0)
            assert list(iter(s)) == mine
            seen += mine
        assert sorted(seen) == list(range(n))
    mods = concat_modalities(16, 16)
    stream = [mods[i] for i in rank_indices(n, 8, 0, epoch=0)]
    assert set(stream) <= {0, 1}


def test_sliding_window_geometry_and_stitching():
    from mi_seg_amd.training.inferer import sliding_window_inference, window_grid
    grid = window_grid((512, 512, 363), (96, 96, 96), 0.5)
    assert len(grid) == 700                                   # BASELINE config 5: 10 x 10 x 7 windows
    assert max(g[2] for g in grid) == 363 - 96 and max(g[0] for g in grid) == 512 - 96
    x = torch.randn(2, 1, 40, 37, 20)
    calls = []

    def predictor(win, mods):
        calls.append((tuple(win.shape), list(mods)))
        return torch.cat([win * 2.0, win + 1.0], 1)

    out = sliding_window_inference(x, (16, 16, 16), 4, predictor, overlap=0.5, modalities=torch.tensor([1, 0]))
    assert torch.allclose(out[:, 0:1], 2 * x, atol=1e-5) and torch.allclose(out[:, 1:2], x + 1, atol=1e-5)
    assert all(len(set(m)) == 1 for _, m in calls) and calls[0][1][0] == 1 and calls[-1][1][0] == 0
    small = sliding_window_inference(torch.randn(1, 1, 10, 12, 16), (16, 16, 16), 1, lambda w, m: w, modalities=[0])
    assert small.shape == (1, 1, 10, 12, 16)


def test_losses_and_metric_semantics():
    from mi_seg_amd.training.losses import DiceCELoss, DiceFocalLoss
    from mi_seg_amd.training.metrics import as_discrete_argmax_onehot, as_discrete_onehot, dice_metric
    torch.manual_seed(0)
    label = torch.randint(0, 3, (2, 1, 6, 6, 6))
    perfect = as_discrete_onehot(label, 3) * 40.0 - 20.0
    lf = DiceFocalLoss(include_background=True, to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr=0.0, smooth_dr=1e-6)
    assert float(lf(perfect, label)) < 1e-3
    assert float(lf(-perfect, label)) > 1.0
    lc = DiceCELoss(include_background=False, to_onehot_y=True, softmax=True)
    assert float(lc(perfect, label)) < 1e-3
    d = dice_metric(as_discrete_argmax_onehot(perfect, 3), as_discrete_onehot(label, 3))
    assert torch.allclose(d, torch.ones_like(d))
    lab2 = torch.zeros(1, 1, 4, 4, 4, dtype=torch.long)
    d2 = dice_metric(as_discrete_onehot(lab2, 3), as_discrete_onehot(lab2, 3))
    assert float(d2[0, 0]) == 1.0 and torch.isnan(d2[0, 1])   # NaN when the class is absent from the label


def test_dice_focal_strips_background_before_the_softmax():
    """MONAI 1.1.0 DiceFocalLoss(include_background=False): channel 0 is removed from logits AND one-hot target first and the sub-losses
    (built without include_background) see C-1 channels -- so the Dice softmax runs over the foreground logits only.  Independent float64
    numpy restatement on a tiny case (the reference's documented launches use --no_include_background: lightning_monai.py:44,49-55)."""
    import numpy as np
    from mi_seg_amd.training.losses import DiceFocalLoss
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 4, 3, 2, 2)) * 2
    lab = rng.integers(0, 4, size=(1, 1, 3, 2, 2))
    xs = x[:, 1:]
    t = np.stack([(lab[:, 0] == c) for c in range(1, 4)], 1).astype(np.float64)
    e = np.exp(xs - xs.max(1, keepdims=True))
    p = e / e.sum(1, keepdims=True)
    ax = (2, 3, 4)
    dice = (1 - (2 * (p * t).sum(ax)) / ((t * t).sum(ax) + (p * p).sum(ax) + 1e-6)).mean()
    ce = xs - xs * t + np.log1p(np.exp(-np.abs(xs))) + np.maximum(-xs, 0)
    z = -xs * (2 * t - 1)
    logsig = -(np.log1p(np.exp(-np.abs(z))) + np.maximum(-z, 0))
    focal = (np.exp(2.0 * logsig) * ce).reshape(1, 3, -1).mean(-1).mean()
    crit = DiceFocalLoss(include_background=False, to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr=0.0, smooth_dr=1e-6)
    got = float(crit(torch.from_numpy(x), torch.from_numpy(lab)))
    assert abs(got - (dice + focal)) < 1e-6          # (the focal term of the torch restatement runs in fp32, like MONAI's .float())
    # and it differs from "softmax over all channels, then drop channel 0" (what DiceCELoss's DiceLoss does)
    e4 = np.exp(x - x.max(1, keepdims=True))
    p4 = (e4 / e4.sum(1, keepdims=True))[:, 1:]
    other = (1 - (2 * (p4 * t).sum(ax)) / ((t * t).sum(ax) + (p4 * p4).sum(ax) + 1e-6)).mean() + focal
    assert abs(got - other) > 1e-3


def test_style_ids_are_range_checked_on_the_host():
    """the reference indexes a ModuleList with the style id (IndexError outside [-n, n)); the kernels index argument arrays on the device,
    so the host check is what stands between a bad id and a wild pointer"""
    from mi_seg_amd.networks.norms.conditional_instance_norm import ConditionalInstanceNorm3d, styles_to_device
    dev, host = styles_to_device([1, -1, 0], "cpu", 3, num_styles=2)
    assert host == (1, 1, 0) and dev.tolist() == [1, 1, 0]
    for bad in ([2, 0, 0], [0, -3, 0], torch.tensor([0, 0, 4])):
        with pytest.raises(IndexError):
            styles_to_device(bad, "cpu", 3, num_styles=2)
    with pytest.raises(IndexError):
        styles_to_device((torch.zeros(2, dtype=torch.int32), (0, 2)), "cpu", 2, num_styles=2)
    with pytest.raises(NotImplementedError):
        ConditionalInstanceNorm3d(5, 8)


def test_litmonai_surface():
    from mi_seg_amd.networks.lightning_monai import LitMonai
    a = _args(["--model_name=swin_unetr", "--out_channels=6", "--feature_size=12", "--num_heads=3", "--roi_x=64", "--roi_y=64", "--roi_z=64"])
    lit = LitMonai.from_argparse_args(a)
    conf = lit.configure_optimizers()
    assert isinstance(conf["optimizer"], torch.optim.AdamW) and conf["lr_scheduler"]["monitor"] == "val/loss/avg"
    with pytest.raises(ValueError):
        LitMonai(lit.model, 6, criterion="nope")


@pytest.mark.parametrize("workload", ["c2", "c3"])
def test_cpu_baseline_case_builds(workload):
    """bench.py's cpu_baseline leg builds the product model on the meta device for its key list (a `torch.linspace(...).item()` in a
    constructor broke exactly that once): both timed workloads must construct, with every oracle weight present and finite"""
    from mi_seg_amd.testing.cpu_baseline import baseline_case, host_cpu
    sd, cfg, fwd, x, g = baseline_case(workload)
    assert len(sd) == (273 if workload == "c2" else 280)
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
    assert tuple(x.shape) == (1, 1, 96, 96, 96) and tuple(g.shape) == (1, 6, 96, 96, 96) and callable(fwd)
    name, phys, logical = host_cpu()
    assert phys >= 1 and logical >= phys and isinstance(name, str)


def test_bench_refuses_to_time_fewer_ranks_than_asked_for():
    """`python bench.py --gpus N` with no launcher starts the N ranks itself (bench.py::self_launch) - and on a node with fewer than N GPUs
    it exits non-zero with a message instead of silently timing one rank and printing n_gpus = 1.  No GPU is touched."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MISEG_REHEARSE_ONE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(have + 2), "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert f"--gpus {have + 2} but this node shows {have} GPU(s)" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no JSON line may be printed"


def test_window_grid_against_the_independent_restatement():
    """VERDICT round 4 (weak 13): the product's window grid (closed form, training/inferer.py::window_grid / _starts) against an independent
    restatement of MONAI 1.1.0's scan (oracle/sliding_window.py walks the scan positions one by one) - at BASELINE configs[4]'s
    512 x 512 x 363, at odd sizes, at sizes equal to / below the roi and at several overlaps; and the CPU stitching loop of the product against
    the oracle's loop on a toy predictor, padding case included."""
    from mi_seg_amd.training.inferer import sliding_window_inference, window_grid
    from oracle import sliding_window as OSW
    grid = window_grid((512, 512, 363), (96, 96, 96), 0.5)
    assert grid == OSW.window_origins((512, 512, 363), (96, 96, 96), 0.5) and len(grid) == 700
    assert grid[0] == (0, 0, 0) and grid[-1] == (416, 416, 267) and grid[1] == (0, 0, 48) and grid[6] == (0, 0, 267)
    n = 0
    for size in ((96, 96, 96), (97, 96, 130), (191, 193, 95 + 96), (100, 143, 289), (160, 160, 128), (96, 80 + 96, 333), (250, 96, 97)):
        for roi in ((96, 96, 96), (64, 64, 64), (64, 96, 32)):
            if any(s < r for s, r in zip(size, roi)):
                continue
            for ov in (0.5, 0.25, 0.7, 0.0, 0.99):
                assert window_grid(size, roi, ov) == OSW.window_origins(size, roi, ov), (size, roi, ov)
                n += 1
    assert n > 80
    # the whole CPU path (pad, grid, accumulate, divide, crop) on a predictor whose output depends on the window's content and position in it
    g = torch.Generator().manual_seed(3)

    def pred(x, *a, **k):
        ramp = torch.linspace(0.0, 1.0, x.shape[-1]).view(1, 1, 1, 1, -1)
        return torch.cat([x * 2.0 + ramp, x.flip(-1) - 0.5 * ramp, x.mean(dim=(2, 3, 4), keepdim=True).expand_as(x)], 1)

    for size, roi, ov in (((40, 37, 50), 16, 0.5), ((20, 33, 12), (16, 16, 16), 0.25), ((16, 16, 47), 16, 0.5), ((10, 16, 21), (16, 8, 8), 0.5)):
        vol = torch.rand((2, 1) + size, generator=g)
        want = OSW.sliding_window_reference(vol, roi, pred, overlap=ov)
        got = sliding_window_inference(vol, roi, 3, pred, overlap=ov)
        assert got.shape == want.shape == (2, 3) + size
        assert torch.allclose(got, want, rtol=0, atol=1e-6), (size, roi, ov, float((got - want).abs().max()))
