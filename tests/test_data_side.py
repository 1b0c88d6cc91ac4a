"""SURVEY.md 8(f) row 4, the data side: Decathlon JSON + modality reader (reference data/utils.py:11-60), NIfTI-1 reader + RAS
reorientation (data/multi_modal.py:37-40), checkpoint import / export (tune.py:27-38), GPU-resident augmentation (data/multi_modal.py:50-65)."""
import json
import os

import numpy as np
import pytest
import torch


def test_decathlon_datalist_with_modality(tmp_path):
    from mi_seg_amd.data.decathlon import load_decathlon_datalist_with_modality, modality_id
    js = {"modality": {"0": "CT"}, "training": [{"image": "imagesTr/ct_1.nii.gz", "label": "labelsTr/ct_1.nii.gz"}, {"image": "/abs/ct_2.nii.gz", "label": "l2.nii.gz"}],
          "test": ["imagesTs/ct_9.nii.gz"]}
    p = tmp_path / "dataset.json"
    p.write_text(json.dumps(js))
    items = load_decathlon_datalist_with_modality(str(p))
    assert items[0] == {"image": str(tmp_path / "imagesTr/ct_1.nii.gz"), "label": str(tmp_path / "labelsTr/ct_1.nii.gz"), "modality": {"0": "CT"}}
    assert items[1]["image"] == "/abs/ct_2.nii.gz" and items[1]["label"] == str(tmp_path / "l2.nii.gz")
    assert modality_id(items[0]["modality"]) == 0 and modality_id("MRI") == 1 and modality_id("mr") == 1
    other = load_decathlon_datalist_with_modality(str(p), base_dir="/data")
    assert other[0]["image"] == "/data/imagesTr/ct_1.nii.gz"
    with pytest.raises(ValueError, match="does not exist"):
        load_decathlon_datalist_with_modality(str(tmp_path / "nope.json"))
    with pytest.raises(ValueError, match="not specified"):
        load_decathlon_datalist_with_modality(str(p), data_list_key="validation")
    with pytest.raises(ValueError):
        modality_id("PET")


@pytest.mark.parametrize("dtype", [np.int16, np.float32, np.uint8])
def test_nifti_round_trip_and_ras(tmp_path, dtype):
    from mi_seg_amd.data.nifti import read_nifti, reorient_to_ras, write_nifti
    rng = np.random.default_rng(0)
    vol = (rng.normal(size=(5, 6, 7)) * 50).astype(dtype)
    # voxel axis 0 runs along -Y (posterior), axis 1 along +X, axis 2 along -Z: an "PRI"-like scan
    A = np.array([[0, 1.2, 0, -10.0], [-0.8, 0, 0, 4.0], [0, 0, -2.5, 30.0], [0, 0, 0, 1.0]])
    for name in ("v.nii", "v.nii.gz"):
        write_nifti(str(tmp_path / name), vol, A)
        got, B = read_nifti(str(tmp_path / name))
        assert got.dtype == vol.dtype and np.array_equal(got, vol) and np.allclose(B, A)
    ras, R = reorient_to_ras(got, B)
    assert ras.shape == (6, 5, 7)
    assert np.all(np.diag(R[:3, :3]) > 0) and np.allclose(R[:3, :3], np.diag(np.diag(R[:3, :3])))
    # the voxel that sat at world position w still does
    for ijk in ((0, 0, 0), (4, 5, 6), (2, 1, 3)):
        w = A @ np.array([*ijk, 1.0])
        j = np.linalg.solve(R, w)[:3].round().astype(int)
        assert ras[tuple(j)] == vol[ijk]
    with pytest.raises(ValueError):
        (tmp_path / "bad.nii").write_bytes(b"\0" * 400)
        read_nifti(str(tmp_path / "bad.nii"))


def test_nifti_scl_slope_inter_follow_nibabel(tmp_path):
    """scl_slope / scl_inter: NaN (or slope 0) means "no scaling", a NaN intercept counts as 0 - never a NaN volume"""
    import struct
    from mi_seg_amd.data.nifti import read_nifti, write_nifti
    vol = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    write_nifti(str(tmp_path / "v.nii"), vol)
    raw = bytearray((tmp_path / "v.nii").read_bytes())
    nan = float("nan")
    for slope, inter, want in ((nan, nan, vol), (nan, 0.0, vol), (0.0, 5.0, vol), (1.0, nan, vol), (1.0, 0.0, vol),
                               (2.0, nan, vol.astype(np.float32) * 2), (2.0, -1.0, vol.astype(np.float32) * 2 - 1), (1.0, 3.0, vol.astype(np.float32) + 3)):
        struct.pack_into("<2f", raw, 112, slope, inter)
        (tmp_path / "s.nii").write_bytes(bytes(raw))
        got, _ = read_nifti(str(tmp_path / "s.nii"))
        assert np.all(np.isfinite(got)) and np.array_equal(got, want), (slope, inter)


def test_checkpoint_import_export(tmp_path):
    from mi_seg_amd.data.checkpoint import export_state, load_model_state
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.networks.norms.utils import parse_normalization
    from mi_seg_amd.utils.detfill import fill_module_
    cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
    mk = lambda: SwinUNETR((64, 64, 64), 1, 3, feature_size=12, vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst)
    a, b, c = mk(), mk(), mk()
    fill_module_(a)
    export_state(a, str(tmp_path / "tune.pt"), epoch=7, best_acc=0.5)                       # tune.py:27-38 layout
    meta = load_model_state(b, str(tmp_path / "tune.pt"))
    assert meta["epoch"] == 7 and all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())
    export_state(a, str(tmp_path / "lit.ckpt"), lightning=True)                             # Lightning: "model." prefix
    raw = torch.load(str(tmp_path / "lit.ckpt"), weights_only=False)
    assert all(k.startswith("model.") for k in raw["state_dict"])
    raw["state_dict"] = {"module." + k: v for k, v in raw["state_dict"].items()}             # saved from a DDP wrapper on top
    load_model_state(c, raw)
    assert all(torch.equal(v, c.state_dict()[k]) for k, v in a.state_dict().items())
    del raw["state_dict"]["module.model.out.conv.conv.bias"]
    with pytest.raises(RuntimeError, match="Missing key"):
        load_model_state(mk(), raw)


def test_load_from_requires_the_reference_key_set():
    """SwinUNETR.load_from (reference swin_unetr.py:303-351 copies an explicit key list and raises KeyError on a missing one)"""
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    m = SwinUNETR((64, 64, 64), 1, 3, feature_size=12)
    good = {"state_dict": {"module." + k.replace("linear1", "fc1").replace("linear2", "fc2"): v.clone() for k, v in m.swinViT.state_dict().items()}}
    m.load_from(good)
    bad = {"state_dict": {k: v for k, v in good["state_dict"].items() if "layers2.0.blocks.1.mlp.fc2.weight" not in k}}
    with pytest.raises(KeyError):
        m.load_from(bad)


@pytest.mark.gpu
def test_gpu_augmentation_matches_torch_ops():
    """crop -> flips -> rot90 -> scale -> shift, image and label, against the same chain of torch ops (bit-identical: a gather and one fma)"""
    from mi_seg_amd.data.augment import GpuAugmenter, ResidentVolume
    from mi_seg_amd.data.synthetic import synthetic_volume
    img, lab = synthetic_volume((72, 80, 64), 3, 1)
    for ldt in (torch.int64, torch.uint8, torch.float32):
        vol = ResidentVolume(img[0].cuda(), lab[0, 0].to(ldt).cuda(), modality=1)
        aug = GpuAugmenter((48, 48, 32), patches_training_sample=6, randFlipd_prob=0.5, randRotate90d_prob=0.7, randScaleIntensityd_prob=0.5,
                           randShiftIntensityd_prob=0.5, seed=11)
        params = aug.draw(vol)
        out = aug(vol, params)
        assert out["image"].shape == (6, 1, 48, 48, 32) and out["label"].shape == (6, 1, 48, 48, 32) and out["modality"].tolist() == [1] * 6
        assert any(p["rot_k"] for p in params) and any(any(p["flip"]) for p in params)
        for i, p in enumerate(params):
            o = p["origin"]
            ci = vol.image[:, o[0]:o[0] + 48, o[1]:o[1] + 48, o[2]:o[2] + 32]
            cl = vol.label[o[0]:o[0] + 48, o[1]:o[1] + 48, o[2]:o[2] + 32][None]
            for ax in range(3):
                if p["flip"][ax]:
                    ci, cl = ci.flip(1 + ax), cl.flip(1 + ax)
            ci, cl = torch.rot90(ci, p["rot_k"], (1, 2)), torch.rot90(cl, p["rot_k"], (1, 2))
            want = ci * (1.0 + torch.tensor(p["scale"], dtype=torch.float32)) + torch.tensor(p["shift"], dtype=torch.float32)
            assert torch.equal(out["label"][i], cl)
            assert torch.allclose(out["image"][i], want, rtol=0, atol=1e-6)
    # pos / neg balance (RandCropByPosNegLabeld pos=1, neg=1, data/multi_modal.py:50-59): the share of crop centres that are foreground
    # voxels follows pos / (pos + neg) although the foreground is a small part of the volume; every centre comes from the right voxel list
    vol = ResidentVolume(img[0].cuda(), lab[0, 0].cuda())
    lab_host = lab[0, 0]
    fg_share = float((lab_host > 0).float().mean())
    assert fg_share < 0.35
    for pos, neg, want in ((1.0, 1.0, 0.5), (3.0, 1.0, 0.75)):
        aug = GpuAugmenter((32, 32, 32), patches_training_sample=16, pos=pos, neg=neg, seed=5)
        centres = [p["centre"] for _ in range(40) for p in aug.draw(vol)]
        hits = sum(int(lab_host[c] > 0) for c in centres)
        assert abs(hits / len(centres) - want) < 0.07, (pos, neg, hits / len(centres))            # 640 draws: sigma 0.02
        assert all(0 <= o and o + 32 <= s for p in aug.draw(vol) for o, s in zip(p["origin"], lab_host.shape))
    with pytest.raises(ValueError):
        GpuAugmenter((96, 96, 96))(vol)                       # volume smaller than the roi


@pytest.mark.gpu
def test_resample_and_the_cached_head_of_the_chain(tmp_path):
    """Spacingd-style resampling kernel against torch's interpolate (voxel centres aligned), then the whole deterministic head from two
    NIfTI files to a ResidentVolume that the augmenter and the sliding-window inferer consume."""
    import torch.nn.functional as F
    from mi_seg_amd.data.nifti import write_nifti
    from mi_seg_amd.data.preprocess import load_resident_volume, resample, scale_intensity, spatial_pad
    from mi_seg_amd.data.synthetic import synthetic_volume
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 20, 24, 18, generator=g).cuda()
    for size in ((33, 24, 40), (10, 30, 18), (20, 24, 18)):
        got = resample(x, size, "trilinear")
        want = F.interpolate(x[None], size=size, mode="trilinear", align_corners=False)[0]
        assert torch.allclose(got, want, atol=2e-6, rtol=0), size
        lab = torch.randint(0, 6, (1, 20, 24, 18), generator=g).cuda()
        got_l = resample(lab, size, "nearest")
        want_l = F.interpolate(lab[None].float(), size=size, mode="nearest-exact")[0].long()
        assert torch.equal(got_l, want_l), size
    assert float(scale_intensity(x).amin()) == 0.0 and float(scale_intensity(x).amax()) == 1.0
    assert spatial_pad(x, (24, 24, 24)).shape == (2, 24, 24, 24) and torch.equal(spatial_pad(x, (24, 24, 24))[:, 2:22, :, 3:21], x)
    # files -> resident volume: a 2 mm scan of 40 x 44 x 36 voxels with permuted / flipped axes becomes a RAS 1 mm volume of 80 x 88 x 72
    img, lab = synthetic_volume((40, 44, 36), 4, 1)
    A = np.array([[0, -2.0, 0, 10], [2.0, 0, 0, -5], [0, 0, 2.0, 3], [0, 0, 0, 1.0]])
    write_nifti(str(tmp_path / "mr_1.nii.gz"), img[0, 0].numpy(), A)
    write_nifti(str(tmp_path / "mr_1_label.nii.gz"), lab[0, 0].numpy().astype(np.uint8), A)
    vol = load_resident_volume({"image": str(tmp_path / "mr_1.nii.gz"), "label": str(tmp_path / "mr_1_label.nii.gz"), "modality": "MR"}, roi=(96, 96, 64))
    assert vol.modality == 1 and vol.image.shape == (1, 96, 96, 72) and vol.label.shape == (96, 96, 72)
    assert 0.0 <= float(vol.image.amin()) and float(vol.image.amax()) <= 1.0 and set(vol.label.unique().tolist()) <= set(range(6))
    frac_fg = float((lab > 0).float().mean())
    assert abs(float((vol.label[8:88, 4:92] > 0).float().mean()) - frac_fg) < 0.02      # the label map survived reorientation + resampling
    from mi_seg_amd.data.augment import GpuAugmenter
    batch = GpuAugmenter((64, 64, 64), patches_training_sample=2, seed=1)(vol)
    assert batch["image"].shape == (2, 1, 64, 64, 64) and batch["modality"].tolist() == [1, 1]
