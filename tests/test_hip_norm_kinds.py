"""GPU tests of the smaller API surface the reference's hot path accepts ("what's missing" 2-4): the `batch` / `group` norm
kinds of networks/norms/utils.py:11-14 (factories.py:219-257), multi-channel images in the res-block stem (--in_channels, dynunet_block.py:82-98)
and the UNet activations beyond PReLU (--activation, networks/nets/unet.py:116-134) - each against plain torch fp32 on the same device."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cl(t):      # NCDHW -> channels-last rows tensor
    return t.permute(0, 2, 3, 4, 1).contiguous()


@pytest.mark.parametrize("kind", ["group", "batch"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_group_and_batch_norm_match_torch(kind, dtype, tol):
    from mi_seg_amd.networks.layers.utils import apply_norm, get_norm_layer
    from mi_seg_amd.networks.norms.utils import parse_normalization
    torch.manual_seed(3)
    B, C, S = 2, 24, (6, 10, 8)
    spec = parse_normalization(kind, True, 4, 2)
    m = get_norm_layer(spec, 3, C).to(DEV)
    assert isinstance(m, nn.GroupNorm if kind == "group" else nn.BatchNorm3d)
    ref = get_norm_layer(spec, 3, C).to(DEV)
    with torch.no_grad():
        m.weight.copy_(torch.rand(C) + 0.5)
        m.bias.copy_(torch.randn(C) * 0.2)
    ref.load_state_dict(m.state_dict())
    x = (torch.randn(B, C, *S, device=DEV) * 1.5 + 0.7)
    g = torch.randn(B, C, *S, device=DEV)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(g)
    xh = _cl(x).to(dtype).requires_grad_(True)
    y = apply_norm(m, xh, None)
    y.backward(_cl(g).to(dtype))
    assert rel_err(y.float().permute(0, 4, 1, 2, 3), yr) < tol
    assert rel_err(xh.grad.float().permute(0, 4, 1, 2, 3), xr.grad) < tol
    assert rel_err(m.weight.grad, ref.weight.grad) < tol and rel_err(m.bias.grad, ref.bias.grad) < tol
    if kind == "batch":
        assert rel_err(m.running_mean, ref.running_mean) < 1e-4 + tol and rel_err(m.running_var, ref.running_var) < 1e-4 + tol
        assert int(m.num_batches_tracked) == 1
        m.eval(), ref.eval()
        with torch.no_grad():
            assert rel_err(apply_norm(m, _cl(x).to(dtype), None).float().permute(0, 4, 1, 2, 3), ref(x)) < tol + 1e-3
        with pytest.raises(NotImplementedError):
            apply_norm(m, _cl(x).to(dtype).requires_grad_(True), None)


@pytest.mark.parametrize("kind,cin,image", [("group", 8, 0), ("batch", 8, 0), ("instance", 0, 2), ("instance", 0, 5), ("group", 0, 3)])
def test_res_block_with_other_norm_kinds_and_multi_channel_images(kind, cin, image):
    """UnetResBlock (dynunet_block.py:26-126) with the batch / group kinds and with a 2- / 3- / 5-channel image as its input, against the same
    arithmetic in torch ops"""
    from mi_seg_amd.networks.blocks.dynunet_block import UnetResBlock
    from mi_seg_amd.networks.norms.utils import parse_normalization
    from mi_seg_amd.utils.detfill import fill_module_
    torch.manual_seed(5)
    cout, S = 16, (8, 12, 8)
    c_in = image or cin
    blk = UnetResBlock(3, c_in, cout, 3, 1, parse_normalization(kind, True, 4, 2))
    fill_module_(blk)
    blk = blk.to(DEV)
    x = torch.randn(2, c_in, *S, device=DEV)
    g = torch.randn(2, cout, *S, device=DEV)

    def tnorm(m, t):
        if isinstance(m, nn.GroupNorm):
            return F.group_norm(t, m.num_groups, m.weight, m.bias, m.eps)
        if isinstance(m, nn.BatchNorm3d):
            return F.batch_norm(t, None, None, m.weight, m.bias, True, 0.1, m.eps)
        return F.instance_norm(t, weight=m.weight, bias=m.bias, eps=m.eps)

    xr = x.clone().requires_grad_(image == 0)
    o = F.leaky_relu(tnorm(blk.norm1, F.conv3d(xr, blk.conv1.conv.weight, padding=1)), 0.01)
    o = tnorm(blk.norm2, F.conv3d(o, blk.conv2.conv.weight, padding=1))
    o = F.leaky_relu(o + tnorm(blk.norm3, F.conv3d(xr, blk.conv3.conv.weight)), 0.01)
    o.backward(g)
    want = {k: p.grad.clone() for k, p in blk.named_parameters()}
    for p in blk.parameters():
        p.grad = None
    if image:
        y = blk(None, None, image=x, dtype=torch.float32)
    else:
        xh = _cl(x).requires_grad_(True)
        y = blk(xh, None)
    y.backward(_cl(g))
    assert rel_err(y.permute(0, 4, 1, 2, 3), o) < 1e-4
    if not image:
        assert rel_err(xh.grad.permute(0, 4, 1, 2, 3), xr.grad) < 1e-3
    for k, p in blk.named_parameters():
        if "running" in k:
            continue
        assert rel_err(p.grad, want[k]) < 2e-3, k


@pytest.mark.parametrize("act", ["relu", "leakyrelu", "gelu", ("leakyrelu", {"negative_slope": 0.2})])
def test_unet_activations_beyond_prelu(act):
    """--activation of the reference's parser reaches MONAI's Act factory (networks/nets/unet.py:116-134): relu / leakyrelu / gelu on the HIP path"""
    from mi_seg_amd.networks.nets.unet import UNet
    from mi_seg_amd.networks.norms.utils import parse_normalization
    from mi_seg_amd.utils.detfill import fill_module_
    inst = parse_normalization("instance", True, 4, 2)
    net = UNet(3, 1, 3, channels=(8, 16), strides=(2,), num_res_units=0, act=act, norm_down=inst, norm_up=parse_normalization("instance", True, 4, 2))
    fill_module_(net)
    net = net.to(DEV).set_compute_dtype(torch.float32)
    x = torch.randn(2, 1, 16, 16, 16, device=DEV)
    y = net(x)
    y.backward(torch.randn_like(y))
    assert bool(torch.isfinite(y).all()) and all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    # the activation itself against torch, through the same ADN path
    from mi_seg_amd.hip import functional as HF
    t = torch.randn(3, 4, 4, 4, 8, device=DEV).requires_grad_(True)
    kind = act if isinstance(act, str) else act[0]
    slope = 0.0 if kind == "relu" else (0.01 if isinstance(act, str) else act[1]["negative_slope"])
    got = HF.gelu(t) if kind == "gelu" else HF.leaky_relu(t, slope)
    want = F.gelu(t.detach()) if kind == "gelu" else F.leaky_relu(t.detach(), slope)
    assert rel_err(got, want) < 1e-6
    with pytest.raises(NotImplementedError):
        UNet(3, 1, 3, channels=(8, 16), strides=(2,), act="mish")
