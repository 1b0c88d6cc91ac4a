import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402

ge.load_package()
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must not silently run on a CPU-only box
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(bytes(self.z["__meta__"]).decode())

    def t(self, key, dtype=torch.float32):
        return torch.from_numpy(np.asarray(self.z[key])).to(dtype)

    def has(self, key):
        return key in self.z.files

    def grads(self, tag):
        pre = f"{tag}/grad:"
        return {k[len(pre):]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith(pre)}

    def grads2(self, tag):
        pre = f"{tag}/grad2:"
        return {k[len(pre):]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith(pre)}

    def gnorms(self, tag):
        pre = f"{tag}/gnorm:"
        return {k[len(pre):]: float(self.z[k]) for k in self.z.files if k.startswith(pre)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def sample(t, n=4096):
    """Same deterministic strided sample as oracle/tools/make_golden.py."""
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.float().cpu()
    idx = torch.linspace(0, f.numel() - 1, n).round().long()
    return f[idx.to(f.device)].float().cpu()


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def state_from_meta(case, requires_grad=True):
    """Rebuild a reference-layout state_dict from fixture metadata with the deterministic fill."""
    from mi_seg_amd.utils.detfill import det_values
    from oracle.functional import relative_position_index
    sd = {}
    for k, shp in zip(case["state_keys"], case["state_shapes"]):
        if k.endswith("relative_position_index"):
            sd[k] = relative_position_index()
        else:
            sd[k] = torch.from_numpy(det_values(k, shp)).requires_grad_(requires_grad)
    return sd


def compare_grads(got_named, want, tol, sampled=False, vanish_tol=1e-2, skip=()):
    """Per-parameter relative L2 check of gradients against golden vectors: no pooling, no escape for small errors.

    Parameters whose TRUE gradient vanishes (a bias / 1-channel conv in front of an instance norm: the norm removes
    any per-channel constant or scale) hold pure rounding noise in the golden vectors; they are recognised by a
    per-element RMS below 1e-3 of the median over all parameters and only required to be (numerically) zero too.
    Returns the worst (name, error)."""
    rms = {k: float(g.double().norm()) / max(1, g.numel()) ** 0.5 for k, g in want.items()}
    med = sorted(rms.values())[len(rms) // 2]
    worst = ("", 0.0)
    scal = scalar_scale(want)
    for k, g in want.items():
        if k in skip:
            continue
        got = got_named[k]
        assert got is not None, f"{k}: gradient missing"
        got = sample(got) if sampled else got.detach().float().cpu().reshape(g.shape)
        if rms[k] < 1e-3 * med:
            got_rms = float(got.double().norm()) / max(1, got.numel()) ** 0.5
            assert got_rms < vanish_tol * med, (k, "should vanish", got_rms, med)
            continue
        e = rel_err(got, g)
        if g.numel() == 1:
            e = min(e, float((got.double() - g.double()).abs().max()) / scal)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < tol, worst
    return worst


def scalar_scale(want):
    """One-element parameters (the UNet's PReLU slopes) are sums over every voxel of a layer with heavy cancellation: d slope = sum dy * min(x, 0)
    has terms adding up to ~600 in absolute value where the result is -1.49 (C1, level 3), so a 1.5e-3 relative change of dy - ONE
    activation-sign flip somewhere behind it, see SMALL_NET_BAR in test_hip_modules.py - moves it by 60 % while the other slope gradients of
    the same net (|g| ~ 100 - 1000) move by 1e-3.  Such a parameter is judged on the scale of its peers: error / max(|g|, median |g| over the
    one-element parameters).  Returns that median (1.0 if the net has none)."""
    vals = sorted(float(g.double().abs().max()) for g in want.values() if g.numel() == 1)
    return max(vals[len(vals) // 2], 1e-30) if vals else 1.0


def evidence(line):
    """print a measured-parity line and, when MISEG_EVIDENCE_FILE is set, append it to that file (scripts/profile_round.sh collects the
    `vs truth` lines of the final build into profiles/rNN_vs_truth.txt: pytest -q swallows prints of passing tests)"""
    print(line)
    path = os.environ.get("MISEG_EVIDENCE_FILE")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")
