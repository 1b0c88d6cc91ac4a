"""The RCCL leg of the data-parallel step on the GPU box (one rank): reference tune.py:103-109, 286-288 (DDP gradient all-reduce).

The collectives run in a FRESH child process (tests/rccl_child.py) that initialises a one-rank `nccl` process group before its first GPU
call, with MISEG_FORCE_COLLECTIVE=1 so that runtime/arena.py launches every exchange although one rank has nothing to exchange."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_exchange_one_rank():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(29600 + os.getpid() % 300)
    env.pop("MISEG_HIP_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py")], env=env, capture_output=True, text=True, timeout=580)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    lines = [l for l in r.stdout.splitlines() if l.startswith("RCCL_ONE_RANK_OK")]
    assert lines, tail
    out = json.loads(lines[-1].split(" ", 1)[1])
    # every collective of the N > 1 step was really launched: 4 buckets + bitmap, and bitmap + tail + head for the overlapped exchange
    # ... and the step with the collectives recorded into its hipGraph replays what the one-rank graph computes (5 captured collectives: the bitmap and four ranges)
    assert out == {"allreduce_collectives": 5, "overlapped_collectives": 3, "bf16_collectives": 3, "captured_collectives_per_step": 5}, out


def test_force_collective_switch_is_off_by_default(monkeypatch):
    """CPU: a one-rank group skips its collectives unless MISEG_FORCE_COLLECTIVE / force_collective asks for them"""
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    from mi_seg_amd.runtime.arena import ParamArena
    p = [torch.nn.Parameter(torch.zeros(8))]
    monkeypatch.delenv("MISEG_FORCE_COLLECTIVE", raising=False)
    a = ParamArena(p, torch.float32)
    assert not a.force_collective and a.collectives_launched == 0
    a.detach()
    monkeypatch.setenv("MISEG_FORCE_COLLECTIVE", "1")
    a = ParamArena(p, torch.float32)
    assert a.force_collective
    a.detach()
    a = ParamArena(p, torch.float32, force_collective=False)
    assert not a.force_collective
    a.detach()


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["forward", "step"])
def test_graph_captured_as_first_gpu_work_replays_like_eager(mode):
    """regression for the statistics-pool bug of round 3 (a chunk recycled short under capture: replays >= 1 normalised with accumulated sums):
    tests/pool_child.py captures the headline net's graphs as the first GPU work of a fresh process and compares replays 0-3 with eager"""
    env = dict(os.environ)
    env.pop("MISEG_HIP_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pool_child.py"), mode], env=env, capture_output=True, text=True, timeout=880)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0 and "POOL_CHILD_OK" in r.stdout, tail
