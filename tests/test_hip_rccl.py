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


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("flags", ["device", "host"])
def test_two_rank_step_on_one_card(flags):
    """the N > 1 code with N > 1 PROCESSES where the driver's `-m gpu` run sees it.  Two ranks of `bench.py --gpus 2`
    share this box's one card (MISEG_REHEARSE_ONE_GPU=1; gloo between them - RCCL wants a device per rank), started as fresh children by a
    launcher that never touches the GPU (bench.py::self_launch: torch.distributed.run as a child process, no exec of a process that has
    initialised HIP): the real split step (two hipGraphs, the early range exchanged between them, the bitmap on the device or through the
    host), sharded sampler, exchange_check (the arena after a data-parallel step against the plain mean of the ranks' local gradients),
    and "unused on every rank => flagged unused / grad is None" for the conditional-norm rows of a modality no rank drew.
    Reference: tune.py:103-109 (DDP, find_unused_parameters), data/multi_modal.py:282-292 (DistributedSampler)."""
    env = dict(os.environ)
    env.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "MISEG_REHEARSE_ONE_GPU": "1", "MISEG_DIST_BACKEND": "gloo"})
    env.pop("MISEG_HIP_LIB", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-roofline", "--no-secondary"]
    if flags == "host":
        cmd.append("--host-flags")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=880)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, tail
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 and out["collective"]["ranks"] == 2 and out["collective"]["backend"] == "gloo"
    assert out["collective"]["launched_per_step"] >= 2, out["collective"]              # the early range and the rest (+ the bitmap)
    assert out["exchange_check"]["grad_rel_err_vs_mean_of_ranks"] <= 1e-6, out["exchange_check"]
    gu = out["exchange_check"]["globally_unused"]
    assert isinstance(gu, dict), gu
    assert gu["as_expected"] and gu["params_flagged_unused"] == 62, gu                 # the 31 conditional norms' rows of the absent modality
    assert gu["flags"].startswith(flags)
    if flags == "host":
        assert gu["grad_is_none_exactly_for_them"], gu
    chk = out["replay_check"]
    assert chk["finite"] and chk["logits_rel_err"] <= 1e-2 and not chk["params_without_grad_unexpected"] and not chk["params_with_zero_grad"], chk


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
