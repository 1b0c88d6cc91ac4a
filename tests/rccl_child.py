"""Child process of tests/test_hip_rccl.py (not collected by pytest): the RCCL leg of the data-parallel step on ONE rank.

A one-rank `nccl` (= RCCL) process group is initialised before this process makes any GPU call; `MISEG_FORCE_COLLECTIVE=1` disables the
one-rank short-circuits of runtime/arena.py, so every collective of the N > 1 step is really launched: the bucketed mean all-reduce
(`allreduce`), the overlapped exchange (`allreduce_begin` / `allreduce_end` with the "used on any rank" bitmap left on the device), fp32 and
bf16 buckets, on the gradient arena of the headline net (C-Swin-UNETR fs=48, 96^3) after a real forward + backward.  With one rank the mean
of the ranks is the local sum: the arena must come back bit-identical (fp32 buckets) / rounded once to bf16 (bf16 buckets), `used_dev` must
equal the local flags, `p.grad` must alias the arena slots.  Reference: tune.py:103-109, 286-288 (DDP gradient all-reduce).

Prints RCCL_ONE_RANK_OK and a JSON summary on success; any failed check raises."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
os.environ["MISEG_FORCE_COLLECTIVE"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# ProcessGroupNCCL recycles its works' events through a cache; an event that a CAPTURED collective recorded last comes back to an eager one,
# and the watchdog thread's poll of it ("operation not permitted on an event last recorded in a capturing stream") aborts the process:
# 1 run in 8 - 10 here with the cache, 0 in 24 without (round 4)
os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def stage(name):
    print("STAGE " + name, flush=True)


def main():
    dist.init_process_group("nccl", rank=0, world_size=1)          # before any GPU call of this process
    torch.cuda.set_device(0)
    import __graft_entry__ as ge
    ge.load_package()
    from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
    from mi_seg_amd.networks.norms.utils import parse_normalization
    from mi_seg_amd.runtime.arena import ParamArena
    from mi_seg_amd.utils.detfill import det_input, fill_module_
    cond, inst = parse_normalization("instance_cond", True, 4, 2), parse_normalization("instance", True, 4, 2)
    net = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond, decoder_norm_name=inst).cuda()
    fill_module_(net)
    net.set_compute_dtype(torch.bfloat16)
    x = det_input(3, (1, 1, 96, 96, 96)).cuda()
    cot = det_input(4, (1, 6, 96, 96, 96)).cuda()
    params = [p for p in net.parameters() if p.requires_grad]
    out = {}

    def local_step(arena, modality):
        for _ in range(2):                      # the first step registers the weight re-layouts
            arena.begin_step()
            net(x, [modality]).backward(cot)
            arena.end_backward()
        torch.cuda.synchronize()
        return arena.flat.clone(), [bool(p._miseg_used) for p in arena.params]

    def check_grads(arena, flags):
        for p, v, u in zip(arena.params, arena.views, flags):
            if u:
                assert p.grad is not None and p.grad.data_ptr() == v.data_ptr(), "p.grad must alias the arena slot"
            else:
                assert p.grad is None, "a parameter unused on every rank keeps grad None"

    # ---- fp32 buckets -----------------------------------------------------------------------------------------------------------------
    stage("fp32 buckets")
    arena = ParamArena(params, torch.bfloat16)
    assert arena.force_collective
    local, flags = local_step(arena, 0)
    assert 0 < sum(flags) < len(flags)          # the absent modality's conditional-norm rows are unused
    assert float(local.abs().max()) > 0 and bool(torch.isfinite(local).all())
    n0 = arena.collectives_launched
    arena.allreduce(1)                          # the 4-bucket exchange after the whole backward pass (bench.py --no-overlap)
    torch.cuda.synchronize()
    out["allreduce_collectives"] = arena.collectives_launched - n0
    assert out["allreduce_collectives"] == len(arena.buckets) + 1, out      # every bucket + the bitmap
    assert torch.equal(arena.flat, local), "mean over one rank must return the local sums bit for bit"
    check_grads(arena, flags)

    # the overlapped exchange as bench.py issues it at N > 1: bitmap on the device, the decoder-side tail first, the rest after
    local, flags = local_step(arena, 1)
    tail = arena.tail_offset(net.late_backward_parameters())
    n0 = arena.collectives_launched
    ub = arena.used_begin(host=False)
    works = arena.allreduce_begin(tail, arena.flat.numel())
    assert len(works) == 1
    arena.allreduce_end(works, 1, rest=(0, tail), used_work=ub, host_flags=False)
    torch.cuda.synchronize()
    out["overlapped_collectives"] = arena.collectives_launched - n0
    assert out["overlapped_collectives"] == 3, out          # bitmap, tail, head
    assert torch.equal(arena.flat, local)
    assert arena.used_on_device and arena.used_dev.tolist() == [int(f) for f in flags], "used_dev must hold this rank's flags"
    check_grads(arena, flags)
    # ... and with the host read of the bitmap (what a torch optimiser needs)
    local, flags = local_step(arena, 0)
    works = arena.allreduce_begin(tail, arena.flat.numel(), piece=(arena.flat.numel() - tail) // 3 + 1)
    assert len(works) == 3
    arena.allreduce_end(works, 1, rest=(0, tail))
    torch.cuda.synchronize()
    assert torch.equal(arena.flat, local) and not arena.used_on_device
    check_grads(arena, flags)
    arena.detach()

    # ---- bf16 buckets (bench.py --grad-dtype bf16): rounded into the staging buffer, averaged there, written back ------------------------
    stage("bf16 buckets")
    arena16 = ParamArena(params, torch.bfloat16, grad_dtype=torch.bfloat16)
    local, flags = local_step(arena16, 1)
    n0 = arena16.collectives_launched
    ub = arena16.used_begin(host=False)
    works = arena16.allreduce_begin(tail, arena16.flat.numel())
    arena16.allreduce_end(works, 1, rest=(0, tail), used_work=ub, host_flags=False)
    torch.cuda.synchronize()
    out["bf16_collectives"] = arena16.collectives_launched - n0
    assert out["bf16_collectives"] == 3
    assert torch.equal(arena16.flat, local.to(torch.bfloat16).float()), "bf16 buckets: one rounding of the local sums"
    assert arena16.used_dev.tolist() == [int(f) for f in flags]
    check_grads(arena16, flags)
    local, flags = local_step(arena16, 0)
    arena16.allreduce(1)
    torch.cuda.synchronize()
    assert torch.equal(arena16.flat, local.to(torch.bfloat16).float())
    check_grads(arena16, flags)
    arena16.detach()

    # ---- the data-parallel step with its collectives recorded INTO the step's hipGraph (bench.py --captured-collective, hook mode) ---------
    from mi_seg_amd.runtime.graph import GraphedStep
    torch.cuda.synchronize()
    time.sleep(0.3)          # ProcessGroupNCCL's watchdog (100 ms poll) reaps the eager works above before the first capture starts
    stage("one-rank graph")
    arena = ParamArena(params, torch.bfloat16)
    single = GraphedStep(net, x.shape, cot.shape, arena=arena)
    ref = {}
    for m in (0, 1):
        y = single(x, [m], cot)
        torch.cuda.synchronize()
        ref[m] = (y.detach().clone(), arena.flat.clone(), [p.grad is None for p in params])
    n = arena.flat.numel()
    hole = arena.param_range(net.deferred_backward_parameters())
    early_ranges, late_ranges = [(hole[1], n), (tail, hole[0])], [(0, tail), hole]

    class Comm:
        captured = 0

        @staticmethod
        def early():
            works = []
            for lo, hi in early_ranges:
                works.extend(arena.allreduce_begin(lo, hi))
            return works

        @staticmethod
        def late(works):
            for lo, hi in late_ranges:
                works.extend(arena.allreduce_begin(lo, hi))
            ub = arena.used_begin(host=False)
            for w in works:
                w.wait()
            ub()
            arena._unstage()
            Comm.captured = len(works) + 1

    stage("captured step: warm-up + first capture")
    cap = GraphedStep(net, x.shape, cot.shape, arena=arena, fused_comm=Comm)
    for it, m in enumerate([1, 0, 0, 1]):
        stage(f"captured step: replay {it}")
        y = cap(x, [m], cot, publish=False)                  # bitmap + 4 ranges: all nodes of the graph, no eager collective
        arena.allreduce_finish(lambda: None, 1)
        torch.cuda.synchronize()
        y_ref, g_ref, none_ref = ref[m]
        assert torch.equal(y.detach(), y_ref), f"captured step, replay {it}: logits differ from the one-rank graph"
        assert [p.grad is None for p in params] == none_ref
        err = float((arena.flat - g_ref).norm() / g_ref.norm())
        assert err < 1e-4, f"captured step, replay {it}: arena differs from the one-rank graph by {err:.2e}"
        assert arena.used_dev.tolist() == [int(not v) for v in none_ref]
    assert Comm.captured == 5, Comm.captured
    out["captured_collectives_per_step"] = Comm.captured
    arena.detach()

    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
