#!/usr/bin/env python3
"""bench.py -- 96^3 patches/s, forward+backward, C-Swin-UNETR fs=48 / 6 classes (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
N > 1 runs one rank per GPU.  Under an external launcher (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
every rank reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; invoked bare, this process starts the N ranks itself as
child processes through torch.distributed.run - before it has made any GPU call - and exits with the launcher's code (non-zero unless all
N ranks came up and finished).

One step = forward + backward of one 96^3 patch per rank (batch 1/rank, synthetic CT/MR volume already resident in
HBM, cotangent = d(sum of logits * fixed noise)); at N > 1 the fp32 gradient arena is mean-all-reduced over RCCL
inside the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402


WORKLOADS = {
    "c2": ("96^3 patches/sec fwd+bwd, C-Swin-UNETR fs=48 6-class", "patches/s",
           "configs[1]: C-Swin-UNETR feature_size=48 heads=(3,6,12,24) instance_cond, 96^3 patch, 6 classes, batch 1 per GPU, fwd+bwd "
           "(+ gradient all-reduce at N>1)"),
    "c3": ("96^3 patches/sec fwd+bwd, C-UNETR (ViT-B/16 encoder, instance_cond) 6-class", "patches/s",
           "configs[2]: C-UNETR hidden 768 mlp 3072 heads 12 layers 12 feature_size=16 perceptron, 96^3 patch, 6 classes, batch 1 per GPU, fwd+bwd"),
    "c5": ("96^3 windows/sec, sliding-window inference of a 512x512x363 volume (overlap 0.5, 700 windows), C-Swin-UNETR fs=48 6-class", "windows/s",
           "configs[4]: whole 512x512x363 CT volume resident in HBM, roi 96^3, overlap 0.5 -> 700 windows in batches of 20 (modality broadcast), "
           "hipGraph forward, all window logits resident (14.9 GB), one gather-stitch pass; a step = one volume"),
}


def build_model(dtype, workload="c2"):
    from mi_seg_amd.networks.norms.utils import parse_normalization
    from mi_seg_amd.utils.detfill import fill_module_
    cond = parse_normalization("instance_cond", True, 4, 2)
    inst = parse_normalization("instance", True, 4, 2)
    if workload == "c3":
        from mi_seg_amd.networks.nets.unetr import UNETR
        m = UNETR(1, 6, (96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12, pos_embed="perceptron", vit_norm_name=cond,
                  encoder_norm_name=cond, decoder_norm_name=inst)
    else:
        from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
        m = SwinUNETR((96, 96, 96), 1, 6, feature_size=48, num_heads=(3, 6, 12, 24), vit_norm_name=cond, encoder_norm_name=cond,
                      decoder_norm_name=inst)
    fill_module_(m)
    return m.cuda().set_compute_dtype(dtype)


def synthetic_pool(n, seed, device):
    """n volumes in [0,1] (what ScaleIntensityd yields): smooth noise + ellipsoids; first half CT (0), second half MR (1)."""
    g = torch.Generator().manual_seed(seed)
    vols, mods = [], []
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, 96)] * 3, indexing="ij")
    for i in range(n):
        v = torch.nn.functional.interpolate(torch.rand(1, 1, 12, 12, 12, generator=g), size=96, mode="trilinear")[0, 0]
        for _ in range(5):
            c = torch.rand(3, generator=g) * 1.2 - 0.6
            r = torch.rand(3, generator=g) * 0.3 + 0.1
            v = v + 0.5 * ((((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2) < 1).float()
        v = (v - v.min()) / (v.max() - v.min())
        mod = 0 if i < n // 2 else 1
        if mod == 1:
            v = v ** 0.6          # different intensity transfer curve for "MR"
        vols.append(v)
        mods.append(mod)
    return torch.stack(vols).unsqueeze(1).to(device), mods


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (this process has not touched the GPU and never
    will: counting devices does not initialise HIP), one per GPU over RCCL, rendezvous on 127.0.0.1; returns the launcher's exit code -
    torch.distributed.run fails the whole job when any rank fails to start or exits non-zero, so fewer than N ranks never print a line"""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("MISEG_REHEARSE_ONE_GPU"):
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's intra-node transport on this driver
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS), help="c2 = BASELINE configs[1] (the metric); c3 / c5 = configs[2] / configs[4]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write the roofline leg's per-launch table (JSON) here")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-arena", action="store_true", help="torch-style per-parameter gradient tensors and per-call weight casts")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: all-reduce after the whole backward pass instead of overlapping the encoder half")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"],
                    help="N > 1: dtype of the gradient buckets on the wire (bf16 halves the xGMI bytes; the reference's DDP averages fp32: default)")
    ap.add_argument("--host-flags", dest="device_flags", action="store_false",
                    help="N > 1: read the exchanged 'used on any rank' bitmap back to the host every step (what a torch optimiser needs); default: it stays "
                         "on the device for the fused optimiser and the step has no host synchronisation")
    ap.add_argument("--captured-mode", default="hook", choices=["hook", "cut"],
                    help="--captured-collective: 'hook' = the backward pass stays whole, the model calls back when its decoder side is done; 'cut' = the two "
                         "halves of the split step recorded into one graph")
    ap.add_argument("--captured-collective", action="store_true",
                    help="N > 1 (or --force-dist): record the gradient all-reduces INTO the step's hipGraph (one graph per step, RCCL's nodes between the "
                         "two halves of the backward pass) instead of launching them between two graphs")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group (and take the N > 1 code path) even with one rank")
    ap.add_argument("--train-step", action="store_true",
                    help="c2 / c3, N = 1: time a whole OPTIMISATION step instead - versioned weight refresh + forward + fused DiceFocal loss + backward + "
                         "one-launch AdamW in ONE replayed hipGraph (reference lightning_monai.py:149-166, 255-278); reported under `secondary`")
    ap.add_argument("--refresh-weights", action="store_true",
                    help="bump the arena's device-side parameter version before every step, so that every replay re-casts and re-packs all weights as a "
                         "training loop's step does (the headline's timed steps have no optimiser step: their refresh launches find the copies current)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short configs[2] / configs[4] / fp32 runs reported under `secondary` (N = 1 only)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start one rank per GPU (or run `python bench.py --gpus {a.gpus}` bare)")
    if os.environ.get("MISEG_REHEARSE_ONE_GPU"):      # rehearsal of the N > 1 code path with every rank on the one card of a 1-GPU box
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ge.load_package()
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # (ProcessGroupNCCL's event cache hands an event last recorded by a CAPTURED collective to a later eager one; the watchdog thread's
        # poll of it aborts the process - tests/rccl_child.py: 1 run in 8 with the cache, 0 in 24 without)
        os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")
        dist.init_process_group(os.environ.get("MISEG_DIST_BACKEND", "nccl"), rank=rank, world_size=world)    # "nccl" is RCCL on ROCm
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but the process group has {dist.get_world_size()} ranks")
        dist.all_reduce(torch.zeros(8, device=dev))      # RCCL's communicator is built by the first collective: not inside a capture, not inside the timed region
        torch.cuda.synchronize()

    from mi_seg_amd.hip import lib as hiplib
    hiplib.check_device(local)
    out = measure(a, rank, world, dist, dev)
    if rank == 0:
        if world == 1 and dist is None and not a.no_secondary and a.workload == "c2" and a.dtype == "bf16":
            # driver-timed figures of the other configurations ("builder-run only"): short runs of the same harness, AFTER the
            # headline measurement and outside its timed region; the headline line and its timed region are unchanged
            import copy
            out["secondary"] = {}
            for name, over in (("c2_train_step_bf16", dict(train_step=True, steps=10, warmup=3)), ("c2_with_weight_refresh", dict(refresh_weights=True, steps=10, warmup=3)),
                               ("c2_fp32_parity_mode", dict(dtype="f32", steps=5, warmup=2)), ("c3_c_unetr_bf16", dict(workload="c3", steps=10, warmup=3)),
                               ("c5_sliding_window_bf16", dict(workload="c5", steps=2, warmup=2))):
                b = copy.copy(a)
                b.no_roofline, b.no_cpu_baseline = True, True
                for k, v in over.items():
                    setattr(b, k, v)
                r = measure(b, rank, world, dist, dev)
                out["secondary"][name] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "train_step", "weight_refresh") if k in r}
                out["secondary"][name]["workload"] = r["config"]["workload"]
                if "sw_batch_size" in r["config"]:      # (round 4 moved c5 from 4 to 20 windows per forward: 593 -> 643 windows/s of the 590 -> 641 between the rounds)
                    out["secondary"][name]["sw_batch_size"] = r["config"]["sw_batch_size"]
                torch.cuda.empty_cache()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measure(a, rank, world, dist, dev):
    """one workload of the harness: builds the model, times a.steps steps after a.warmup, returns the JSON line as a dict (rank 0 prints it)"""
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    model = build_model(dtype, a.workload)
    from mi_seg_amd.data.sampler import rank_indices
    from mi_seg_amd.hip import ops
    if a.workload == "c5":
        return bench_sliding_window(a, model, dtype, dev, rank, world, dist)
    # ONE global dataset = ConcatDataset([8 CT, 8 MR]) (identical on every rank), sharded like the reference's DistributedSampler
    # (data/multi_modal.py:282-292, tune.py:164): epoch-seeded permutation, rank r takes perm[r::world]; epochs follow one another
    pool, mods = synthetic_pool(16, 1000, dev)
    order, epoch = [], 0
    while len(order) < a.warmup + a.steps + 4:
        order += rank_indices(len(mods), world_size=world, rank=rank, epoch=epoch, seed=0)
        epoch += 1
    cot = torch.randn(1, 6, 96, 96, 96, generator=torch.Generator().manual_seed(4321)).to(dev)
    params = [p for p in model.parameters() if p.requires_grad]

    if getattr(a, "train_step", False):
        return bench_train_step(a, model, dtype, dev, pool, mods, order, params)
    arena = None
    if not a.no_arena:
        from mi_seg_amd.runtime.arena import ParamArena
        arena = ParamArena(params, dtype, grad_dtype=torch.bfloat16 if a.grad_dtype == "bf16" else torch.float32,
                           force_collective=True if (a.force_dist and os.environ.get("MISEG_FORCE_COLLECTIVE") != "0") else None)
    # N > 1: the backward pass is split behind the decoder side (autograd runs it first), whose gradients - 87 % of the bytes - are
    # all-reduced by RCCL while the encoder / Swin half still runs
    overlap = dist is not None and arena is not None and not a.no_overlap and hasattr(model, "late_backward_parameters")
    tail = arena.tail_offset(model.late_backward_parameters()) if overlap else None
    # decoder1's two 96^3 weight gradients leave the main stream's chain (ops.defer_to_branch).  MISEG_SPLIT_DEFER=early (default): they run on
    # the branch stream inside the FIRST half, beside decoder2 .. 5 / encoder10, and the whole tail is final at the hook.  =late (round 2):
    # they wait for the branch's backward pass in the second half - their slots are then a hole in the tail that goes out after the first
    # half, and join the ranges reduced at the end
    hole, hook_deep = None, None
    if overlap and hasattr(model, "deferred_backward_parameters") and getattr(model, "side_branch", False) and dtype == torch.bfloat16 and not os.environ.get("MISEG_NO_DEFER"):
        hook_mode = a.captured_collective and a.captured_mode == "hook" and not a.no_graph
        if hook_mode:
            # the backward pass is not cut: decoder1's two deferred weight gradients run at the head of the branch's backward pass as in the
            # one-rank step - after the callback - so their slots are a hole in the early range
            hole = arena.param_range(model.deferred_backward_parameters())
            assert tail <= hole[0] < hole[1] <= arena.flat.numel()
            if os.environ.get("MISEG_CAPTURED_EARLY", "all") == "deep":
                # (option) early range = encoder10 + decoder5 only: 70 % of the gradient bytes (175 of 249 MB), and exactly the layers whose
                # conv weight gradients are written INLINE by the tiny-volume kernel - at the callback only the small queued launches are
                # issued (GraphedStep: fused_comm.flush = "small"), the grouped conv weight gradients stay at the end of the pass.  One rank:
                # 147.6 patches/s without collectives (all: 145.2; one-rank graph 150.6) but 112 with RCCL's one-rank kernels forced (all: 140) -
                # with the collective forked that early the hipGraph executor serialises the three chains; not the default
                deep = arena.param_range([p for k, p in model.named_parameters() if p.requires_grad and k.startswith(("encoder10.", "decoder5."))])
                assert deep[0] == tail and deep[1] <= hole[0]
                hook_deep = deep
        elif os.environ.get("MISEG_SPLIT_EARLY", "all") == "deep":
            # (option, two-graph step) only encoder10 + decoder5 go out between the graphs - 70 % of the bytes, conv gradients written inline by
            # the tiny-volume kernel - so the first graph ends with the small queued launches only (GraphedStep first_flush="small")
            hook_deep = arena.param_range([p for k, p in model.named_parameters() if p.requires_grad and k.startswith(("encoder10.", "decoder5."))])
            assert hook_deep[0] == tail
            model.split_defers = True      # decoder1's deferred weight gradients with the branch's backward pass, in the second graph
        elif os.environ.get("MISEG_SPLIT_DEFER", "early") == "early":
            model.split_defers = "early"          # decoder1's deferred weight gradients run on the idle branch stream inside the first half: no hole
        else:
            hole = arena.param_range(model.deferred_backward_parameters())
            assert tail <= hole[0] < hole[1] <= arena.flat.numel()
            model.split_defers = True
    early_ranges = [(tail, arena.flat.numel())] if (overlap and hole is None) else ([(hole[1], arena.flat.numel()), (tail, hole[0])] if overlap else [])
    late_ranges = [(0, tail)] + ([hole] if hole is not None else []) if overlap else []
    if hook_deep is not None:
        early_ranges, late_ranges = [hook_deep], [(0, tail), (hook_deep[1], arena.flat.numel())]
    graphed = None
    if not a.no_graph:
        from mi_seg_amd.runtime.graph import GraphedStep
        fused = None
        if overlap and a.captured_collective:
            class _Comm:      # issued under capture: the bitmap, the early ranges between the halves, the late ones + the waits (RCCL's stream joins) at the end
                @staticmethod
                def early():
                    works = []
                    for lo, hi in early_ranges:
                        if hi > lo:
                            works.extend(arena.allreduce_begin(lo, hi))
                    return works

                @staticmethod
                def late(works):
                    for lo, hi in late_ranges:
                        if hi > lo:
                            works.extend(arena.allreduce_begin(lo, hi))
                    ub = arena.used_begin(host=False)      # this graph's "used" flags are known now: their exchange is a node of the graph too
                    for w in works:
                        w.wait()
                    ub()                                   # (the current stream waits for the bitmap exchange; the global flags stay in `used_dev`)
                    arena._unstage()
                    _Comm.captured = len(works) + 1
            _Comm.captured = 0
            _Comm.flush = "small" if hook_deep is not None else "all"
            fused = _Comm
        graphed = GraphedStep(model, (1, 1, 96, 96, 96), (1, 6, 96, 96, 96), arena=arena, split=overlap and not (fused is not None and a.captured_mode == "hook"),
                              fused_comm=fused, first_flush="small" if (hook_deep is not None and fused is None) else "all")
        graphed.cot.copy_(cot)
        cot = graphed.cot          # the cotangent is constant here: it lives in the graph's static buffer (a loss kernel would write it there)

    refresh = bool(getattr(a, "refresh_weights", False)) and arena is not None

    def step(i, eager=False, sample=None, comm=True):
        k = order[i % len(order)] if sample is None else sample
        if refresh:
            arena.invalidate()          # one counter_add launch: the step's refresh kernels then re-lay-out every weight (as after an optimiser step)
        if overlap and sample is None and comm:
            works, ub = [], []
            if graphed is not None and not eager and a.captured_collective:
                # the collectives are nodes of the step's graph: exchange the bitmap (known up front), replay, settle the flags
                graphed(pool[k:k + 1], [mods[k]], cot, publish=False)
                arena.allreduce_finish(lambda: None, world, None)      # every exchange was a node of the graph; the global flags are in `used_dev`
                return
            if graphed is not None and not eager:
                def between():          # the flags of a replayed graph are known up front: the bitmap exchange starts here as well
                    ub.append(arena.used_begin(host=not a.device_flags))
                    for lo, hi in early_ranges:
                        if hi > lo:
                            works.extend(arena.allreduce_begin(lo, hi))
                graphed(pool[k:k + 1], [mods[k]], cot, between=between, publish=False)
            else:
                arena.begin_step()
                cut = []
                model(pool[k:k + 1], [mods[k]], cut=cut).backward(cot)
                arena.flush()
                for lo, hi in early_ranges:
                    if hi > lo:
                        works.extend(arena.allreduce_begin(lo, hi))
                torch.autograd.backward([o for o, _ in cut], [l.grad for _, l in cut])
            arena.allreduce_end(works, world, rest=late_ranges, used_work=ub[0] if ub else None, host_flags=not a.device_flags)
            return
        if graphed is not None and not eager:
            graphed(pool[k:k + 1], [mods[k]], cot)
        elif arena is not None:
            arena.begin_step()
            model(pool[k:k + 1], [mods[k]]).backward(cot)
            arena.publish()
        else:
            ops.begin_step()
            for p in params:
                p.grad = None
            y = model(pool[k:k + 1], [mods[k]])
            y.backward(cot)
        if dist is not None and comm:
            if arena is not None:
                arena.allreduce(world)
            else:
                from mi_seg_amd.parallel.ddp import allreduce_gradients
                allreduce_gradients(params, world)

    # setup (untimed, not part of the W warm-up steps): touch both modalities once so that hipGraph capture / allocator growth of
    # either conditional-norm row set never lands inside the timed region
    for m_ in sorted(set(mods)):
        k_ = mods.index(m_)
        if graphed is not None:
            graphed(pool[k_:k_ + 1], [m_], cot)
        else:
            for _ in range(2):            # with an arena the first step registers the weight re-layouts, the second uses them
                step(0, eager=True, sample=k_)
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    coll0 = arena.collectives_launched if arena is not None else 0
    t0 = time.perf_counter()
    per_step = []
    for i in range(a.steps):
        ts = time.perf_counter()
        step(a.warmup + i)
        if os.environ.get("MISEG_BENCH_VERBOSE"):
            torch.cuda.synchronize()
            per_step.append(round(1e3 * (time.perf_counter() - ts), 2))
    torch.cuda.synchronize()
    if per_step and rank == 0:
        print("per-step ms (with a sync each):", per_step, file=sys.stderr)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    coll = (arena.collectives_launched - coll0) if arena is not None else 0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if os.environ.get("MISEG_BENCH_EARLY_PRINT") and rank == 0:      # experiments whose replay check may abort: the timing first
        print(f"early: {world * a.steps / dt:.2f} {WORKLOADS[a.workload][1]}", file=sys.stderr)
    if ops.STAMPS and rank == 0:          # MISEG_STEP_STAMPS=1: where the streams of the last replayed step were when (device clock, us)
        st = ops.read_stamps()
        t0_ = st.get("step_begin", min(st.values()))
        print("step stamps (us after step_begin): " + ", ".join(f"{k} {v - t0_:.0f}" for k, v in sorted(st.items(), key=lambda kv: kv[1])), file=sys.stderr)

    def replay_check():
        """untimed: the last replays of the captured step against the same step launched eagerly (logits and all gradients)"""
        k = order[(a.warmup + a.steps) % len(order)]
        flat = lambda: arena.flat.clone() if arena is not None else torch.cat([p.grad.reshape(-1).float() for p in params if p.grad is not None])
        y = graphed(pool[k:k + 1], [mods[k]], cot).detach().float().clone()
        g = flat()
        bg0 = ops.BACKGROUND_LAUNCHES
        if arena is not None:
            arena.begin_step()
            ye = model(pool[k:k + 1], [mods[k]])
            ye.backward(cot)
            arena.publish()
        else:
            ops.begin_step()
            for p in params:
                p.grad = None
            ye = model(pool[k:k + 1], [mods[k]])
            ye.backward(cot)
        ge_ = flat()
        rel = lambda u, v: float((u - v).norm() / v.norm())
        res = {"logits_rel_err": rel(y, ye.detach().float()), "grad_rel_err": rel(g, ge_), "finite": bool(torch.isfinite(g).all())}
        # no work skipped: every parameter received a gradient except the conditional-norm rows of the modality absent from this batch
        # (reference: grad is None for them), and none of the received ones is identically zero
        got = [(n, p.grad) for n, p in model.named_parameters() if p.requires_grad]
        missing = [n for n, g_ in got if g_ is None]
        other = 1 - int(mods[k])
        unexpected = [n for n in missing if f".norms.{other}." not in n]
        have = [(n, g_) for n, g_ in got if g_ is not None]       # (one multi-tensor launch and one read-back, not a launch pair and a sync per parameter)
        peaks = torch.stack(torch._foreach_norm([g_.float() for _, g_ in have], float("inf"))).tolist() if have else []
        zero = [n for (n, _), m_ in zip(have, peaks) if m_ == 0.0]
        res.update({"params": len(got), "params_without_grad": len(missing), "params_without_grad_unexpected": unexpected[:5], "params_with_zero_grad": zero[:5]})
        # launches per step that the model's side branch issues throttled (hip/ops.py::_background): conv3_fwd96 at one workgroup per CU,
        # conv3_wgrad on few CUs.  They run beside the main stream's launches in the step; the roofline leg times the kernels alone
        res["side_branch_background_launches_per_step"] = ops.BACKGROUND_LAUNCHES - bg0      # (the split step of N > 1 defers nothing: 2 fewer)
        # The timed steps have no optimiser step, so the versioned refresh of the weights' compute-dtype copies / conv packs (runtime/arena.py)
        # found them current and re-laid-out nothing.  Proof that a replay DOES pick up changed weights: scale every parameter a little
        # (in place), replay, and compare with the eager step on the changed weights - and with the replay before the change
        if arena is not None:
            with torch.no_grad():
                torch._foreach_mul_(list(params), 1.0 + 2.0 ** -7)      # (in place on the Parameters themselves: bumps Tensor._version, which the arena watches)
            y2 = graphed(pool[k:k + 1], [mods[k]], cot).detach().float().clone()
            g2 = flat()
            arena.begin_step()
            ye2 = model(pool[k:k + 1], [mods[k]])
            ye2.backward(cot)
            arena.publish()
            ge2 = flat()
            with torch.no_grad():
                torch._foreach_div_(list(params), 1.0 + 2.0 ** -7)
            res["after_weight_change"] = {"logits_rel_err_vs_eager": rel(y2, ye2.detach().float()), "grad_rel_err_vs_eager": rel(g2, ge2),
                                          "logits_moved_by": rel(y2, y), "grads_moved_by": rel(g2, g)}
            graphed(pool[k:k + 1], [mods[k]], cot)      # (and back: the restored weights are re-laid-out by this replay)
        return res

    def exchange_check():
        """untimed, N > 1: the arena after one (overlapped) data-parallel step against the plain mean of the ranks' local gradients"""
        k = order[(a.warmup + a.steps + 1) % len(order)]
        if graphed is not None:
            graphed(pool[k:k + 1], [mods[k]], cot)
        else:
            arena.begin_step()
            model(pool[k:k + 1], [mods[k]]).backward(cot)
            arena.publish()
        want = arena.flat.clone()
        dist.all_reduce(want)
        want /= world
        step(a.warmup + a.steps + 1)
        res = {"grad_rel_err_vs_mean_of_ranks": float((arena.flat - want).norm() / want.norm())}
        # "unused on EVERY rank => no gradient, no update" (torch DDP with find_unused_parameters, reference tune.py:105-109): find a step in
        # which all ranks drew the same modality, run it, and compare the exchanged flags with the conditional-norm rows of the other one
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        for i in range(a.warmup + a.steps + 2, a.warmup + a.steps + 2 + 12):
            mods_all = [None] * world
            dist.all_gather_object(mods_all, int(mods[order[i % len(order)]]))
            if len(set(mods_all)) != 1:
                continue
            step(i)
            torch.cuda.synchronize()
            flags = arena.used_dev.tolist() if arena.used_on_device else [int(bool(p._miseg_used)) for p in arena.params]
            unused = sorted(n for n, f in zip(names, flags) if not f)
            expected = sorted(n for n in names if f".norms.{1 - mods_all[0]}." in n)
            res["globally_unused"] = {"modality_of_every_rank": mods_all[0], "params_flagged_unused": len(unused), "as_expected": unused == expected,
                                      "flags": "device (used_dev)" if arena.used_on_device else "host"}
            if not arena.used_on_device:      # host flags: p.grad follows the GLOBAL flags
                named = dict(model.named_parameters())
                res["globally_unused"]["grad_is_none_exactly_for_them"] = all((named[n].grad is None) == (n in set(unused)) for n in names)
            break
        else:
            res["globally_unused"] = "not exercised: no step of the next 12 in which every rank draws the same modality"
        return res

    out = {
        "metric": WORKLOADS[a.workload][0], "value": world * a.steps / dt, "unit": WORKLOADS[a.workload][1],
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000 * dt / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic", "launch": "eager" if graphed is None else "hipGraph", "allreduce": "overlapped with the encoder half of backward" if overlap else ("after backward" if dist is not None else "none"),
        "config": {"workload": WORKLOADS[a.workload][2], "global_batch": world, "sharding": "one 16-volume CT+MR dataset, rank r takes perm[r::world] per epoch"},
    }
    if refresh:
        out["weight_refresh"] = "every step (device-side parameter version bumped before each replay: all casts and conv packs re-made)"
    if graphed is not None:
        check = replay_check()
        if not check["finite"] or check["logits_rel_err"] > 1e-2 or check["grad_rel_err"] > 5e-2:
            raise SystemExit(f"hipGraph replay does not reproduce the eager step: {check}")
        if check["params_without_grad_unexpected"] or check["params_with_zero_grad"]:
            raise SystemExit(f"the timed step left parameters without a gradient: {check}")
        awc = check.get("after_weight_change")
        if awc and (awc["logits_rel_err_vs_eager"] > 1e-2 or awc["grad_rel_err_vs_eager"] > 5e-2 or awc["logits_moved_by"] < 1e-4):
            raise SystemExit(f"a replay after a weight change does not use the changed weights: {check}")
        out["replay_check"] = check
    if dist is not None and arena is not None:
        out["exchange_check"] = exchange_check()
        out["collective"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "grad_dtype": a.grad_dtype,
                             "payload_bytes_per_step": int(arena.flat.numel() * (4 if a.grad_dtype == "f32" else 2)),
                             "used_flags": "device" if (a.device_flags and overlap) else "host read per step",
                             "launched_per_step": coll / a.steps, "captured_in_graph": bool(a.captured_collective and overlap and graphed is not None),
                             "captured_collectives_per_step": (fused.captured if (graphed is not None and fused is not None) else 0), "forced_on_one_rank": bool(arena.force_collective and world == 1)}
    if rank == 0:
        if not a.no_roofline:
            from mi_seg_amd.testing.roofline import profile_step, summarize
            prof = profile_step(lambda: step(a.warmup + a.steps, eager=True, comm=False))
            out["roofline"] = summarize(prof, dtype, counters=(a.workload == "c2" and a.dtype == "bf16"))      # rank 0 alone: no collectives
            if a.dump_launches:      # every hooked launch of that step: {kernel: [(ms, flops, algorithmic bytes, ms of all its kernels)]} in issue order
                with open(a.dump_launches, "w") as f:
                    json.dump(prof, f)
        if world == 1 and not a.no_cpu_baseline:
            from mi_seg_amd.testing.cpu_baseline import cpu_baseline
            out["cpu_baseline"] = cpu_baseline(a.workload)
    if arena is not None:
        arena.detach()
    return out


def bench_train_step(a, model, dtype, dev, pool, mods, order, params):
    """one step = one whole optimisation step of the reference's loop (LitMonai.training_step + optimizer.step, lightning_monai.py:149-166, 255-278)
    as ONE replayed hipGraph: versioned weight refresh (live: the optimiser changed every weight), forward, fused DiceFocal loss + dlogits,
    backward, one-launch AdamW over the gradient arena.  N = 1."""
    from mi_seg_amd.runtime.arena import ParamArena
    from mi_seg_amd.runtime.graph import GraphedTrainStep
    from mi_seg_amd.training.losses import DiceFocalLoss
    from mi_seg_amd.training.optim import ArenaOptimizer
    arena = ParamArena(params, dtype)
    opt = ArenaOptimizer(arena, "adamw", lr=1e-4, weight_decay=1e-5)                       # LitMonai's defaults (lightning_monai.py:30)
    crit = DiceFocalLoss(include_background=False, to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr=0.0, smooth_dr=1e-6)
    labels = (pool * 6).floor().clamp_(0, 5).to(torch.int32)                               # synthetic 6-class labels from the intensity bands
    gts = GraphedTrainStep(model, crit, opt, (1, 1, 96, 96, 96), (1, 1, 96, 96, 96), arena)
    w0 = arena.params[0].detach().clone()
    losses = []
    for m_ in sorted(set(mods)):          # capture both graphs outside the timed region
        k_ = mods.index(m_)
        gts(pool[k_:k_ + 1], labels[k_:k_ + 1], [m_])
    for i in range(a.warmup):
        k = order[i % len(order)]
        gts(pool[k:k + 1], labels[k:k + 1], [mods[k]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        k = order[(a.warmup + i) % len(order)]
        losses.append(gts(pool[k:k + 1], labels[k:k + 1], [mods[k]]).clone())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = [float(l) for l in losses]
    moved = float((arena.params[0].detach() - w0).abs().max())
    if not all(l == l and abs(l) < 1e6 for l in losses) or moved == 0.0:
        raise SystemExit(f"train step: losses {losses}, first parameter moved by {moved}")
    out = {"metric": WORKLOADS[a.workload][0].replace("fwd+bwd", "full optimisation step (refresh + fwd + DiceFocal + bwd + AdamW)"), "value": a.steps / dt,
           "unit": WORKLOADS[a.workload][1], "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000 * dt / a.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic", "launch": "hipGraph",
           "config": {"workload": WORKLOADS[a.workload][2] + " + fused DiceFocal loss + one-launch AdamW + live weight refresh, one hipGraph per step"},
           "train_step": {"losses_first_last": [losses[0], losses[-1]], "optimizer": "adamw lr 1e-4 wd 1e-5 (one launch over the arena)",
                          "loss": "DiceFocal (fused forward + dlogits)"}}
    arena.detach()
    return out


def bench_sliding_window(a, model, dtype, dev, rank, world, dist):
    """BASELINE configs[4]: one step = the sliding-window inference of one whole 512 x 512 x 363 volume resident in HBM (700 windows)."""
    from mi_seg_amd.runtime.graph import GraphedForward
    from mi_seg_amd.training.inferer import sliding_window_inference, window_grid
    size, sw = (512, 512, 363), int(os.environ.get("MISEG_SW_BATCH", "20"))      # windows per forward (modality broadcast): 4 / 7 / 10 / 14 / 20 -> 593 / 619 / 627 / 632 / 643 windows/s
    vol = torch.rand(1, 1, *size, generator=torch.Generator().manual_seed(2000 + rank)).to(dev)
    nwin = len(window_grid(size, (96, 96, 96), 0.5))
    from mi_seg_amd.runtime.arena import ParamArena
    arena = ParamArena(list(model.parameters()), dtype)        # inference: the weights' compute-dtype copies / conv packs are made once
    pred = GraphedForward(model, (sw, 1, 96, 96, 96), arena=arena)
    run = lambda: sliding_window_inference(vol, 96, sw, pred, overlap=0.5, modalities=[0])
    for _ in range(max(1, a.warmup)):
        y = run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # untimed: replays of the captured forward against the eager forward on window batches from the corners, the centre and the clamped
    # last windows of the volume - `finite` alone let a replay bug through in round 2 (statistics pool recycled short: every replay after
    # the first normalised with accumulated sums, logits 43 % off and perfectly finite)
    grid = window_grid(size, (96, 96, 96), 0.5)
    worst = 0.0
    with torch.no_grad():
        for i0 in (0, 348, nwin - sw, sw, 0):
            xb = torch.cat([vol[:, :, d:d + 96, h:h + 96, w:w + 96] for (d, h, w) in grid[i0:i0 + sw]], 0).contiguous()
            yg = pred(xb, [0] * sw).float().clone()
            ye = model(xb, [0] * sw).float()
            worst = max(worst, float((yg - ye).norm() / ye.norm()))
    if not worst < 1e-2:
        raise SystemExit(f"sliding window: the replayed forward is {worst:.3e} from the eager forward of the same window batch")
    out = {"metric": WORKLOADS["c5"][0], "value": world * a.steps * nwin / dt, "unit": "windows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1000 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "launch": "hipGraph", "config": {"workload": WORKLOADS["c5"][2], "windows_per_volume": nwin, "sw_batch_size": sw,
                                            "replicas": "one volume per GPU, no collective (windows are independent)"},
           "finite": bool(torch.isfinite(y).all()), "replay_check": {"window_batches": 5, "worst_rel_err_vs_eager": worst},
           "volumes_per_s": world * a.steps / dt}
    if rank == 0:
        if not a.no_roofline:
            from mi_seg_amd.testing.roofline import profile_step, summarize
            xw = vol[:, :, :96, :96, :96].expand(sw, -1, -1, -1, -1).contiguous()

            def one_batch():
                with torch.no_grad():
                    model(xw, [0] * sw)
            out["roofline"] = summarize(profile_step(one_batch), dtype, counters=False)
    arena.detach()
    return out


if __name__ == "__main__":
    main()
