"""Multi-process (gloo, world_size 2, CPU) tests of the N>1 path: flat gradient arena mean all-reduce with the
"no gradient on any rank => stays None" rule, and the interleaved CT/MR rank sharding."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    from mi_seg_amd.data.sampler import concat_modalities, rank_indices
    from mi_seg_amd.parallel.ddp import allreduce_gradients
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    # a toy "conditional" parameter set: shared weight, style-0 rows, style-1 rows, a never-used parameter
    shared = torch.nn.Parameter(torch.zeros(5, 3))
    s0 = torch.nn.Parameter(torch.zeros(4))
    s1 = torch.nn.Parameter(torch.zeros(4))
    unused = torch.nn.Parameter(torch.zeros(2))
    params = [shared, s0, s1, unused]
    shared.grad = torch.full((5, 3), float(rank + 1))
    if rank == 0:
        s0.grad = torch.full((4,), 2.0)      # rank 0 saw a CT patch
    else:
        s1.grad = torch.full((4,), 6.0)      # rank 1 saw an MR patch
    allreduce_gradients(params, world)
    ok = True
    ok &= torch.allclose(shared.grad, torch.full((5, 3), 1.5))
    ok &= torch.allclose(s0.grad, torch.full((4,), 1.0))       # (2 + 0) / 2: absent on rank 1 -> zero slot
    ok &= torch.allclose(s1.grad, torch.full((4,), 3.0))
    ok &= unused.grad is None                                   # unused on every rank: untouched by the optimiser
    # second step reuses the arena
    shared.grad = torch.full((5, 3), 4.0)
    s0.grad, s1.grad, unused.grad = None, None, None
    allreduce_gradients(params, world)
    ok &= torch.allclose(shared.grad, torch.full((5, 3), 4.0)) and s0.grad is None and s1.grad is None
    # the training arena (what bench.py uses at N > 1): kernels have already summed into the flat buffer, `_miseg_used` says which slots
    from mi_seg_amd.runtime.arena import ParamArena
    for p in params:
        p.grad = None
    arena = ParamArena(params, torch.float32)
    for it in range(2):
        arena.flat.zero_()
        for p in params:
            p._miseg_used = False
        shared._miseg_grad.fill_(float(rank + 1) + it)
        shared._miseg_used = True
        mine_p = s0 if rank == 0 else s1
        mine_p._miseg_grad.fill_(2.0 if rank == 0 else 6.0)
        mine_p._miseg_used = True
        arena.allreduce(world)
        ok &= torch.allclose(shared.grad, torch.full((5, 3), 1.5 + it))
        ok &= torch.allclose(s0.grad, torch.full((4,), 1.0)) and torch.allclose(s1.grad, torch.full((4,), 3.0))
        ok &= unused.grad is None
        ok &= shared.grad.data_ptr() == shared._miseg_grad.data_ptr()      # p.grad IS the arena slot (no copies)
    # the overlapped exchange of bench.py at N > 1: the tail of the arena (parameters whose gradients are final after the first half
    # of the backward pass) starts its all-reduce early, the rest follows after the second half
    tail = arena.tail_offset([s1, unused])
    ok &= tail == arena._offs[2]
    try:
        arena.tail_offset([s0, unused])
        ok = False
    except ValueError:
        pass
    for it in range(2):
        arena.flat.zero_()
        for p in params:
            p._miseg_used = False
        mine_p = s0 if rank == 0 else s1
        mine_p._miseg_grad.fill_(2.0 if rank == 0 else 6.0)
        mine_p._miseg_used = True
        works = arena.allreduce_begin(tail, arena.flat.numel(), piece=3)          # tiny pieces: several collectives per range
        shared._miseg_grad.fill_(float(rank + 1) + it)                             # "second half of the backward pass"
        shared._miseg_used = True
        arena.allreduce_end(works, world, rest=(0, tail))
        ok &= torch.allclose(shared.grad, torch.full((5, 3), 1.5 + it))
        ok &= torch.allclose(s0.grad, torch.full((4,), 1.0)) and torch.allclose(s1.grad, torch.full((4,), 3.0))
        ok &= unused.grad is None
    # a hole in the early range (the side branch defers some decoder-side weight gradients into the second half: bench.py): the tail goes out
    # as the ranges around it, the hole joins the ranges reduced at the end
    hole = arena.param_range([s1])
    ok &= hole == (arena._offs[2], arena._offs[3]) and tail <= hole[0]
    try:
        arena.param_range([shared, s1])       # not adjacent
        ok = False
    except ValueError:
        pass
    for it in range(2):
        arena.flat.zero_()
        for p in params:
            p._miseg_used = False
        works = []
        for lo, hi in ((hole[1], arena.flat.numel()), (tail, hole[0])):
            if hi > lo:
                works += arena.allreduce_begin(lo, hi)
        shared._miseg_grad.fill_(float(rank + 1) + it)                             # second half: the early part of the arena ...
        shared._miseg_used = True
        s1._miseg_grad.fill_(6.0 if rank == 1 else 0.0)                            # ... and the deferred slot
        s1._miseg_used = rank == 1
        arena.allreduce_end(works, world, rest=[(0, tail), hole])
        ok &= torch.allclose(shared.grad, torch.full((5, 3), 1.5 + it)) and torch.allclose(s1.grad, torch.full((4,), 3.0))
        ok &= s0.grad is None and unused.grad is None
    # the same exchange with the "used on any rank" flags left on the device side (bench.py's default at N > 1: no host read per step) and with
    # bf16 gradient buckets (bench.py --grad-dtype bf16): the local sums are rounded to bf16 for the wire, the mean returns into the fp32 arena
    arena16 = ParamArena(params, torch.float32, grad_dtype=torch.bfloat16)
    for it in range(2):
        arena16.flat.zero_()
        for p in params:
            p._miseg_used = False
        mine_p = s0 if rank == 0 else s1
        mine_p._miseg_grad.fill_(2.0 if rank == 0 else 6.0)
        mine_p._miseg_used = True
        shared._miseg_used = True                                                  # (a replayed hipGraph knows all its flags before it starts)
        ub = arena16.used_begin(host=False)
        works = arena16.allreduce_begin(tail, arena16.flat.numel())
        shared._miseg_grad.fill_((1.0 + 2.0 ** -10) * (rank + 1))                 # rounds to (rank + 1) in bf16
        arena16.allreduce_end(works, world, rest=(0, tail), used_work=ub, host_flags=False)
        ok &= torch.equal(shared.grad, torch.full((5, 3), 1.5))                    # the bf16-rounded values were averaged
        ok &= arena16.used_on_device and arena16.used_dev.tolist() == [1, 1, 1, 0]  # the GLOBAL flags, for the fused optimiser
        ok &= (s0.grad is not None) == (rank == 0) and (s1.grad is not None) == (rank == 1)      # p.grad follows the LOCAL flags
        ok &= torch.allclose(s0._miseg_grad, torch.full((4,), 1.0)) and torch.allclose(s1._miseg_grad, torch.full((4,), 3.0))
        ok &= unused.grad is None
    arena16.detach()
    for p, v in zip(params, arena.views):                                           # back to the fp32 arena for the rest of the test
        p._miseg_grad, p._miseg_arena, p._miseg_used = v, arena, False
    # gradient accumulation (reference utils/trainer.py:55-68: DDP no_sync() on the micro-batches that do not step): two local micro-batches,
    # ONE collective over their sum; the window after it starts from zero again
    import mi_seg_amd.hip.ops as _ops
    _ops.fill32 = lambda t, word=0: t.zero_()                  # CPU stand-ins for the two device fills begin_step issues
    _ops.begin_step = lambda: None
    for window in range(2):
        with arena.no_sync():
            arena.begin_step()
            shared._miseg_grad.add_(float(rank + 1))
            shared._miseg_used = True
            before = shared._miseg_grad.clone()
            arena.allreduce(world)
            ok &= torch.equal(shared.grad, before)            # no exchange happened
        arena.begin_step()                                     # keeps the arena: still accumulating
        shared._miseg_grad.add_(10.0 * (rank + 1))
        arena.allreduce(world)
        ok &= torch.allclose(shared.grad, torch.full((5, 3), 1.5 + 15.0))
    arena.detach()
    # rank sharding: disjoint cover of the concatenated CT+MR index range
    mine = rank_indices(32, world, rank, epoch=1, seed=0)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    ok &= sorted(sum(gathered, [])) == list(range(32))
    mods = concat_modalities(16, 16)
    ok &= set(mods[i] for i in mine) <= {0, 1}
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradient_arena_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(30)
    assert res == {0: True, 1: True}
