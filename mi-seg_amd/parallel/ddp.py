"""Data-parallel gradient exchange: one process per GPU, mean all-reduce of a flat fp32 gradient arena over RCCL
(backend "nccl" on ROCm) -- the counterpart of the reference's DDP wrap (tune.py:103-109, find_unused_parameters=True).

Parameters of a style absent from the local batch have no local gradient; their arena slots are zero-filled, and a
"used anywhere" bitmap is max-reduced so that parameters unused on EVERY rank keep ``grad is None`` (no optimiser
update, no weight decay) exactly like torch DDP's reducer.
"""
from typing import List

import torch
import torch.distributed as dist

_ARENAS = {}


class GradArena:
    def __init__(self, params: List[torch.nn.Parameter], n_buckets: int = 4):
        self.params = params
        self.numels = [p.numel() for p in params]
        total = sum(self.numels)
        dev = params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p, n in zip(params, self.numels):
            self.views.append(self.flat[off:off + n].view(p.shape))
            off += n
        # bucket boundaries on parameter edges, roughly equal bytes; reduced last-to-first (reverse autograd order)
        self.buckets, acc, start, target = [], 0, 0, total / n_buckets
        offs = [0]
        for n in self.numels:
            offs.append(offs[-1] + n)
        for i, n in enumerate(self.numels):
            acc += n
            if acc >= target or i == len(self.numels) - 1:
                self.buckets.append((offs[start], offs[i + 1]))
                start, acc = i + 1, 0
        self.used = torch.zeros(len(params), dtype=torch.int32, device=dev)

    def reduce(self, world_size: int, group=None):
        present = [p.grad is not None for p in self.params]
        src = [p.grad for p, ok in zip(self.params, present) if ok]
        dst = [v for v, ok in zip(self.views, present) if ok]
        absent = [v for v, ok in zip(self.views, present) if not ok]
        if absent:
            torch._foreach_zero_(absent)
        if src:
            torch._foreach_copy_(dst, src)
        self.used.copy_(torch.tensor(present, dtype=torch.int32), non_blocking=True)
        works = [dist.all_reduce(self.used, op=dist.ReduceOp.MAX, group=group, async_op=True)]
        for lo, hi in reversed(self.buckets):
            works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()
        self.flat.mul_(1.0 / world_size)
        used = self.used.tolist()
        for p, v, u in zip(self.params, self.views, used):
            p.grad = v if u else None


def allreduce_gradients(params, world_size: int, group=None):
    key = (id(params[0]), len(params))
    arena = _ARENAS.get(key)
    if arena is None:
        arena = _ARENAS[key] = GradArena(list(params))
    arena.reduce(world_size, group)
    return arena
