"""Per-model training arena: ONE flat fp32 gradient buffer that the weight-gradient kernels accumulate into directly
(one fill per step instead of one per parameter, and the buffer RCCL all-reduces at N > 1), plus the compute-dtype
re-layouts of the parameters (bf16 copies, transposes, ConvTranspose regroupings) refreshed by ONE kernel per step.

WHO MAY CHANGE THE PARAMETERS, AND HOW THE COPIES FOLLOW (the versioned refresh).  The refresh launches of a step re-lay-out the weights only
when the device-side parameter version moved since they last ran.  It moves when
  * the fused optimiser steps (training/optim.py: `miseg_opt_step` bumps it on the device - also inside a replayed hipGraph);
  * a parameter is modified THROUGH THE PARAMETER with torch (an in-place op, `copy_`, a torch optimiser, `load_state_dict`: they bump
    `Tensor._version`) or re-pointed (`set_`, `.data = ...`: another `data_ptr`) - `params_changed()`, called by `begin_step()`,
    `refresh_weights()`, `GraphedStep` / `GraphedForward` / `GraphedTrainStep` before every replay, notices both and bumps it;
  * the caller says so: `arena.invalidate()` - REQUIRED after any other write: in-place ops on the `p.data` / `p.detach()` alias (it has a
    version counter of its own: an EMA swap-in or a weight clamp written that way is invisible), raw-pointer writes of custom kernels.
`MISEG_REFRESH_ALWAYS=1` in the environment makes every refresh unconditional (the round-2 behaviour, ~0.2 ms per C-Swin-UNETR step).

Without an arena the autograd Functions in hip/functional.py allocate and return each gradient (torch semantics,
``grad is None`` for parameters that were not used); with one they add into ``p._miseg_grad`` and return None, and
``publish()`` sets ``p.grad`` to the arena views of the parameters used since the last ``begin_step`` (None otherwise,
like the reference's find_unused_parameters DDP: tune.py:103-109).  Gradient accumulation over micro-batches:
``begin_step(zero=False)``.
"""
import ctypes as C

import os

import torch

from ..hip import lib as L
from ..hip import ops


SKIP_OVERWRITTEN_FILL = not os.environ.get("MISEG_FULL_ARENA_FILL")      # A/B switch of round 5 (read once)


class ParamArena:
    def __init__(self, params, dtype=torch.bfloat16, n_buckets=4, overlap_wgrad=False, grad_dtype=torch.float32, force_collective=None):
        """force_collective (default: the environment's MISEG_FORCE_COLLECTIVE=1): a one-rank process group normally skips every collective (the
        local sums ARE the mean); with this set the exchanges are launched anyway - the RCCL leg then runs, and is testable, on a one-GPU box
        (tests/test_hip_rccl.py, `bench.py --force-dist`).  `collectives_launched` counts every collective this arena has issued.
        grad_dtype: the dtype the gradient buckets travel in.  float32 (default) averages the fp32 sums themselves, like the reference's DDP
        (tune.py:103-109); bfloat16 halves the bytes on xGMI (248.9 -> 124.5 MB per step for C-Swin-UNETR fs=48): every range is rounded into a
        bf16 staging buffer, averaged there, and written back to the fp32 arena when the exchange is waited for - the local sums stay fp32,
        the exchanged mean carries one bf16 rounding per rank."""
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        offs, off = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("parameters must be contiguous float32")
            offs.append(off)
            off += (p.numel() + 3) & ~3          # 16-byte aligned slots
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, self.params)]
        for p, v in zip(self.params, self.views):
            p._miseg_grad, p._miseg_arena, p._miseg_used = v, self, False
        self.dtype = dtype
        self.wgrad_stream = torch.cuda.Stream(device=dev) if (overlap_wgrad and dev.type == "cuda") else None
        self.epoch = 0
        self._req = {}          # (id(p), transpose, inner, outer) -> [param, shadow, filled_epoch]
        self._table = None      # (device descriptor bytes, ndesc, total tiles)
        self._retired = []      # tables a captured hipGraph may still point at
        self._packs = {}        # id(p) -> [param, fwd pack, bwd pack, filled_epoch, in table]   (3x3x3 conv weights)
        self._ptable = None
        self._pdirty = False
        self._dirty = False
        # versioned refresh (include/miseg_hip.h, miseg_pack_conv3_batch): [0] = version of the fp32 parameters (bumped on the device by the
        # fused optimiser, by invalidate() otherwise), [1..2] / [3..4] = (version held, arrival counter) of the cast copies / the conv packs.
        # The refresh launches of a step re-lay-out nothing while the parameters have not changed - decided on the DEVICE, so a replayed
        # hipGraph does the right thing after an optimiser step without the host knowing
        self.versions = torch.zeros(5, dtype=torch.int64, device=dev)
        self.versions[1] = -1
        self.versions[3] = -1
        # all-reduce buckets on slot edges, roughly equal bytes, reduced last-to-first (reverse autograd order)
        ends = offs[1:] + [off]
        self.buckets, start, target = [], 0, off / n_buckets
        for i, e in enumerate(ends):
            if e - start >= target or i == len(ends) - 1:
                self.buckets.append((start, e))
                start = e
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_dtype must be torch.float32 or torch.bfloat16")
        self.grad_dtype = grad_dtype
        # bf16 staging copy of the arena (grad_dtype bfloat16): allocated HERE, eagerly - first use inside a capture would put it into the
        # graph's private pool, and exchange_check reads it eagerly afterwards
        self._stage = torch.empty(off, dtype=grad_dtype, device=dev) if grad_dtype != torch.float32 else None
        self._staged = []       # ranges whose exchanged values still sit in the staging buffer
        self.used_dev = torch.zeros(len(self.params), dtype=torch.int32, device=dev)
        self._bm = None         # (stream, pinned result, event) of the bitmap exchange on a card
        self.used_on_device = False     # True after an exchange that left the global "used" flags in `used_dev` only (allreduce_end(host_flags=False))
        self._offs, self._size = offs, off
        import os
        self.force_collective = bool(int(os.environ.get("MISEG_FORCE_COLLECTIVE", "0") or 0)) if force_collective is None else bool(force_collective)
        self.collectives_launched = 0
        self._qkey = self.flat.untyped_storage().data_ptr()
        self.queues = None
        self.pool = ops._ZeroPool() if self.flat.is_cuda else None      # this model's statistics scratch (ops.use_pool); CPU arenas (gloo tests) have none

    # ---------------------------------------------------------------------------------------------- gradient accumulation
    def no_sync(self):
        """context manager for the micro-batches of a gradient-accumulation window that do NOT end in an optimiser step (reference
        utils/trainer.py:55-68 wraps them in DDP's `model.no_sync()`): inside it `allreduce()` only publishes the local sums, and the next
        `begin_step()` keeps the arena (`zero=False` is implied) - the collective runs once, on the last micro-batch, over the accumulated sums."""
        arena = self

        class _NoSync:
            def __enter__(self_):
                arena._sync = False
                return arena

            def __exit__(self_, *exc):
                arena._sync = True
                return False
        return _NoSync()

    # ---------------------------------------------------------------------------------------------- per step
    def begin_step(self, zero=True):
        """once per optimisation step, before the forward: recycles the statistics pool, zeroes the gradient arena and
        refreshes every registered parameter re-layout (safe inside hipGraph capture once the table exists)."""
        ops.use_pool(self.pool)          # this arena's statistics pool is the active one from here on
        ops.begin_step()
        if not (self.flat.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.params_changed()
        # the launch-bound tails of this step's backward pass (bias-gradient reductions, small weight-gradient GEMMs, partial-tile sums,
        # the conv weight gradients of the 48^3-and-smaller layers) are queued per arena and issued by end_backward()
        self.queues = ops.QUEUES[self._qkey] = ops.StepQueues()
        self.queues.on_branch_end = getattr(self, "branch_end_hook", None)
        ops.WGRAD_STREAM = self.wgrad_stream      # None unless overlap_wgrad: measured SLOWER on one MI355X (93.3 -> 90 patches/s: the
                                                  # cross-stream edges of the hipGraph cost more than the idle CUs they fill)
        self.epoch += 1
        if getattr(self, "_accumulating", False):     # the previous micro-batch ran under no_sync(): keep summing into the arena
            zero = False
        if zero:
            self._zero_fill()
            for p in self.params:
                p._miseg_used = False
        self._refresh()

    def _zero_fill(self):
        """the step's zero fill of the gradient arena - minus the slots whose weight-gradient kernel overwrites them whole (round 5): the
        tiny-volume conv weights of encoder10 / decoder5 are 70 % of the headline net's 249 MB, and their kernel stores every element without
        reading it.  A slot is left out when the PREVIOUS step wrote it that way (StepQueues.inline_final -> self._overwritten); the queue of
        this step carries the set (`unzeroed`), so that ops.conv3_wgrad fills a slot after all if the launch it is about to issue is not the
        overwriting one, and end_backward fills the ones nothing wrote (an unused parameter's slot must read zero for the all-reduce)."""
        skip = [i for i in getattr(self, "_overwritten", ()) if SKIP_OVERWRITTEN_FILL]
        if skip:
            # data-parallel steps reduce ranges of the arena BEFORE end_backward settles the left-out slots: they keep the full fill
            import torch.distributed as dist
            if self.force_collective or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
                skip = []
        if not skip or not self.flat.is_cuda:
            ops.fill32(self.flat)
            return
        key = tuple(skip)
        cached = self.__dict__.get("_fill_plan")
        if cached is None or cached[0] != key:
            ranges, pos = [], 0
            for i in skip:                                       # ascending parameter order = ascending offsets
                lo = self._offs[i]
                hi = self._offs[i + 1] if i + 1 < len(self._offs) else self._size
                if lo > pos:
                    ranges.append((pos, lo - pos))
                pos = hi
            if self._size > pos:
                ranges.append((pos, self._size - pos))
            cached = self._fill_plan = (key, ranges)
        ops.fill32_ranges(self.flat, cached[1])
        self.queues.unzeroed = {self.views[i].data_ptr() for i in skip}

    def _refresh(self):
        """one launch each: every registered parameter re-layout (casts / transposes / regroupings) and every 3x3x3 weight pack"""
        if self._dirty:
            self._build_table()
        if self._table is not None:
            buf, n, tiles = self._table
            lib = L.load()
            L.check(lib.miseg_param_cast_batch(buf.data_ptr(), n, tiles, L.F32 if self.dtype == torch.float32 else L.BF16, *self._ver(1), ops._stream()),
                    "param_cast_batch")
            for ent in self._req.values():
                if ent[3]:
                    ent[2] = self.epoch
        if self._pdirty:
            self._build_pack_table()
        if self._ptable is not None:
            buf, n, tiles = self._ptable
            lib = L.load()
            L.check(lib.miseg_pack_conv3_batch(buf.data_ptr(), n, tiles, L.F32 if self.dtype == torch.float32 else L.BF16, *self._ver(3), ops._stream()),
                    "pack_conv3_batch")
            for ent in self._packs.values():
                if ent[4]:
                    ent[3] = self.epoch

    def _ver(self, slot):
        """(params_version, state) device pointers of the versioned refresh launches; (None, None) = unconditional (MISEG_REFRESH_ALWAYS=1)"""
        import os
        if os.environ.get("MISEG_REFRESH_ALWAYS"):
            return None, None
        base = self.versions.data_ptr()
        return C.c_void_p(base), C.c_void_p(base + 8 * slot)

    def params_version_ptr(self):
        """device int64 that counts parameter changes: hand it to miseg_opt_step (training/optim.py does)"""
        return self.versions.data_ptr()

    def params_changed(self):
        """True when a parameter was modified through torch since the last call (in-place ops bump `Tensor._version`: a torch optimiser's
        step, load_state_dict, .copy_()); the device-side parameter version is bumped then, so the next refresh launch - eager or inside a
        replayed hipGraph - re-lays-out the copies.  The fused optimiser writes through raw pointers and bumps the device version itself."""
        v = 0
        for p in self.params:
            v += p._version + (p.data_ptr() & 0xFFFFFFFFFFFF) * 31          # (a re-pointed parameter - set_, .data = ... - changes the address)
        if v != self.__dict__.get("_pver"):
            first = "_pver" not in self.__dict__
            self._pver = v
            if not first:
                self.invalidate()
            return True
        return False

    def refresh_weights(self):
        """inference / validation: bring the compute-dtype copies of the parameters up to date WITHOUT opening a training step (no arena
        fill, no queues).  They stay valid - for any number of forward passes - until `invalidate()`; buffers keep their addresses, so
        hipGraphs captured on them (runtime/graph.py::GraphedForward) stay valid across refreshes.  The first forward after a model was
        built registers its re-layouts lazily (and casts per call); call this again after it."""
        self.params_changed()
        self.epoch += 1
        self._refresh()

    def invalidate(self):
        """PUBLIC: tell the arena that parameters were written behind its back (see the module docstring: in-place ops on `p.data`, raw-pointer
        writes, anything that does not bump `Parameter._version`).  Bumps the device-side parameter version - the next refresh launch, eager
        or inside a replayed hipGraph, re-casts and re-packs every weight - and retires the host-side epoch, so that eager forwards cast per
        call until the next `begin_step()` / `refresh_weights()`.  Cheap (one one-thread launch); calling it needlessly only costs the
        re-layout.  (The fused optimiser does NOT call this: its kernel bumps the version on the device itself; torch-level updates are
        noticed by `params_changed()`.)"""
        self.epoch += 1
        if self.versions.is_cuda:
            L.check(L.load().miseg_counter_add(C.c_void_p(self.versions.data_ptr()), 1, ops._stream()), "counter_add")

    def shadow(self, p, transpose, inner, outer):
        """the re-layout of parameter p refreshed this step, or None (first request: registered for the next step)."""
        key = (id(p), bool(transpose), inner, outer)
        ent = self._req.get(key)
        if ent is None:
            R = p.shape[0]
            Cc = p.numel() // R
            dst = torch.empty((Cc, R) if transpose else (R, Cc), dtype=self.dtype, device=p.device)
            self._req[key] = [p, dst, -1, False]     # param, shadow, epoch of last refresh, in table
            self._dirty = True
            return None
        return ent[1] if ent[2] == self.epoch else None

    def conv_packs(self, p):
        """(fwd pack, bwd pack) of the 3x3x3 weight p refreshed this step, or None (first request: registered)."""
        ent = self._packs.get(id(p))
        if ent is None:
            lib = L.load()
            dt = L.F32 if self.dtype == torch.float32 else L.BF16
            Cout, Cin = p.shape[0], p.shape[1]
            fwd = torch.empty(lib.miseg_pack_conv3_elems(Cin, Cout, dt, 0), dtype=self.dtype, device=p.device)
            bwd = torch.empty(lib.miseg_pack_conv3_elems(Cin, Cout, dt, 1), dtype=self.dtype, device=p.device)
            self._packs[id(p)] = [p, fwd, bwd, -1, False]
            self._pdirty = True
            return None
        return (ent[1], ent[2]) if ent[3] == self.epoch else None

    def _build_pack_table(self):
        if torch.cuda.is_current_stream_capturing():
            return
        descs = (L.PackConv3Desc * len(self._packs))()
        lib = L.load()
        dt = L.F32 if self.dtype == torch.float32 else L.BF16
        tile0 = 0
        for i, ent in enumerate(self._packs.values()):
            p = ent[0]
            Cout, Cin = p.shape[0], p.shape[1]
            descs[i] = L.PackConv3Desc(p.data_ptr(), ent[1].data_ptr(), ent[2].data_ptr(), Cin, Cout, tile0, 0)
            tile0 += lib.miseg_pack_conv3_tiles(Cin, Cout, dt)
            ent[4] = True
        if self._ptable is not None:
            self._retired.append(self._ptable[0])
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.flat.device)
        self._ptable = (raw, len(self._packs), tile0)
        self._pdirty = False
        self.versions[3] = -1          # new packs in the table: the next refresh launch fills all of them

    def _build_table(self):
        if torch.cuda.is_current_stream_capturing():
            return                                   # host->device copy: wait for an eager step
        descs = (L.CastDesc * len(self._req))()
        tile0 = 0
        for i, (key, ent) in enumerate(self._req.items()):
            p, dst = ent[0], ent[1]
            R = p.shape[0]
            Cc = p.numel() // R
            descs[i] = L.CastDesc(p.data_ptr(), dst.data_ptr(), R, Cc, int(key[1]), key[2], key[3], tile0)
            tile0 += ((R + 31) // 32) * ((Cc + 31) // 32)
            ent[3] = True
        if self._table is not None:
            self._retired.append(self._table[0])
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.flat.device)
        self._table = (raw, len(self._req), tile0)
        self._dirty = False
        self.versions[1] = -1          # new copies in the table: the next refresh launch fills all of them

    # ---------------------------------------------------------------------------------------------- after backward
    def end_backward(self):
        """issue the queued bias-gradient column sums (one launch per 32); call after loss.backward(), inside the captured
        region when the step is a hipGraph.  publish() / allreduce() call it too."""
        # the main stream's grouped launches first (beside the branch's last kernels), the wait, then what the branch's backward queued
        ops.join_branch(queues=self.queues, flush_main=ops.FLUSH_MAIN_BEFORE_JOIN)
        if self.queues is not None:
            self.queues.flush()
            # parameters whose gradient was complete when the main chain of the pass ended (see StepQueues.inline_final)
            ptrs = {t.data_ptr() for t in self.queues.inline_final}
            self.inline_final_params = [p for p, v in zip(self.params, self.views) if v.data_ptr() in ptrs]
            # slots the fill left out and nothing wrote (the parameter went unused): they must read zero from here on
            for i, v in enumerate(self.views):
                if v.data_ptr() in self.queues.unzeroed:
                    ops.fill32(v)
            # ... and the slots this step's kernels overwrote whole: the next step's fill leaves them out (_zero_fill)
            self._overwritten = sorted(i for i, v in enumerate(self.views) if v.data_ptr() in ptrs)
            self.queues = None
            ops.QUEUES.pop(self._qkey, None)
        ops.stamp("queues_flushed")
        ops.check_no_pending()
        ops.join_wgrad()
        ops.WGRAD_STREAM = None

    def flush_small(self):
        """between the two halves of a split backward pass when the range that goes out early holds only layers whose conv weight gradients were
        written where the pass reached them (encoder10 / decoder5: the tiny-volume kernel): the queued GEMM weight gradients, partial sums and
        column sums complete its other parameters; the grouped conv weight gradients keep waiting for the end of the pass"""
        ops.join_branch(flush_deferred=False, queues=None, flush_main=False)      # (a capture must not end with the branch stream unjoined)
        if self.queues is not None:
            self.queues.flush_small()

    def flush(self):
        """issue everything queued so far and keep queueing (between the two halves of a split backward pass)."""
        # (weight gradients deferred to the branch's backward pass wait for the second half unless the model already issued them:
        # SwinUNETR.split_defers = "early").  As in end_backward: the grouped GEMM gradients / column sums go out on the branch stream beside
        # the main stream's grouped conv weight gradients
        ops.join_branch(flush_deferred=False, queues=self.queues, flush_main=ops.FLUSH_MAIN_BEFORE_JOIN)
        if self.queues is not None:
            self.queues.flush()

    def publish(self):
        self.end_backward()
        self._accumulating = not getattr(self, "_sync", True)
        for p, v in zip(self.params, self.views):
            p.grad = v if p._miseg_used else None

    def allreduce(self, world_size, group=None):
        """mean all-reduce of the arena over RCCL in n_buckets pieces; parameters unused on EVERY rank keep grad None."""
        import torch.distributed as dist
        self.end_backward()
        if not getattr(self, "_sync", True):          # inside no_sync(): local accumulation only
            self.publish()
            return
        ub = self.used_begin(group)
        works = []
        avg = self._avg(group)
        if not self._single(group):
            for lo, hi in reversed(self.buckets):
                works.append(self._reduce_range(lo, hi, group))
        self.used_on_device = False
        used = ub()
        for w in works:
            w.wait()
        self._unstage()
        if not avg:
            self.flat.mul_(1.0 / world_size)
        for p, u in zip(self.params, used):
            p._miseg_used = bool(u)
        self.publish()

    def _single(self, group=None):
        """a one-rank group whose collectives are skipped (nothing to exchange) - unless `force_collective` asks for them anyway"""
        import torch.distributed as dist
        return dist.get_world_size(group) == 1 and not self.force_collective

    def _all_reduce(self, t, op, group):
        import torch.distributed as dist
        self.collectives_launched += 1
        return dist.all_reduce(t, op=op, group=group, async_op=True)

    def _convert(self, src, dst):
        """dtype-converting copy of a 16-byte-aligned range (our strided-copy kernel on the card, torch on the CPU of the gloo tests)"""
        if src.is_cuda:
            ops.copy2d(src.view(-1, 4), dst.view(-1, 4))
        else:
            dst.copy_(src)

    def _reduce_range(self, lo, hi, group):
        """start the mean (RCCL) / sum (gloo) all-reduce of flat[lo:hi] in `grad_dtype`; returns the work handle"""
        import torch.distributed as dist
        op = dist.ReduceOp.AVG if self._avg(group) else dist.ReduceOp.SUM
        if self.grad_dtype == torch.float32:
            return self._all_reduce(self.flat[lo:hi], op, group)
        if self._stage is None:
            self._stage = torch.empty(self._size, dtype=self.grad_dtype, device=self.flat.device)
        self._convert(self.flat[lo:hi], self._stage[lo:hi])
        self._staged.append((lo, hi))
        return self._all_reduce(self._stage[lo:hi], op, group)

    def _unstage(self):
        """after the exchanges have been waited for: the averaged bf16 ranges go back into the fp32 arena"""
        for lo, hi in self._staged:
            self._convert(self._stage[lo:hi], self.flat[lo:hi])
        self._staged = []

    # ------------------------------------------------------------------ all-reduce overlapped with the rest of the backward pass
    def tail_offset(self, late_params):
        """`late_params`: the parameters whose gradients are final after the FIRST half of a split backward pass (the decoder
        side of the net: its backward runs first).  They must form the tail of the arena; returns the element offset where
        that tail starts."""
        late = {id(p) for p in late_params if p.requires_grad}
        flags = [id(p) in late for p in self.params]
        first = flags.index(True) if True in flags else len(flags)
        if not all(flags[first:]) or any(flags[:first]):
            raise ValueError("the late parameters are not a contiguous tail of the arena (parameter registration order changed?)")
        return self._offs[first] if first < len(flags) else self._size

    def param_range(self, params):
        """(lo, hi) element range of `params` in the flat arena; they must be adjacent (registration order)"""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)) or len(idx) != len(ids):
            raise ValueError("the parameters are not adjacent in the arena")
        hi = self._offs[idx[-1] + 1] if idx[-1] + 1 < len(self._offs) else self._size
        return self._offs[idx[0]], hi

    def allreduce_begin(self, lo, hi, group=None, piece=None):
        """start the sum all-reduce of flat[lo:hi] (pieces of `piece` elements, last first) behind everything already queued on the
        current stream; returns the work handles.  RCCL runs them on its own stream: kernels launched afterwards overlap."""
        import torch.distributed as dist
        if self._single(group):
            return []                   # a one-rank group: the local sums ARE the mean (RCCL would run a 249 MB copy kernel per step)
        piece = piece or (hi - lo)      # the whole range is final when this is called: one collective (each costs a ring latency)
        works, e = [], hi
        while e > lo:
            b = max(lo, e - piece)
            works.append(self._reduce_range(b, e, group))
            e = b
        return works

    def _flags_to_device(self):
        """this rank's "used" flags of the step in `used_dev`, without a host staging buffer that a later step could overwrite while the copy
        is still queued: each distinct pattern (which modalities were in the batch) is uploaded once and then copied device-to-device"""
        key = bytes(1 if p._miseg_used else 0 for p in self.params)
        cache = self.__dict__.setdefault("_flag_cache", {})
        dev = cache.get(key)
        if dev is None:
            # grow-only: a captured step records this copy with the pattern tensor's raw address as its source - evicting an entry would
            # free memory live hipGraphs still read.  One entry per present-modality set: 2^num_styles at most in practice
            dev = cache[key] = torch.tensor(list(key), dtype=torch.int32, device=self.used_dev.device)
        self.used_dev.copy_(dev, non_blocking=True)

    def used_begin(self, group=None, host=True):
        """start the max all-reduce of the "used on this rank" bitmap (call once the flags of this step are known: for a replayed
        hipGraph that is before the replay); hand the result - a callable that waits for the exchange and returns the flags -
        to allreduce_end.
        host=False: the exchanged flags stay on the device (`used_dev`, what training/optim.py::ArenaOptimizer consumes: "unused on every
        rank => no update, no decay, no step count") and the callable returns None after making the current STREAM wait for the exchange -
        the step has no host synchronisation at all, so the host keeps launching the next step's graphs while this one runs (with the
        host read every step ended in a stream drain and the next graph launch started on an idle card)."""
        import torch.distributed as dist
        if not self.used_dev.is_cuda:
            self.used_dev.copy_(torch.tensor([int(p._miseg_used) for p in self.params], dtype=torch.int32))
            work = self._all_reduce(self.used_dev, dist.ReduceOp.MAX, group)

            def finish():
                work.wait()
                return self.used_dev.tolist() if host else None
            return finish
        single = self._single(group)                      # nothing to exchange: the local flags are the global ones
        if not host:
            self._flags_to_device()
            work = None if single else self._all_reduce(self.used_dev, dist.ReduceOp.MAX, group)

            def finish_dev():
                if work is not None:
                    work.wait()                          # the current stream waits for RCCL's; the host does not
                return None
            return finish_dev
        if single:
            flags = [int(p._miseg_used) for p in self.params]
            return lambda: flags
        # On the card the exchange and the read-back run on a stream of their own, through pinned host buffers: a `.tolist()` on the
        # compute stream would queue behind the whole step and stall the host until the step has drained - the launch of the next
        # step's graphs would then start on an idle card (+0.75 ms per step measured with one rank)
        if self._bm is None:
            n = len(self.params)
            self._bm = (torch.cuda.Stream(device=self.used_dev.device), torch.empty(n, dtype=torch.int32).pin_memory(), torch.cuda.Event())
        side, dst, done = self._bm
        self._flags_to_device()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            work = self._all_reduce(self.used_dev, dist.ReduceOp.MAX, group)
            work.wait()                                  # the side stream waits for RCCL's
            dst.copy_(self.used_dev, non_blocking=True)  # (the host reads `dst` in finish(), before any later step can queue another copy)
            done.record()

        def finish():
            done.synchronize()
            return dst.tolist()
        return finish

    def allreduce_end(self, works, world_size, group=None, rest=None, used_work=None, host_flags=True):
        """finish a split exchange: the "used on any rank" bitmap (unless used_begin already started it), then the all-reduce of
        flat[rest[0]:rest[1]] (the part the second half of the backward pass produced), wait for everything, mean, publish.  The
        bitmap goes first so that the host read of it (the one host sync of a step, like torch DDP's find_unused_parameters)
        completes while the rest is still in flight.
        host_flags=False (with used_begin(host=False)): no host read - the global flags are in `used_dev` for the fused optimiser, `p.grad`
        follows this rank's own flags (a parameter unused here but used elsewhere still holds the averaged gradient in its arena slot)."""
        self.end_backward()
        ub = used_work if used_work is not None else self.used_begin(group, host=host_flags)
        works = list(works)
        if rest is not None:
            for lo, hi in ([rest] if isinstance(rest[0], int) else rest):      # one range or several (a hole left in the tail, see bench.py)
                if hi > lo:
                    works += self.allreduce_begin(lo, hi, group)
        self.allreduce_finish(ub, world_size, group, works)

    def allreduce_finish(self, used_work, world_size, group=None, works=()):
        """the tail of an exchange whose collectives have been issued (allreduce_end; or recorded into the step's hipGraph - GraphedStep
        fused_comm - and replayed): wait for the bitmap and the given works, bf16 staging back to fp32, mean where the backend only sums,
        "used" flags, publish"""
        used = used_work()
        for w in works:
            w.wait()
        self._unstage()
        if not self._avg(group):
            self.flat.mul_(1.0 / world_size)
        if used is not None:
            for p, u in zip(self.params, used):
                p._miseg_used = bool(u)
        self.used_on_device = used is None
        self.publish()

    @staticmethod
    def _avg(group=None):
        """RCCL averages in the collective (ReduceOp.AVG); gloo (CPU tests) sums and the arena is scaled afterwards"""
        import torch.distributed as dist
        return dist.get_backend(group) == "nccl"

    def detach(self):
        self.queues = None
        ops.QUEUES.pop(self._qkey, None)
        ops.join_wgrad()
        ops.WGRAD_STREAM = None
        for p in self.params:
            for a in ("_miseg_grad", "_miseg_arena", "_miseg_used"):
                if hasattr(p, a):
                    delattr(p, a)
