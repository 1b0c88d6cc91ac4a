"""Optimiser step of the whole model as ONE kernel launch over the gradient arena.

The reference builds torch.optim.AdamW / Adam / SGD(nesterov) over `self.parameters()` (networks/lightning_monai.py:255-278); torch
skips a parameter whose `grad is None` completely -- no update, no weight decay, no step count -- which is what happens every step to the
conditional-norm rows of the modality that is absent from the batch.  `ArenaOptimizer` keeps that rule: the "used" flags of
runtime/arena.py::ParamArena (set by the weight-gradient kernels' autograd nodes, max-reduced over ranks under DDP) travel to the
device and the kernel leaves unused tensors alone, with a per-parameter step count for Adam's bias correction.  The moments live in two
flat fp32 buffers laid out like the arena."""
import ctypes as C
import os

import torch

from ..hip import lib as L
from ..hip import ops

KINDS = {"adamw": L.OPT_ADAMW, "adam": L.OPT_ADAM, "sgd": L.OPT_SGD_NESTEROV}
FUSE_CONV_PACKS = os.environ.get("MISEG_NO_OPT_PACK") is None      # A/B switch of round 5 (ArenaOptimizer._fused_tables)


class ArenaOptimizer:
    def __init__(self, arena, kind="adamw", lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, momentum=0.99):
        if kind not in KINDS:
            raise ValueError("Optimization {} not implemented, please chose another optimizer.".format(kind))
        self.arena, self.kind = arena, kind
        self.lr, self.betas, self.eps, self.weight_decay, self.momentum = float(lr), betas, float(eps), float(weight_decay), float(momentum)
        dev = arena.flat.device
        self.state1 = torch.zeros_like(arena.flat)
        self.state2 = torch.zeros_like(arena.flat) if kind != "sgd" else None
        n = len(arena.params)
        self.steps = torch.zeros(n, dtype=torch.int32, device=dev)
        self.used = torch.ones(n, dtype=torch.int32, device=dev)
        self._used_cache = {}      # bytes of a step's "used" flags -> the same flags on the device (one entry per present-modality set)
        descs = (L.OptDesc * n)()
        block0 = 0
        for i, (p, off) in enumerate(zip(arena.params, arena._offs)):
            if not p.is_contiguous():
                raise ValueError("parameters must be contiguous float32")
            descs[i] = L.OptDesc(p.data_ptr(), off, p.numel(), block0)
            block0 += (p.numel() + L.OPT_BLOCK - 1) // L.OPT_BLOCK
        self._ptrs = [p.data_ptr() for p in arena.params]
        self._table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        self._blocks = block0
        self.lr_dev = None
        self._early = None          # (table, index, blocks, n) of the parameters updated by step_early(), and of the rest
        self._late = None
        self._early_done = False

    def _subset_table(self, sel):
        """(descriptor table, index, blocks, count) of the arena parameters `sel` (a list of indices) on the device"""
        dev = self.state1.device
        descs, index, block0 = (L.OptDesc * len(sel))(), [], 0
        for j, i in enumerate(sel):
            p = self.arena.params[i]
            descs[j] = L.OptDesc(p.data_ptr(), self.arena._offs[i], p.numel(), block0)
            block0 += (p.numel() + L.OPT_BLOCK - 1) // L.OPT_BLOCK
            index.append(i)
        return (torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev), torch.tensor(index, dtype=torch.int32, device=dev), block0, len(sel))

    def _fused_tables(self):
        """Round 5: the 3x3x3 conv weights (55 of C-Swin-UNETR's 62 M parameters) are updated by the launch that also writes their packs
        (miseg_opt_step_pack_conv3), everything else by the element-wise launch in front of it.  Returns (rest table | None, map, pack table) once
        the arena holds a current conv-pack table whose weights are all trainable arena parameters, else None (the plain one-launch step)."""
        a = self.arena
        if not FUSE_CONV_PACKS or self._early is not None or not a.flat.is_cuda or a._ptable is None or a._pdirty or a.dtype not in (torch.float32, torch.bfloat16):
            return None
        key = a._ptable[0].data_ptr()
        cur = self.__dict__.get("_fused")
        if cur is not None and cur[0] == key:
            return cur[1]
        if torch.cuda.is_current_stream_capturing():
            return None                                  # (tables are uploaded eagerly: the warm-up steps in front of a capture build them)
        idx_of = {id(p): i for i, p in enumerate(a.params)}
        maps, conv = (L.OptPackMap * len(a._packs))(), set()
        for j, ent in enumerate(a._packs.values()):
            i = idx_of.get(id(ent[0]))
            if i is None or not ent[4]:                  # a frozen weight has packs but no arena slot: the plain step handles such models
                self._fused = (key, None)
                return None
            maps[j] = L.OptPackMap(a._offs[i], i, 0)
            conv.add(i)
        rest = [i for i in range(len(a.params)) if i not in conv]
        tabs = (self._subset_table(rest) if rest else None, torch.frombuffer(bytearray(bytes(maps)), dtype=torch.uint8).to(self.state1.device), a._ptable)
        self.__dict__.setdefault("_retired", []).append(cur)      # a captured hipGraph may still point at the old tables
        self._fused = (key, tabs)
        return tabs

    def prepare(self):
        """build (and upload) the launch tables now - call before capturing a step that contains `step()` into a hipGraph"""
        self._fused_tables()

    def _launch_pack(self, tabs, lr, count_n):
        _, maps, (ptab, n, tiles) = tabs
        a = self.arena
        p = L.OptStep(C.sizeof(L.OptStep), KINDS[self.kind], None, 0, 0, a.flat.data_ptr(),
                      self.state1.data_ptr(), self.state2.data_ptr() if self.state2 is not None else None, self.used.data_ptr(), self.steps.data_ptr(),
                      self.lr if lr is None else float(lr), self.betas[0], self.betas[1], self.eps, self.weight_decay, self.momentum,
                      self.lr_dev.data_ptr() if self.lr_dev is not None else None, a.params_version_ptr(), None, count_n)
        state = a._ver(3)[1]
        L.check(L.load().miseg_opt_step_pack_conv3(C.byref(p), C.c_void_p(ptab.data_ptr()), C.c_void_p(maps.data_ptr()), n, tiles,
                                                   L.F32 if a.dtype == torch.float32 else L.BF16, state, ops._stream()), "opt_step_pack_conv3")

    def split_early(self, early_params):
        """Two launches per step: `early_params` (parameters whose gradients are final before the end of the backward pass: what
        ParamArena.inline_final_params reports after a step) are updated by step_early() - which the caller may issue on a side stream beside the
        rest of the pass - and all the others by step().  Same arithmetic, same result as the one-launch step (each parameter is touched once)."""
        ids = {id(p) for p in early_params}
        # (tables a captured hipGraph points at by raw address must never be freed: an unchanged set keeps its tables - a second capture, of
        # another modality set, asks again - and replaced ones are retired, not dropped)
        if getattr(self, "_early_ids", None) == ids:
            return
        dev = self.state1.device

        def table(sel):
            descs, index, block0 = (L.OptDesc * len(sel))(), [], 0
            for j, i in enumerate(sel):
                p = self.arena.params[i]
                descs[j] = L.OptDesc(p.data_ptr(), self.arena._offs[i], p.numel(), block0)
                block0 += (p.numel() + L.OPT_BLOCK - 1) // L.OPT_BLOCK
                index.append(i)
            return (torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev), torch.tensor(index, dtype=torch.int32, device=dev), block0, len(sel))
        n = len(self.arena.params)
        early = [i for i in range(n) if id(self.arena.params[i]) in ids]
        late = [i for i in range(n) if id(self.arena.params[i]) not in ids]
        if len(early) != len(ids) or (ids and not late):
            raise ValueError("split_early: parameters that are not in the arena (or nothing left for the second launch)")
        new = (table(early), table(late)) if ids else (None, None)
        self.__dict__.setdefault("_retired", []).append((self._early, self._late))
        self._early, self._late = new
        self._early_ids = ids

    def _launch(self, tab, lr, count_n):
        table, index, blocks, n = tab
        p = L.OptStep(C.sizeof(L.OptStep), KINDS[self.kind], table.data_ptr(), n, blocks, self.arena.flat.data_ptr(),
                      self.state1.data_ptr(), self.state2.data_ptr() if self.state2 is not None else None, self.used.data_ptr(), self.steps.data_ptr(),
                      self.lr if lr is None else float(lr), self.betas[0], self.betas[1], self.eps, self.weight_decay, self.momentum,
                      self.lr_dev.data_ptr() if self.lr_dev is not None else None, self.arena.params_version_ptr(),
                      index.data_ptr() if index is not None else None, count_n)
        ops._call("miseg_opt_step", p)

    def step_early(self, lr=None):
        """the first launch of a split step (split_early): the `used` flags must be on the device already (set_used_from_arena / a captured
        step's static flags); step() then updates the rest and closes the step"""
        if self._early is None or self._early_done:
            return
        self._launch(self._early, lr, 0)
        self._early_done = True

    def set_used_from_arena(self):
        """copy the host "used" flags of this step (arena.publish / allreduce have settled them) to the device; under hipGraph replay call
        this BEFORE the replay that contains the step"""
        # The flags of a step are one of a handful of patterns (which modalities were in the batch).  Each pattern is uploaded ONCE from a
        # host buffer of its own and then copied device-to-device on the stream: a single pinned staging buffer rewritten every step raced
        # with its own asynchronous upload once the host ran ahead of the device (sync-free loops), and a step then saw the next step's flags.
        key = bytes(1 if p._miseg_used else 0 for p in self.arena.params)
        dev = self._used_cache.get(key)
        if dev is None:      # (grow-only: a captured train step reads the pattern tensor by its raw address - see ParamArena._flags_to_device)
            dev = self._used_cache[key] = torch.tensor(list(key), dtype=torch.int32, device=self.used.device)
        self.used.copy_(dev, non_blocking=True)

    def step(self, lr=None, update_flags=True):
        if [p.data_ptr() for p in self.arena.params] != self._ptrs:
            raise RuntimeError("a parameter was re-allocated after the optimiser was built (e.g. model.to(...)): rebuild the ArenaOptimizer")
        if update_flags:
            if getattr(self.arena, "used_on_device", False):      # a data-parallel exchange left the GLOBAL flags on the device (no host read)
                self.used.copy_(self.arena.used_dev, non_blocking=True)
            else:
                self.set_used_from_arena()
        fused = self._fused_tables()
        if self._early_done:          # the early launch of this step is out: the rest, then the step counts of ALL parameters and the version bump
            self._launch(self._late, lr, len(self.arena.params))
            self._early_done = False
        elif fused is not None:       # everything but the 3x3x3 conv weights, then those together with their packs (which closes the step)
            if fused[0] is not None:
                self._launch(fused[0], lr, 0)
            self._launch_pack(fused, lr, len(self.arena.params))
        else:
            self._launch((self._table, None, self._blocks, len(self.arena.params)), lr, -1)
        # the compute-dtype copies of the parameters are stale now: the kernel bumped the arena's device-side parameter version (the refresh
        # launches of the next step re-lay-out everything); the host-side epoch moves on too, so that eager forwards cast per call until then
        self.arena.epoch += 1

    def state_dict(self):
        return {"kind": self.kind, "state1": self.state1.clone(), "state2": None if self.state2 is None else self.state2.clone(), "steps": self.steps.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "momentum": self.momentum}

    def load_state_dict(self, sd):
        if sd["kind"] != self.kind or sd["state1"].numel() != self.state1.numel():
            raise ValueError("optimizer state does not match this model / optimiser kind")
        self.state1.copy_(sd["state1"])
        if self.state2 is not None:
            self.state2.copy_(sd["state2"])
        self.steps.copy_(sd["steps"])
