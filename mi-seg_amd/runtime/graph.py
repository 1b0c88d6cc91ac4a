"""hipGraph capture of one whole training step (forward + backward of the drop-in model).

The step is ~1000 short kernel launches; replaying it as one graph removes the host launch path from the critical
path (no tracing compiler involved: the graph is recorded from the very same C-ABI calls).  One graph is captured per
set of modalities present in the batch, because which conditional-norm rows receive a gradient (and which keep
``grad is None``) is part of the recorded work; the per-sample style ids themselves live in a static device tensor
that is overwritten before each replay.
"""
import gc
import os
from typing import Dict, List, Sequence, Tuple

import torch

from ..hip import ops


def _graph_capture(g, **kw):
    """torch.cuda.graph(g, ...) for this package.  With a process group alive the capture is THREAD-LOCAL: ProcessGroupNCCL's watchdog thread
    polls the events of earlier, eager collectives (hipEventQuery), which a capture in the default "global" error mode forbids to every thread
    of the process - the watchdog then dies with "operation not permitted when stream is capturing" and takes the process with it (seen once in
    six runs of tests/rccl_child.py, depending on whether the watchdog had already reaped the works)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        kw.setdefault("capture_error_mode", "thread_local")
    return _Capture(g, kw)


class _Capture:
    """`with _graph_capture(g):` - torch.cuda.graph around a _Graph, then the split plan.

    Python's cyclic collector is held off for the length of the capture: torch.cuda.graph collects once on entry, but a step is ~10^4 Python
    allocations and an automatic collection in the middle of it runs the destructors of whatever cycles the step itself has closed by then
    (autograd contexts holding events / streams) while the stream is capturing - the process was seen to abort() inside such a collection
    (gpurun_out/final/gpu_tests.log of round 5: "Garbage-collecting" under GraphedTrainStep._capture, once in two full suites)."""

    def __init__(self, g, kw):
        self.g = g
        self.ctx = torch.cuda.graph(g.g, **kw)
        self.gc_was_on = False

    def __enter__(self):
        r = self.ctx.__enter__()
        self.gc_was_on = gc.isenabled()
        gc.disable()
        return r

    def __exit__(self, *exc):
        try:
            r = self.ctx.__exit__(*exc)
        finally:
            if self.gc_was_on:
                gc.enable()
        if exc[0] is None:
            self.g.finish()
        return r


# replay multi-stream captures as single-stream pieces (csrc/graphsplit.cpp): OFF by default - correct, and 1.6 x faster than the runtime's
# replay on chains of tiny kernels beside a side stream (scripts/debug/graph_split_probe.py), but 2 - 3 % slower on the training step
# (149 against 153 patches/s)
SPLIT_REPLAY = os.environ.get("MISEG_GRAPH_SPLIT", "0") == "1"
SPLIT_SIDE_STREAMS = int(os.environ.get("MISEG_GRAPH_SPLIT_STREAMS", "2"))


class _Graph:
    """one captured graph of this package: a torch.cuda.CUDAGraph that is replayed either by torch (`replay()` = hipGraphLaunch of the whole
    graph) or - when the capture holds more than one chain of nodes, i.e. the model's side branch or RCCL's stream took part - as the
    single-stream pieces of csrc/graphsplit.cpp (`miseg_graph_split_*`, include/miseg_hip.h; MISEG_GRAPH_SPLIT=1).  Torch's replay is the
    default: see SPLIT_REPLAY."""

    def __init__(self):
        self.g = torch.cuda.CUDAGraph(keep_graph=True) if SPLIT_REPLAY else torch.cuda.CUDAGraph()
        self.plan = None
        self.info = None

    def pool(self):
        return self.g.pool()

    def finish(self):
        """after the capture: build the split plan where it pays"""
        if not SPLIT_REPLAY:
            return
        import ctypes as C
        from ..hip import lib as L
        lib = L.load()
        plan, info = C.c_void_p(), L.GraphSplitInfo()
        rc = lib.miseg_graph_split_create(C.c_void_p(self.g.raw_cuda_graph()), C.c_void_p(torch.cuda.current_stream().cuda_stream), SPLIT_SIDE_STREAMS, C.byref(plan),
                                          C.byref(info))
        if rc == -2:          # MISEG_E_UNSUPPORTED: more concurrent chains than side streams allowed: torch's replay
            self.info = {"unsupported": lib.miseg_last_error().decode()}
            self.g.instantiate()
            return
        L.check(rc, "miseg_graph_split_create")
        self.info = {k: getattr(info, k) for k, _ in L.GraphSplitInfo._fields_}
        if self.info["lanes"] > 1:
            self.plan = plan
            self._launch = lib.miseg_graph_split_launch
        else:                               # one chain: the runtime's own replay is the fast path already
            lib.miseg_graph_split_destroy(plan)
            self.g.instantiate()

    def replay(self):
        if self.plan is None:
            self.g.replay()
            return
        rc = self._launch(self.plan, torch.cuda.current_stream().cuda_stream)
        if rc:
            from ..hip import lib as L
            L.check(rc, "miseg_graph_split_launch")

    def __del__(self):
        if getattr(self, "plan", None) is not None:
            try:
                from ..hip import lib as L
                L.load().miseg_graph_split_destroy(self.plan)
            except Exception:
                pass
            self.plan = None


def _check_styles(model, modalities, batch):
    """host tuple of style ids, range-checked BEFORE it reaches the static device tensor a captured graph reads (the kernels index
    by-value argument arrays with it: an id outside [0, num_styles) would be a wild pointer)"""
    from ..networks.norms.conditional_instance_norm import _check_range, styles_limit
    host = tuple(int(m) for m in modalities)
    if len(host) != batch:
        raise ValueError("Expected number of styles as batch size.")
    return _check_range(host, styles_limit(model))


class GraphedForward:
    """hipGraph of the inference forward pass (no autograd tape) for a fixed batch shape: sliding-window validation runs 700 windows per
    volume through the same ~270 launches, so the host launch path is the cost to remove.  One graph per set of modalities present;
    use as the `predictor` of training/inferer.py::sliding_window_inference (returns a static logits buffer: consume it before the
    next call)."""

    def __init__(self, model: torch.nn.Module, batch_shape: Sequence[int], warmup: int = 1, arena=None):
        """arena: optional runtime.arena.ParamArena of the model.  Without one every forward re-casts and re-packs every weight (66 of
        the ~240 launches of a C-Swin-UNETR forward); with one the copies are refreshed ONCE (`arena.refresh_weights()`, here after the
        warm-up that registers them) and the graph reads them - call `arena.refresh_weights()` again whenever the parameters changed."""
        self.model = model
        self.arena = arena
        dev = next(model.parameters()).device
        self.x = torch.zeros(*batch_shape, dtype=torch.float32, device=dev)
        self.styles = torch.zeros(batch_shape[0], dtype=torch.int32, device=dev)
        self._styles_host = None
        self.warmup = warmup
        self.graphs = {}

    def _run(self, host):
        with torch.no_grad():
            ops.begin_step()
            return self.model(self.x, (self.styles, host))

    def __call__(self, x, modalities=None):
        if tuple(x.shape) != tuple(self.x.shape):
            with torch.no_grad():                       # a tail batch of another size: plain launches
                return self.model(x, modalities)
        host = _check_styles(self.model, modalities, self.x.shape[0]) if modalities is not None else None
        self.x.copy_(x, non_blocking=True)
        if host is not None and host != self._styles_host:
            self.styles.copy_(torch.tensor(host, dtype=torch.int32), non_blocking=True)
            self._styles_host = host
        if self.arena is not None and self.graphs:
            # the graphs read the arena's copies: bring them up to date before EVERY replay.  Unconditional: torch-level updates bump
            # Tensor._version (seen by params_changed()), the fused optimiser writes through raw pointers and bumps only the DEVICE-side
            # version - the versioned refresh kernels see both and are a device-side no-op (two words read) when nothing changed
            self.arena.refresh_weights()
        key = None if host is None else tuple(sorted(set(host)))
        if key not in self.graphs:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(max(self.warmup, 1 if self.arena is not None else 0)):
                    self._run(host)
                if self.arena is not None:
                    self.arena.refresh_weights()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = _Graph()
            with _graph_capture(g):
                y = self._run(host)
            ops.STAT_POOL.pin()
            self.graphs[key] = (g, y)
        g, y = self.graphs[key]
        g.replay()
        return y


class GraphedStep:
    def __init__(self, model: torch.nn.Module, batch_shape: Sequence[int], out_shape: Sequence[int], warmup: int = 2, arena=None, split=False, fused_comm=None,
                 first_flush="all"):
        """arena: optional runtime.arena.ParamArena of the model (gradients accumulate in its flat buffer, parameter
        re-layouts are refreshed by one kernel per step); without it every gradient is a tensor of the graph's pool.
        Capture needs a quiescent model: drop every reference to outputs of earlier eager steps first - a live autograd graph keeps its
        AccumulateGrad nodes on the eager stream, torch then inserts a cross-stream wait into the capture and hipStreamEndCapture crashes.
        split (needs an arena and a model with `forward(..., cut=)`): the step is recorded as TWO graphs - forward + the decoder
        side of the backward pass, then the encoder side - and `__call__(..., between=fn)` runs `fn()` between the two replays:
        the data-parallel step starts the all-reduce of the decoder-side gradients there, so RCCL overlaps the second graph.
        fused_comm (round 4; with split): an object with `early() -> works` and `late(works)`.  The two halves are then recorded into ONE
        graph with `early()` called - under capture - between them and `late(works)` after the second: the collectives they issue
        (ProcessGroupNCCL joins RCCL's stream to the capture with ordinary events; scripts/debug/rccl_capture_probe.py) become nodes of the
        step's graph, the host launches one graph per step and `between` is not used.  The warm-up runs in front of the capture issue
        NO collective (and `late()` should exchange the "used" bitmap under capture as well): a step with captured collectives is best left
        without eager ones - ProcessGroupNCCL's watchdog thread polling events around a capture that records into RCCL's stream killed 1 run in 8.
        fused_comm WITHOUT split: the backward pass is not cut at all - the model calls back once its decoder side has been back-propagated
        (`forward(..., on_decoder_done=)`), the main stream's queued launches go out there and `early()` is called; the side branch starts
        where it always starts.  `late(works)` follows the end of the pass."""
        self.model = model
        self.arena = arena
        self.split = bool(split)
        self.fused_comm = fused_comm
        self._warming = False
        self.first_flush = first_flush          # split step: "all" queued launches at the end of the first half, or "small" (arena.flush_small)
        if self.split and arena is None:
            raise ValueError("a split step needs a ParamArena")
        if fused_comm is not None and arena is None:
            raise ValueError("fused_comm needs a ParamArena")
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        self.x = torch.zeros(*batch_shape, dtype=torch.float32, device=dev)
        self.cot = torch.zeros(*out_shape, dtype=torch.float32, device=dev)
        self.styles = torch.zeros(batch_shape[0], dtype=torch.int32, device=dev)
        self._styles_host = None
        self.warmup = warmup
        self.graphs: Dict[Tuple[int, ...], Tuple[torch.cuda.CUDAGraph, List, torch.Tensor]] = {}

    def _run(self, host):
        if self.arena is not None:
            self.arena.begin_step()
        else:
            for p in self.params:
                p.grad = None
            ops.begin_step()
        ops.stamp("step_begin")
        works = []
        if self.fused_comm is not None:
            def early():
                ops.stamp("decoder_done")
                # the main stream's queued launches so far = the decoder side's: they complete the gradients that `early()` sends.  Inline on
                # the main stream: on a stream of their own (a third chain beside main and the side branch) the captured step replayed 23 %
                # slower (151 -> 117 patches/s) - the hipGraph executor serialises a third chain, as in round 3
                if getattr(self.fused_comm, "flush", "all") == "small":
                    # the early range holds only layers whose conv weight gradients are written inline (the tiny-volume kernel): the small
                    # queued launches complete its other parameters, the grouped conv weight gradients stay at the end of the pass
                    self.arena.queues.flush_small()
                else:
                    self.arena.queues.flush(side=False)
                if not self._warming:          # (the warm-up runs issue no collective: __init__)
                    works.extend(self.fused_comm.early())
            y = self.model(self.x, (self.styles, host), on_decoder_done=early)
        else:
            y = self.model(self.x, (self.styles, host))
        ops.stamp("forward_end")
        y.backward(self.cot)
        if self.arena is not None:
            self.arena.end_backward()
        else:
            ops.join_branch()          # a capture must not end with a model's side branch unjoined
        if self.fused_comm is not None and not self._warming:
            self.fused_comm.late(works)
        ops.stamp("step_end")
        return y

    def _run_first(self, host):
        """forward + the decoder-side half of the backward pass; everything queued so far is issued, so the late parameters'
        arena slots are final when this returns"""
        self.arena.begin_step()
        ops.stamp("step_begin")
        cut = []
        y = self.model(self.x, (self.styles, host), cut=cut)
        ops.stamp("forward_end")
        y.backward(self.cot)
        ops.stamp("first_half_chain_end")
        if self.first_flush == "small":
            self.arena.flush_small()
        else:
            self.arena.flush()
        ops.stamp("first_half_end")
        return y, cut

    def _run_second(self, cut):
        ops.stamp("second_half_begin")
        torch.autograd.backward([o for o, _ in cut], [l.grad for _, l in cut])
        self.arena.end_backward()
        ops.stamp("step_end")

    def _capture(self, host):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        self._warming = True          # (no collective in the warm-up runs of a step with captured collectives: see __init__)
        try:
            with torch.cuda.stream(s):
                for _ in range(max(self.warmup, 2 if self.arena is not None else 0)):   # arena: step 1 registers the re-layouts
                    if self.split:
                        self._run_second(self._run_first(host)[1])
                    else:
                        self._run(host)
                if self.fused_comm is not None and self.arena is not None and self.arena.used_dev.is_cuda:
                    self.arena._flags_to_device()      # this graph's "used" pattern on the device now: the captured exchange finds it cached (no upload under capture)
        finally:
            self._warming = False
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for p in self.params:
            p.grad = None
        g = _Graph()
        g2 = None
        if self.split and self.fused_comm is not None:
            with _graph_capture(g):
                y, cut = self._run_first(host)
                works = self.fused_comm.early()
                self._run_second(cut)
                self.fused_comm.late(works)
            del cut
        elif self.split:
            with _graph_capture(g):
                y, cut = self._run_first(host)
            g2 = _Graph()
            with _graph_capture(g2, pool=g.pool()):        # the second graph consumes activations the first one saved: one pool
                self._run_second(cut)
            del cut
        else:
            with _graph_capture(g):
                y = self._run(host)
        ops.STAT_POOL.pin()        # the graph holds raw pointers into the statistics chunk it was captured on
        if self.arena is not None:
            grads = [bool(p._miseg_used) for p in self.arena.params]       # which slots this graph writes
        else:
            grads = [p.grad for p in self.params]
        return (g, g2), grads, y

    def __call__(self, x, modalities: Sequence[int], cot, between=None, publish=True, before=None):
        """x [B,C,D,H,W] fp32, cotangent d(loss)/d(logits); returns logits (static buffer) with p.grad populated.
        between: called between the two replays of a split step; publish=False leaves `p.grad` to the caller
        (arena.allreduce_end does it after the exchange).  before: called right before the (first) replay, once this step's "used" flags
        are set on the parameters - where a step with captured collectives starts its bitmap exchange."""
        host = _check_styles(self.model, modalities, self.x.shape[0])
        self.x.copy_(x, non_blocking=True)
        if cot.data_ptr() != self.cot.data_ptr():      # a caller that writes d(loss)/d(logits) into `self.cot` itself (the fused loss) skips the copy
            self.cot.copy_(cot, non_blocking=True)
        if host != self._styles_host:                  # (a pageable host tensor per step was a synchronous staging copy)
            self.styles.copy_(torch.tensor(host, dtype=torch.int32), non_blocking=True)
            self._styles_host = tuple(host)
        if self.arena is not None:
            self.arena.params_changed()           # torch-level parameter updates since the last step bump the device-side version the graph reads
        key = tuple(sorted(set(host))) + (len(host),)
        if key not in self.graphs:
            self.graphs[key] = self._capture(host)
        (g, g2), grads, y = self.graphs[key]
        if self.arena is not None:
            for p, used in zip(self.arena.params, grads):      # static per graph: known before the replay (the split step's hook
                p._miseg_used = used                           # exchanges the "used" bitmap early)
        if before is not None:
            before()
        g.replay()
        if g2 is not None:
            if between is not None:
                between()
            g2.replay()
        if self.arena is not None:
            if publish:
                self.arena.publish()
        else:
            for p, gr in zip(self.params, grads):
                p.grad = gr
        return y


class GraphedTrainStep:
    """hipGraph of one whole OPTIMISATION step: versioned weight refresh + forward + fused segmentation loss + backward + the one-launch
    optimiser - what the reference runs per iteration (LitMonai.training_step + optimizer.step, networks/lightning_monai.py:149-166, 255-278)
    as one replay, with no host work between its kernels.

    The optimiser kernel bumps the arena's device-side parameter version, so the refresh launches at the head of the NEXT replay re-cast
    and re-pack every weight (runtime/arena.py: versioned refresh) - the cost a real training loop pays and the forward + backward
    benchmark does not.  The learning rate lives in a device scalar (`set_lr`), so schedulers work under replay.  One graph per set of
    modalities present in the batch (which conditional-norm rows get a gradient, and are touched by the optimiser, is recorded work)."""

    def __init__(self, model, criterion, optimizer, batch_shape, label_shape, arena, label_dtype=torch.int32, warmup=2):
        if arena is None or optimizer.arena is not arena:
            raise ValueError("GraphedTrainStep needs the model's ParamArena and an ArenaOptimizer built over it")
        self.model, self.criterion, self.opt, self.arena = model, criterion, optimizer, arena
        dev = arena.flat.device
        self.x = torch.zeros(*batch_shape, dtype=torch.float32, device=dev)
        self.label = torch.zeros(*label_shape, dtype=label_dtype, device=dev)
        self.styles = torch.zeros(batch_shape[0], dtype=torch.int32, device=dev)
        self.one = torch.ones((), dtype=torch.float32, device=dev)
        self._styles_host = None
        self.warmup = warmup
        self.graphs = {}
        # opt-in (MISEG_EARLY_OPT=1): measured SLOWER on the headline net - 138.1 / 137.3 patches/s against 140.1 / 139.1 with the one-launch
        # step behind the pass (same box): 1.2 GB of optimiser traffic beside the end-of-pass grouped launches lengthens the tail by more than
        # the 0.28 ms it takes off the chain (the same finding as the tiny-volume weight gradients on the branch, hip/ops.py::TINY_WGRAD_AT)
        self.early_optimizer = bool(os.environ.get("MISEG_EARLY_OPT"))
        if optimizer.lr_dev is None:
            optimizer.lr_dev = torch.tensor([optimizer.lr], dtype=torch.float32, device=dev)

    def set_lr(self, lr):
        self.opt.lr = float(lr)
        self.opt.lr_dev.fill_(float(lr))

    def _run(self, host, optimise):
        # round 5, opt-in (self.early_optimizer): the parameters whose gradients are final when the main chain of the backward pass ends (the
        # tiny-volume conv weights of encoder10 / decoder5: 70 % of the headline net's bytes) are updated on the branch stream beside the
        # end-of-pass grouped launches (ops.join_branch -> branch_end_hook), the rest behind the pass: two launches over disjoint tables
        self.arena.branch_end_hook = (lambda: self.opt.step_early()) if (optimise and self.early_optimizer) else None
        try:
            self.arena.begin_step()
            logits = self.model(self.x, (self.styles, host))
            loss = self.criterion(logits, self.label)
            loss.backward(self.one)
            self.arena.end_backward()
        finally:
            self.arena.branch_end_hook = None
        if optimise:
            self.opt.step(update_flags=False)      # the flags of this graph are static: copied to the device before every replay
        return loss.detach()

    def _capture(self, host):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(self.warmup, 2)):      # step 1 registers the weight re-layouts; no optimiser step: weights and moments stay put
                self._run(host, False)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self.early_optimizer:      # which parameters the warm-up steps wrote inline (none: the one-launch optimiser step)
            self.opt.split_early(getattr(self.arena, "inline_final_params", []))
        self.opt.prepare()            # device tables of the fused conv-weight launch: uploaded here, not under capture
        for p in self.arena.params:
            p.grad = None
        g = _Graph()
        with _graph_capture(g):
            loss = self._run(host, True)
        ops.STAT_POOL.pin()
        return g, [bool(p._miseg_used) for p in self.arena.params], loss

    def __call__(self, x, label, modalities):
        """one optimisation step on (x [B,C,D,H,W] fp32, label [B,1,D,H,W] class ids); returns the loss (0-dim device tensor, static buffer)"""
        host = _check_styles(self.model, modalities, self.x.shape[0])
        self.x.copy_(x, non_blocking=True)
        self.label.copy_(label, non_blocking=True)
        if host != self._styles_host:
            self.styles.copy_(torch.tensor(host, dtype=torch.int32), non_blocking=True)
            self._styles_host = tuple(host)
        self.arena.params_changed()
        key = tuple(sorted(set(host))) + (len(host),)
        if key not in self.graphs:
            self.graphs[key] = self._capture(host)
        g, used, loss = self.graphs[key]
        for p, u in zip(self.arena.params, used):
            p._miseg_used = u
        self.opt.set_used_from_arena()
        g.replay()
        self.arena.epoch += 1          # (host-side mirror of the device's parameter version: eager forwards cast per call until the next refresh)
        for p, v, u in zip(self.arena.params, self.arena.views, used):
            p.grad = v if u else None
        return loss
