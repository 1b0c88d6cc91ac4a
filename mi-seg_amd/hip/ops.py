"""Tensor-level wrappers over the C ABI (no autograd here).  torch is only the allocator / stream provider.

Activations are "channels-last rows": any tensor whose last dim is the channel dim with stride 1 and whose leading
dims collapse to one uniform row stride (e.g. a contiguous [B, D, H, W, C] tensor or a channel slice of one).
"""
import ctypes as C
import os
import threading

import torch

from . import lib as L


def _dt(t):
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise ValueError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


STAMPS = {} if os.environ.get("MISEG_STEP_STAMPS") in ("1", "2", "3") else None      # measurement aid (bench.py prints them): name -> slot of _STAMP_BUF
STAMPS_FINE = os.environ.get("MISEG_STEP_STAMPS") == "1"      # "2": only the stamps at the ends of the passes (every stamp is a graph node and
                                                              # the nodes around a fork change how the executor cuts the graph into chains)
STAMPS_BRANCH = os.environ.get("MISEG_STEP_STAMPS") == "3"   # "3": also one stamp behind every launch of the side branch's backward pass and of the
                                                              # end-of-pass flush on either stream (name@stream#n): where the tail of the step goes
_STAMP_BUF = None
_FLUSHING = False


def _stamp_launch(fn_name):
    if STAMPS_BRANCH and (_FLUSHING or in_branch_backward()) and len(STAMPS) < 1000:
        on = "b" if (_BRANCH_STREAM is not None and torch.cuda.current_stream() == _BRANCH_STREAM) else "m"
        stamp(f"{fn_name.replace('miseg_', '')}@{on}#{len(STAMPS)}")


def stamp(name, stream=None, fine=False):
    """record the device wall clock when `stream` (default: the current one) gets here (miseg_debug_stamp; capture-safe); off unless
    MISEG_STEP_STAMPS=1.  read_stamps() returns {name: microseconds} of the last run / replay."""
    global _STAMP_BUF
    if STAMPS is None or (fine and not STAMPS_FINE):
        return
    if _STAMP_BUF is None:
        _STAMP_BUF = torch.zeros(1024, dtype=torch.int64, device="cuda")
    slot = STAMPS.setdefault(name, len(STAMPS))
    st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
    L.check(L.load().miseg_debug_stamp(C.c_void_p(_STAMP_BUF.data_ptr() + 8 * slot), st), "debug_stamp")


def read_stamps():
    v = _STAMP_BUF.cpu().tolist()
    return {k: v[i] / 100.0 for k, i in STAMPS.items()}      # 100 MHz


def rows(t):
    """(ld, nrows, C) of a channels-last row view; raises if the leading dims are not uniformly strided."""
    if not t.is_cuda:
        raise L.MisegHipError("miseg ops need CUDA/HIP tensors: the MI355X path has no CPU fallback")
    if t.dim() < 2:
        raise ValueError("need at least [rows, C]")
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        raise ValueError(f"channel dim must be contiguous, got strides {t.stride()}")
    ld = t.stride(-2)
    n = t.shape[-2]
    for i in range(t.dim() - 3, -1, -1):
        if t.shape[i] != 1 and t.stride(i) != ld * n:
            raise ValueError(f"rows are not uniformly strided: shape {tuple(t.shape)} strides {t.stride()}")
        n *= t.shape[i]
    return ld, n, t.shape[-1]


def _fp32(t):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("parameters / gradients must be contiguous float32")
    return t


PROFILE_HOOK = None   # bench.py roofline leg: list collecting (kernel name, flops, algorithmic bytes) per hooked call; its index is the tag
                      # the library's in-situ timing records with every kernel launch of that call (miseg_prof_arm, csrc/common.cpp)


class _ProfRegion:
    """roofline leg only: every kernel the library launches inside the region records its own begin / end timestamps under one tag - once,
    where the step issues it, on its own stream, beside whatever else is running (side branch, grouped launches: the schedule that is timed)"""

    def __init__(self, name, flops=0.0, nbytes=0.0):
        self.on = PROFILE_HOOK is not None
        if self.on:
            PROFILE_HOOK.append((name, float(flops), float(nbytes)))
            self.tag = len(PROFILE_HOOK) - 1

    def __enter__(self):
        if self.on:
            L.load().miseg_prof_arm(self.tag)
        return self

    def __exit__(self, *exc):
        if self.on:
            L.load().miseg_prof_arm(-1)
        return False


def _call(fn_name, params, prof=None, prof_params=None, extra=()):
    lib = L.load()
    if PROFILE_HOOK is not None and prof is not None:
        with _ProfRegion(prof[0], prof[1], prof[2] if len(prof) > 2 else 0.0):
            L.check(getattr(lib, fn_name)(C.byref(params), *extra, _stream()), fn_name)
        return
    L.check(getattr(lib, fn_name)(C.byref(params), *extra, _stream()), fn_name)
    if STAMPS_BRANCH:
        _stamp_launch(fn_name)


def _nb(*tensors):
    """bytes of the given tensors (None skipped): the algorithmic traffic of a launch that touches each of them once"""
    return float(sum(t.numel() * t.element_size() for t in tensors if t is not None))


def _prof_scratch(t):
    """(round 3's roofline leg repeated every launch five times and sent the accumulating outputs of the repeats here; since round 4 a launch
    is timed once, in place: no scratch)"""
    return None


# ------------------------------------------------------------------------------------------ instance norm
class _ZeroPool:
    """fp64 statistics buffers of a step come from one pre-zeroed pool (one fill instead of ~100 small memsets).

    Lifetime rules.  Every statistics tensor handed out is a VIEW of a chunk, so a chunk lives as long as anything (an autograd node's
    saved tensors) still uses it; the pool itself only decides where the NEXT buffers come from:
      * `begin_step()` (a training arena / GraphedStep owns the step and has enqueued the previous backward pass): the current chunk is
        zero-filled in place and reused - captured hipGraphs rely on these fixed addresses, so a chunk that a graph was captured on is
        never dropped (`pin()`; GraphedStep pins what it captured);
      * `fresh()` (a model forward that nobody manages - LitMonai's eager loop, sliding-window validation, unit tests): the pool lets go
        of its chunk and starts a new one sized by the high-water mark, so earlier forwards whose backward is still pending keep their
        statistics, and 700 inference windows do not grow anything;
      * overflow inside a step: a new chunk is started; the old one stays alive through its views and through `_pinned`."""

    def __init__(self, numel=1 << 20):
        self.numel, self.buf, self.off, self.high = numel, None, 0, 0
        self.spilled = 0         # doubles of this step that live in chunks an overflow has left behind
        self._pinned = []
        self._filled = None      # (event recorded behind the zero fill of the current chunk, streams already ordered behind it)

    def take(self, n, device):
        n = (n + 1) & ~1
        if self.buf is None or self.buf.device != device or self.off + n > self.buf.numel():
            capturing = torch.cuda.is_current_stream_capturing()
            if self.buf is not None:
                self.spilled += self.off
                if capturing:
                    # Overflow INSIDE a hipGraph capture: the graph being recorded holds raw pointers into the chunk that is left behind, so it
                    # must never be freed.  Round 3 found this the hard way: a GraphedForward captured as the first GPU work of a process (the
                    # 1 M-double first chunk overflows several times in a batch-4 fs=48 forward) kept only its LAST chunk alive; the others
                    # went back to the allocator, were handed to the window batches of the sliding-window loop, and every replay after the
                    # first normalised with statistics that other tensors had been written over (logits 43 % off, finite, unnoticed by a
                    # finiteness check; tests/test_hip_training.py::test_full_volume_sliding_window_of_the_headline_model).
                    self.pin()
            # the next chunk holds what the WHOLE step has needed so far, not just the chunk that overflowed: one overflow per step at most
            self.high = max(self.high, self.spilled + n)
            self.buf = torch.empty(max(self.numel, n, int(self.high * 1.5)), dtype=torch.float64, device=device)
            lib = L.load()
            L.check(lib.miseg_fill32(_ptr(self.buf), 0, self.buf.numel() * 2, _stream()), "fill32")
            self.off = 0
            self._filled = None
            if not capturing:
                ev = torch.cuda.Event()
                ev.record()
                self._filled = (ev, {torch.cuda.current_stream().cuda_stream})
        elif self._filled is not None and not torch.cuda.is_current_stream_capturing():
            # A chunk is zero-filled on the stream that happened to need it first.  A model's side branch (SwinUNETR / UNETR) takes buffers of
            # the same chunk on ANOTHER stream: ordered behind the fill only if that stream happened to wait for the first one after the
            # fill was queued - true for the chunk a forward starts with, not for one that an overflow starts in the middle of a branch.
            # Every other stream waits for the fill once, and the allocator learns that the chunk is in use there.
            cs = torch.cuda.current_stream()
            if cs.cuda_stream not in self._filled[1]:
                cs.wait_event(self._filled[0])
                self.buf.record_stream(cs)
                self._filled[1].add(cs.cuda_stream)
        t = self.buf[self.off:self.off + n]
        self.off += n
        self.high = max(self.high, self.spilled + self.off)
        return t

    def begin_step(self):
        """call once per training step, after the previous step's backward has been enqueued: recycles the pool."""
        # The WHOLE chunk is zeroed, not the part the previous step used: a chunk an overflow started late in the warm-up step holds only
        # that step's tail ([0, off)), while the step being captured takes its buffers from the chunk's beginning and walks past `off` into
        # memory that was zero when the chunk was created - and never again: every replay after the first added its statistics to the
        # previous replay's (round 3, found with a GraphedForward captured as the first GPU work of a process; see take()).
        if self.buf is not None:
            lib = L.load()
            L.check(lib.miseg_fill32(_ptr(self.buf), 0, self.buf.numel() * 2, _stream()), "fill32")
            self.off = 0
        self.spilled = 0

    def fresh(self):
        """start the next forward on a new chunk (unless the current one is untouched); see the class docstring"""
        if self.buf is not None and self.off > 0 and not torch.cuda.is_current_stream_capturing():
            self.buf, self.off, self._filled = None, 0, None
        if not torch.cuda.is_current_stream_capturing():
            self.spilled = 0

    def pin(self):
        """keep the current chunk alive for good: a captured hipGraph holds raw pointers into it"""
        if self.buf is not None and not any(b is self.buf for b in self._pinned):
            self._pinned.append(self.buf)


DEFAULT_POOL = _ZeroPool()
STAT_POOL = DEFAULT_POOL      # the ACTIVE pool: every ParamArena owns one (round 3) and makes it the active one for its step / its model's forward
                              # (use_pool); models without an arena share DEFAULT_POOL.  Two models in one process no longer recycle each other's chunk.


def use_pool(pool=None):
    global STAT_POOL
    STAT_POOL = pool if pool is not None else DEFAULT_POOL


def begin_step():
    """start of a training step whose previous backward pass has been enqueued.  Under hipGraph capture the current chunk is recycled in
    place (fixed addresses are what a graph replays); eagerly the step simply moves to a fresh chunk, so statistics that some other
    pending autograd graph still holds are never zero-filled under it."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        STAT_POOL.begin_step()
    else:
        STAT_POOL.fresh()
    DROP.advance()
    check_no_pending()


def begin_forward(params=()):
    """called by the nets at the top of `forward`: when no training arena manages the step (its begin_step() recycles the pool), the
    statistics of this forward go to a fresh chunk, so an un-managed loop neither grows the pool nor clobbers statistics that an earlier,
    not yet back-propagated forward still needs."""
    arena = None
    for p in params:
        arena = getattr(p, "_miseg_arena", None)
        break
    use_pool(getattr(arena, "pool", None))
    if torch.cuda.is_current_stream_capturing():
        return
    if torch.is_grad_enabled() and arena is not None:
        return
    STAT_POOL.fresh()
    DROP.advance()


def instnorm_stats(x, B, S):
    """x: rows view of [B*S, C].  Returns stat: float64 [R, B, C, 2]; summed over the R replicas = (sum x, sum x^2)."""
    ld, n, Cc = rows(x)
    assert n == B * S, (n, B, S)
    stat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(B, Cc) // 8, x.device).view(-1, B, Cc, 2)
    sc = _prof_scratch(stat)
    _call("miseg_instnorm_stats", L.InstnormStats(_ptr(x), ld, B, S, Cc, _dt(x), _ptr(stat)), prof=("instnorm", 0.0, _nb(x)),
          prof_params=L.InstnormStats(_ptr(x), ld, B, S, Cc, _dt(x), _ptr(sc)) if sc is not None else None)
    return stat


def _style_arrays(tensors, n):
    arr = (C.c_void_p * L.MAX_STYLES)()
    for i in range(L.MAX_STYLES):
        arr[i] = tensors[i].data_ptr() if (tensors is not None and i < n and tensors[i] is not None) else None
    return arr


def rank1_stats(x1, w):
    """instance-norm statistics of round(x1[row] * w[c]) without storing it (one sample): x1 rows view [.., 1], w [C] / [C, 1] in x1.dtype"""
    ld, n, one = rows(x1)
    assert one == 1 and w.dtype == x1.dtype and w.is_contiguous()
    Cc = w.numel()
    stat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(1, Cc) // 8, x1.device).view(-1, 1, Cc, 2)
    L.check(L.load().miseg_rank1_stats(_ptr(x1), ld, _ptr(w), 1, n, Cc, _dt(x1), _ptr(stat), _stream()), "rank1_stats")
    return stat


def instnorm_apply(x, B, S, stat, styles, gammas, betas, res=None, act=L.ACT_NONE, slope=0.01, eps=1e-5, out=None, res_stat=None,
                   res_gammas=None, res_betas=None, r1=None):
    """res_stat: `res` is the RAW input of a second instance norm (statistics res_stat, affine rows res_gammas / res_betas) applied on the fly.
    r1 = (x1, w): that raw input is round(x1[row] * w[c]) and is not stored (res=None; res_stat from rank1_stats)."""
    ld, n, Cc = rows(x)
    y = out if out is not None else torch.empty(x.shape, dtype=x.dtype, device=x.device)
    ldy, ny, Cy = rows(y)
    assert ny == n and Cy == Cc
    ldr = rows(res)[0] if res is not None else 0
    ns = len(gammas) if gammas is not None else 1
    p = L.InstnormApply(_ptr(x), ld, _ptr(res), ldr, _ptr(y), ldy, B, S, Cc, _dt(x), _ptr(stat), eps, _ptr(styles), ns,
                        _style_arrays(gammas, ns), _style_arrays(betas, ns), act, slope, _ptr(res_stat), _style_arrays(res_gammas, ns),
                        _style_arrays(res_betas, ns))
    if r1 is not None:
        assert res is None and rows(r1[0])[1] == n and r1[1].numel() == Cc and r1[1].dtype == x.dtype
        p.r1x, p.ldr1x, p.r1w = _ptr(r1[0]), rows(r1[0])[0], _ptr(r1[1])
    _call("miseg_instnorm_apply", p, prof=("instnorm", 0.0, _nb(x, res, y, r1[0] if r1 is not None else None)))
    return y


def instnorm_fwd(x, B, S, styles, gammas, betas, res=None, act=L.ACT_NONE, slope=0.01, eps=1e-5, out=None):
    """statistics + normalisation (one register-resident launch for tensors of <= 2048 rows per sample); returns (y, stat)."""
    ld, n, Cc = rows(x)
    assert n == B * S, (n, B, S)
    stat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(B, Cc) // 8, x.device).view(-1, B, Cc, 2)
    y = out if out is not None else torch.empty(x.shape, dtype=x.dtype, device=x.device)
    ldr = rows(res)[0] if res is not None else 0
    ns = len(gammas) if gammas is not None else 1
    p = L.InstnormApply(_ptr(x), ld, _ptr(res), ldr, _ptr(y), rows(y)[0], B, S, Cc, _dt(x), _ptr(stat), eps, _ptr(styles), ns,
                        _style_arrays(gammas, ns), _style_arrays(betas, ns), act, slope)
    _call("miseg_instnorm_fwd", p, prof=("instnorm", 0.0, _nb(x, res, y)))      # (statistics are stored, not accumulated: repeatable)
    return y, stat


def instnorm_fwd_slabs(x, pending, B, S, styles, gammas, betas, res=None, act=L.ACT_NONE, slope=0.01, eps=1e-5, out=None):
    """instnorm_fwd on the output `x` of a split convolution that is still `pending` (PendingSlabs): ONE launch sums the slabs, writes x,
    and normalises; returns (y, stat)."""
    ld, n, Cc = rows(x)
    assert n == B * S and ld == Cc, (n, B, S, ld, Cc)
    stat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(B, Cc) // 8, x.device).view(-1, B, Cc, 2)
    y = out if out is not None else torch.empty(x.shape, dtype=x.dtype, device=x.device)
    ldr = rows(res)[0] if res is not None else 0
    ns = len(gammas) if gammas is not None else 1
    p = L.InstnormApply(_ptr(x), ld, _ptr(res), ldr, _ptr(y), rows(y)[0], B, S, Cc, _dt(x), _ptr(stat), eps, _ptr(styles), ns,
                        _style_arrays(gammas, ns), _style_arrays(betas, ns), act, slope)
    _call("miseg_instnorm_fwd_slabs", p, prof=("instnorm", 0.0, _nb(x, res, y) + 4.0 * pending.n * pending.stride), extra=(_ptr(pending.ws), pending.n, pending.stride))
    return y, stat


def instnorm_bwd(dy, y, x, B, S, stat, styles, gammas, dgammas, dbetas, act=L.ACT_NONE, slope=0.01, eps=1e-5, want_dres=False, gadd=None, betas=None,
                 pending=None):
    """y=None with a LeakyReLU: valid when no residual entered the activation; the kernels recompute its sign from x (needs betas).
    pending (PendingSlabs): dy is the unwritten output of a split data-gradient convolution - the one-launch kernel sums its slabs itself."""
    ld, n, Cc = rows(x)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    dres = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_dres else None
    dstat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(B, Cc) // 8, x.device)
    ns = len(gammas) if gammas is not None else 1
    mk = lambda ds, dg, db: L.InstnormBwd(_ptr(dy), rows(dy)[0], _ptr(y), rows(y)[0] if y is not None else 0, _ptr(x), ld, _ptr(dx), rows(dx)[0],
                                          _ptr(dres), rows(dres)[0] if dres is not None else 0, B, S, Cc, _dt(x), _ptr(stat), eps, _ptr(ds), _ptr(styles), ns,
                                          _style_arrays(gammas, ns), _style_arrays(dg, ns), _style_arrays(db, ns), act, slope,
                                          _ptr(gadd), rows(gadd)[0] if gadd is not None else 0, _style_arrays(betas, ns))
    # (roofline leg: the repeated launches accumulate their reduction into scratch and leave the affine gradients alone)
    sc = _prof_scratch(dstat)
    if pending is not None:
        _call("miseg_instnorm_bwd_slabs", mk(dstat, dgammas, dbetas), prof=("instnorm", 0.0, _nb(y, x, dx, dres, gadd) + 4.0 * pending.n * pending.stride),
              prof_params=mk(sc, None, None) if sc is not None else None, extra=(_ptr(pending.ws), pending.n, pending.stride))
        return dx, dres
    _call("miseg_instnorm_bwd", mk(dstat, dgammas, dbetas), prof=("instnorm", 0.0, _nb(dy, y, x, dx, dres, gadd)),
          prof_params=mk(sc, None, None) if sc is not None else None)
    return dx, dres


def instnorm_bwd_apply(dy, x, B, S, stat, dstat, styles, gammas, dgammas, dbetas, eps=1e-5, gadd=None):
    """the apply half of instnorm_bwd with the backward sums already in `dstat` (the GEMM that produced dy added them in its epilogue)"""
    ld, n, Cc = rows(x)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    ns = len(gammas) if gammas is not None else 1
    p = L.InstnormBwd(_ptr(dy), rows(dy)[0], None, 0, _ptr(x), ld, _ptr(dx), rows(dx)[0], None, 0, B, S, Cc, _dt(x), _ptr(stat), eps, _ptr(dstat), _ptr(styles), ns,
                      _style_arrays(gammas, ns), _style_arrays(dgammas, ns), _style_arrays(dbetas, ns), L.ACT_NONE, 0.0,
                      _ptr(gadd), rows(gadd)[0] if gadd is not None else 0, _style_arrays(None, ns))
    # (roofline leg: a second pass would add the affine gradients twice - the profiled step's numbers are discarded anyway)
    _call("miseg_instnorm_bwd_apply", p, prof=("instnorm", 0.0, _nb(dy, x, dx, gadd)))
    return dx


def instnorm_bwd_reduce(dy, x, B, S, stat, eps=1e-5):
    """dstat [R, B, C, 2] fp64 whose replicas sum to (sum dy, sum dy * xhat) per (sample, channel); xhat from `stat` (group / batch norms)"""
    ld, n, Cc = rows(x)
    assert n == B * S and rows(dy)[1] == n
    dstat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(B, Cc) // 8, x.device).view(-1, B, Cc, 2)
    p = L.InstnormBwd(_ptr(dy), rows(dy)[0], None, 0, _ptr(x), ld, None, 0, None, 0, B, S, Cc, _dt(x), _ptr(stat), eps, _ptr(dstat), None, 1,
                      _style_arrays(None, 1), _style_arrays(None, 1), _style_arrays(None, 1), L.ACT_NONE, 0.0, None, 0, _style_arrays(None, 1))
    _call("miseg_instnorm_bwd_reduce", p)
    return dstat


def affine2(a, x, coef, B, S):
    """y[b, s, c] = coef[b, c, 0] * a + coef[b, c, 1] * x + coef[b, c, 2]; a, x rows views of [B * S, C], coef fp32 [B, C, 3]"""
    lda, n, Cc = rows(a)
    assert n == B * S and rows(x)[1] == n and coef.dtype == torch.float32 and coef.is_contiguous() and tuple(coef.shape) == (B, Cc, 3)
    y = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    _call("miseg_affine2", L.Affine2(C.sizeof(L.Affine2), _ptr(a), lda, _ptr(x), rows(x)[0], _ptr(y), rows(y)[0], _ptr(coef), B, S, Cc, _dt(a)))
    return y


def instnorm_pair_bwd(dy, y, xa, xb, B, S, stat_a, stat_b, styles, gammas_a, gammas_b, dgammas_a, dbetas_a, dgammas_b, dbetas_b, slope=0.01, eps=1e-5,
                      betas_a=None, betas_b=None, r1=None):
    """backward of LeakyReLU(norm_a(xa) + norm_b(xb)): (dxa, dxb) in one reduction + one apply launch.
    y=None: the kernels recompute the activation's sign from xa / xb (needs the betas of affine norms)."""
    ld, n, Cc = rows(xa)
    dxa = torch.empty(xa.shape, dtype=xa.dtype, device=xa.device)
    # r1 = (x1, w, dw): xb = round(x1[row] * w[c]) is not stored (xb = y = None), dw [C] fp32 receives (+=) the 1x1x1 weight gradient
    dxb = torch.empty(xb.shape, dtype=xb.dtype, device=xb.device) if r1 is None else None
    nb = L.load().miseg_instnorm_stat_bytes(B, Cc) // 8
    dsa, dsb = STAT_POOL.take(nb, xa.device), STAT_POOL.take(nb, xa.device)
    ns = len(gammas_a) if gammas_a is not None else 1
    if r1 is not None:
        assert xb is None and y is None and rows(r1[0])[1] == n and r1[1].numel() == Cc and r1[2].numel() == Cc and r1[2].dtype == torch.float32

    def mk(da, db_, ga, ba, gb, bb, dw):
        p = L.InstnormPairBwd(_ptr(dy), rows(dy)[0], _ptr(y), rows(y)[0] if y is not None else 0, _ptr(xa), ld, _ptr(xb), rows(xb)[0] if xb is not None else 0,
                              _ptr(dxa), rows(dxa)[0], _ptr(dxb), rows(dxb)[0] if dxb is not None else 0,
                              B, S, Cc, _dt(xa), _ptr(stat_a), _ptr(stat_b), eps, _ptr(da), _ptr(db_), _ptr(styles), ns,
                              _style_arrays(gammas_a, ns), _style_arrays(gammas_b, ns), _style_arrays(ga, ns), _style_arrays(ba, ns),
                              _style_arrays(gb, ns), _style_arrays(bb, ns), slope, _style_arrays(betas_a, ns), _style_arrays(betas_b, ns))
        if r1 is not None:
            p.r1x, p.ldr1x, p.r1w, p.r1dw = _ptr(r1[0]), rows(r1[0])[0], _ptr(r1[1]), _ptr(dw)
        return p
    sa, sb = _prof_scratch(dsa), _prof_scratch(dsb)
    _call("miseg_instnorm_pair_bwd", mk(dsa, dsb, dgammas_a, dbetas_a, dgammas_b, dbetas_b, r1[2] if r1 is not None else None),
          prof=("instnorm", 0.0, _nb(dy, y, xa, xb, dxa, dxb, r1[0] if r1 is not None else None)),
          prof_params=mk(sa, sb, None, None, None, None, _prof_scratch(r1[2]) if r1 is not None else None) if sa is not None else None)
    return dxa, dxb


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    ld, n, Cc = rows(x)
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    mean = torch.empty(n, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n, dtype=torch.float32, device=x.device)
    _call("miseg_layernorm_fwd", L.LayernormFwd(_ptr(x), ld, _ptr(y), rows(y)[0], n, Cc, _dt(x), eps, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd)))
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta):
    ld, n, Cc = rows(x)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _call("miseg_layernorm_bwd", L.LayernormBwd(_ptr(dy), rows(dy)[0], _ptr(x), ld, _ptr(dx), rows(dx)[0], n, Cc, _dt(x), _ptr(gamma), _ptr(mean),
                                               _ptr(rstd), _ptr(dgamma), _ptr(dbeta)))
    return dx


# ------------------------------------------------------------------------------------------ GEMM family
# Hand-offs between a producer op and the op right behind it are PER THREAD (the forward pass runs on the caller's thread, the backward pass on
# the autograd engine's device thread; two models stepped from two threads never see each other's): the statistics a GEMM left for the norm
# that consumes its output, and the slab sums a split data-gradient convolution left to the norm backward behind it
_TLS = threading.local()
_ALL_PENDING = []          # every thread's pending-slab table (check_no_pending looks at all of them from the stepping thread)
_ALL_PENDING_LOCK = threading.Lock()


def _set_gemm_stat(out, stat):
    """(key of the output, statistics tensor) of the gemm_nt / mlp_fwd call that just fused them, or None"""
    _TLS.gemm_stat = None if out is None else ((out.data_ptr(), out.numel(), out.dtype), stat)


def pop_gemm_stat(y):
    """the statistics the last gemm_nt / mlp_fwd of THIS thread produced for `y` in its epilogue (handed over once), or None"""
    st = getattr(_TLS, "gemm_stat", None)
    _TLS.gemm_stat = None
    return st[1] if st is not None and st[0] == (y.data_ptr(), y.numel(), y.dtype) else None


class NormRef:
    """a (conditional) instance norm whose apply pass a consumer kernel folds into its operand load (one sample): statistics of the norm's
    raw input + affine rows (miseg_norm_ref, include/miseg_hip.h)"""
    __slots__ = ("stat", "styles", "gammas", "betas", "eps")

    def __init__(self, stat, styles, gammas, betas, eps):
        self.stat, self.styles, self.gammas, self.betas, self.eps = stat, styles, gammas, betas, float(eps)

    def fill(self, ref):
        ns = len(self.gammas) if self.gammas is not None else 1
        ref.stat, ref.styles, ref.num_styles, ref.eps = _ptr(self.stat), _ptr(self.styles), ns, self.eps
        ref.gamma, ref.beta = _style_arrays(self.gammas, ns), _style_arrays(self.betas, ns)


FOLD_NORMS = not os.environ.get("MISEG_NO_NORM_FOLD")      # A/B switch of round 5: the Swin block's norm apply / norm-backward reduce as launches of their own
# the norm-backward sums in the epilogue of the SMALL-M data-gradient GEMM (<= 2048 rows: the deep Swin stages): the kernels have it (tested),
# the step does not use it - there the norm backward is ONE register-resident launch already, the fold would trade it for an apply launch
# (same launch count; measured 153.0 / 153.4 with it against 152.9 / 153.2 without: nothing); the forward folds stay on at every stage
SMALL_BSTAT = bool(os.environ.get("MISEG_SMALL_BSTAT"))


def gemm_nt_folds(a, w, anorm=None, bstat_x=None, act=L.ACT_NONE, res=None):
    """would gemm_nt(a, w, anorm=...) / gemm_nt(..., bstat=...) run fused for these operands?  (one sample is the caller's business)"""
    if not FOLD_NORMS or a.dtype != torch.bfloat16:
        return False
    lda, M, K = rows(a)
    N = w.shape[0]
    if bstat_x is not None and M <= 2048 and not SMALL_BSTAT:
        return False
    # (the output will be a fresh [M, N] tensor: A's pointer stands in for its alignment, and for the not-yet-allocated statistics)
    p = L.Gemm(_ptr(a), lda, _ptr(w), K, _ptr(a), N, M, N, K, 0, 0, _dt(a), _dt(a), None, act, 0, 1, None, _ptr(res), rows(res)[0] if res is not None else 0, None, 0, 0, 0, None)
    if anorm is not None:
        anorm.fill(p.an)
        return bool(L.load().miseg_gemm_fuses_anorm(C.byref(p)))
    p.stat, p.stat_mode, p.bs_x, p.ld_bs_x, p.bs_stat = _ptr(a), 2, _ptr(bstat_x), rows(bstat_x)[0], _ptr(a)
    return bool(L.load().miseg_gemm_fuses_bstat(C.byref(p)))


def gemm_nt(a, w, bias=None, act=L.ACT_NONE, out=None, out_dtype=None, split_k=1, res=None, preact_out=None, gelu_grad_of=None, want_stat=False, anorm=None,
            anorm_out=False, bstat=None):
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias) + res ; a rows view, w contiguous [N,K] in a.dtype.
    preact_out: tensor that receives the pre-activation z (for a later GELU backward); gelu_grad_of: pre-activation h of the
    layer in front, the result is multiplied by gelu'(h) (GELU backward folded into this data-gradient GEMM).
    anorm (NormRef; check gemm_nt_folds first): `a` is the RAW input of that instance norm, normalised as it is loaded; anorm_out: also return
    norm(a) - result (out, norm_a).  bstat = (x, stat, eps): `out` is the gradient with respect to the output of the instance norm whose raw
    input is x (forward statistics `stat`); its backward sums are left for pop_gemm_stat(out) (dstat layout of instnorm_bwd)."""
    lda, M, K = rows(a)
    N, Kw = w.shape
    if not (Kw == K and w.is_contiguous() and w.dtype == a.dtype):
        raise ValueError(f"gemm_nt: weight {tuple(w.shape)} {w.dtype} does not match activations K={K} {a.dtype}")
    odt = out_dtype or a.dtype
    if out is None:
        out = torch.empty(a.shape[:-1] + (N,), dtype=odt, device=a.device)
    ldc, Mo, No = rows(out)
    assert Mo == M and No == N
    assert preact_out is None or gelu_grad_of is None
    aux = preact_out if preact_out is not None else gelu_grad_of
    for t in (res, aux):
        if t is not None:
            _, Mt, Nt = rows(t)
            if Mt != M or Nt != N or t.dtype != out.dtype:
                raise ValueError("gemm_nt: res / aux must be [M, N] row views in the output dtype")
    p = L.Gemm(_ptr(a), lda, _ptr(w), K, _ptr(out), ldc, M, N, K, 0, 0, _dt(a), _dt(out), _ptr(_fp32(bias)), act, 0, split_k, None,
               _ptr(res), rows(res)[0] if res is not None else 0, _ptr(aux), rows(aux)[0] if aux is not None else 0,
               1 if preact_out is not None else 2 if gelu_grad_of is not None else 0, 0, None)
    _set_gemm_stat(None, None)
    xn = None
    if anorm is not None:
        assert not want_stat and bstat is None
        anorm.fill(p.an)
        if anorm_out:
            xn = torch.empty(a.shape, dtype=a.dtype, device=a.device)
            p.an_out, p.ld_an_out = xn.data_ptr(), rows(xn)[0]
    if bstat is not None:
        assert not want_stat
        bx, bst, beps = bstat
        dstat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(1, N) // 8, a.device)
        p.stat, p.stat_mode, p.bs_x, p.ld_bs_x, p.bs_stat, p.bs_eps = dstat.data_ptr(), 2, _ptr(bx), rows(bx)[0], _ptr(bst), float(beps)
        _set_gemm_stat(out, dstat)
    if want_stat and not os.environ.get("MISEG_NO_GEMM_STAT") and L.load().miseg_gemm_fuses_stat(C.byref(p)):
        # all M rows are one sample (the caller checked): the kernel leaves the norm statistics of the output in `stat`
        stat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(1, N) // 8, a.device).view(-1, 1, N, 2)
        p.stat = stat.data_ptr()
        _set_gemm_stat(out, stat)
    _call("miseg_gemm", p, prof=("gemm_nt", 2.0 * M * N * K, _nb(a, w, out, res, aux, xn, bstat[0] if bstat is not None else None)))      # (no accumulating output on the NT side: repeatable)
    return (out, xn) if anorm_out else out


def mlp_fused(x, hid):
    """the fused MLP kernels take this problem (bf16, 48 -> 192 -> 48 channels, >= 4096 tokens)"""
    if os.environ.get("MISEG_NO_FUSED_MLP"):
        return False
    _, M, Cc = rows(x)
    return x.dtype == torch.bfloat16 and bool(L.load().miseg_mlp_fused(M, Cc, hid, _dt(x)))


def mlp_fwd(x, w1, b1, w2, b2, res=None, want_stat=False, anorm=None, anorm_out=False):
    """y = w2 gelu(w1 x + b1) + b2 (+ res) in one launch, hidden activations never stored; w1 [HID, C], w2 [C, HID] contiguous in x.dtype.
    want_stat (all rows one sample): the instance-norm statistics of y are left for pop_gemm_stat(y).
    anorm (NormRef): x is the RAW input of that instance norm (one sample), normalised as it is loaded; anorm_out: also return norm(x)."""
    ldx, M, Cc = rows(x)
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    p = L.Mlp(C.sizeof(L.Mlp), M, Cc, w1.shape[0], _dt(x), _ptr(x), ldx, _ptr(w1), _ptr(_fp32(b1)), _ptr(w2), _ptr(_fp32(b2)),
              _ptr(res), rows(res)[0] if res is not None else 0, _ptr(y), rows(y)[0], None)
    xn = None
    if anorm is not None:
        anorm.fill(p.an)
        if anorm_out:
            xn = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            p.an_out, p.ld_an_out = xn.data_ptr(), rows(xn)[0]
    _set_gemm_stat(None, None)
    if want_stat:
        stat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(1, Cc) // 8, x.device).view(-1, 1, Cc, 2)
        p.stat = stat.data_ptr()
        _set_gemm_stat(y, stat)
    _call("miseg_mlp_fwd", p, prof=("gemm_nt", 4.0 * M * Cc * w1.shape[0], _nb(x, y, res, xn)))
    return (y, xn) if anorm_out else y


def mlp_bwd(x, dy, w1, b1, w2t, w1t, need_dx=True, bstat=None):
    """(dz, h, dx) of the fused MLP: dz = (dy w2) * gelu'(z), h = gelu(z) with z = w1 x + b1 recomputed, dx = dz w1.
    w2t [HID, C] = w2 transposed, w1t [C, HID] = w1 transposed (cast_matrix(..., transpose=True)).
    bstat = (x_raw, stat, eps): x is norm(x_raw); dx is the gradient with respect to that norm's output and its backward sums come back as a
    fourth result (dstat layout of instnorm_bwd)."""
    ldx, M, Cc = rows(x)
    hid = w1.shape[0]
    dz = torch.empty(x.shape[:-1] + (hid,), dtype=x.dtype, device=x.device)
    h = torch.empty_like(dz)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device) if need_dx else None
    p = L.Mlp(C.sizeof(L.Mlp), M, Cc, hid, _dt(x), _ptr(x), ldx, _ptr(w1), _ptr(_fp32(b1)), None, None, None, 0, None, 0, None,
              _ptr(dy), rows(dy)[0], _ptr(w2t), _ptr(w1t), _ptr(dz), rows(dz)[0], _ptr(h), rows(h)[0], _ptr(dx), rows(dx)[0] if dx is not None else 0)
    dstat = None
    if bstat is not None and dx is not None:
        bx, bst, beps = bstat
        dstat = STAT_POOL.take(L.load().miseg_instnorm_stat_bytes(1, Cc) // 8, x.device)
        p.bs_x, p.ld_bs_x, p.bs_stat, p.bs_eps, p.bs_dstat = _ptr(bx), rows(bx)[0], _ptr(bst), float(beps), dstat.data_ptr()
    _call("miseg_mlp_bwd", p, prof=("gemm_nt", 8.0 * M * Cc * hid, _nb(x, dy, dz, h, dx)))
    return (dz, h, dx, dstat) if bstat is not None else (dz, h, dx)


def gemm_nt_scatter(a, w, dst, grid):
    """The GEMM of a ConvTranspose3d(k2, s2) with its 2x2x2 scatter as the store: a = the voxels of `grid` = (B, d, h, w) as rows [.., Cin],
    w = [(j, co)][ci] (j = 4 jd + 2 jh + jw), dst = rows view [B, 2d, 2h, 2w, co] (e.g. the left half of a concat buffer).
    Returns False where the kernel path has no scattered store (caller: gemm_nt + channel_to_space)."""
    lda, M, K = rows(a)
    N, Kw = w.shape
    ldc, Md, Cout = rows(dst)
    B, d, h, wd = grid
    if not (Kw == K and w.is_contiguous() and w.dtype == a.dtype and dst.dtype == a.dtype and N == 8 * Cout and Md == 8 * M and M == B * d * h * wd):
        raise ValueError("gemm_nt_scatter: operand shapes")
    p = L.Gemm(_ptr(a), lda, _ptr(w), K, _ptr(dst), ldc, M, N, K, 0, 0, _dt(a), _dt(dst), None, L.ACT_NONE, 0, 1, None, None, 0, None, 0, 0, 0, None,
               d, h, wd, Cout)
    if not L.load().miseg_gemm_fuses_scatter(C.byref(p)):
        return False
    _call("miseg_gemm", p)
    return True


def gemm_tn_regroups(a, b, out):
    """can gemm_tn(a, b, out, accumulate=..., regroup=...) run for these operands?  (needs a step queue: the regrouped store lives in the
    grouped launch / the batched partial-tile sum at the end of the backward pass)"""
    if _queues(out) is None or os.environ.get("MISEG_NO_TN_REGROUP"):      # (the A/B switch of round 5: gemm_tn + permute3)
        return False
    lda, K, M = rows(a)
    N = rows(b)[2]
    if not (M % 48 == 0 and N % 48 == 0 and K >= 2048 and a.dtype == torch.bfloat16):
        return True                      # grouped launch
    p = L.Gemm(_ptr(a), lda, _ptr(b), rows(b)[0], _ptr(out), N, M, N, K, 1, 1, _dt(a), L.F32, None, L.ACT_NONE, 1, 0, None, None, 0, None, 0, 0, 0)
    return L.load().miseg_gemm_tn_splits(C.byref(p)) > 1      # streaming kernel with partial tiles: their deferred sum regroups


FOLD_COLSUM = os.environ.get("MISEG_NO_COLSUM_FOLD") is None      # A/B switch of round 5 (gemm_tn(colsum_out=))


def gemm_tn(a, b, out=None, accumulate=False, split_k=0, regroup=0, colsum_out=None):
    """out[M,N] (fp32) (+)= a[K,M]^T @ b[K,N]; a, b row views sharing the row count K (weight gradients).
    regroup = c > 0 (accumulate mode, after gemm_tn_regroups said yes): column j * c + i of the product is stored at column i * (N / c) + j.
    colsum_out (fp32 [M], accumulate mode): += the column sums of a - with a = dy the bias gradient of the linear layer whose weight gradient
    this is.  On the streaming path they ride in the product's launch (miseg_gemm_params.tn_colsum); elsewhere this is colsum(a, colsum_out,
    accumulate=True)."""
    lda, K, M = rows(a)
    ldb, Kb, N = rows(b)
    assert K == Kb and a.dtype == b.dtype
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
        accumulate = False
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == M * N
    q = _queues(out) if accumulate else None
    assert not regroup or (q is not None and split_k <= 0 and N % regroup == 0)
    if q is not None:
        q.writes[out.data_ptr()] = q.writes.get(out.data_ptr(), 0) + 1      # step-wide writer count of the slot (direct and queued, main and side)
    if q is not None and split_k <= 0 and not (M % 48 == 0 and N % 48 == 0 and K >= 2048 and a.dtype == torch.bfloat16):
        q.lists().gemm_tn.append((a, b, out, int(accumulate) == 2, int(regroup)))      # small problem: grouped launch at the end of the backward pass
        if colsum_out is not None:
            colsum(a, colsum_out, accumulate=True)
        return out
    accumulate = bool(accumulate)      # (the direct kernels always add: "known zero" only saves the grouped launch its read of the slot)
    split_k = max(0, split_k)          # 0: the library picks the kernel and the split over the reduction rows
    p = L.Gemm(_ptr(a), lda, _ptr(b), ldb, _ptr(out), N, M, N, K, 1, 1, _dt(a), L.F32, None, L.ACT_NONE, int(accumulate), split_k, None, None, 0, None, 0, 0, 0)
    lib = L.load()
    if colsum_out is not None:
        if FOLD_COLSUM and split_k == 0 and colsum_out.dtype == torch.float32 and lib.miseg_gemm_tn_fuses_colsum(C.byref(p)):
            p.tn_colsum = colsum_out.data_ptr()
        else:
            colsum(a, colsum_out, accumulate=True)
    wsb = lib.miseg_gemm_workspace_bytes(C.byref(p))
    if wsb:
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=a.device)
        p.workspace = ws.data_ptr()
        if q is not None:      # the per-split partial tiles are summed by one batched launch later
            p.defer_reduce = 1
            q.lists().tn_reduce.append((ws, out, N, M, N, lib.miseg_gemm_tn_splits(C.byref(p)), int(regroup)))
    assert not regroup or p.defer_reduce, "gemm_tn: regroup on a path without a deferred sum (ask gemm_tn_regroups first)"
    _call("miseg_gemm", p)
    return out


def permute3(src, dst, n, strides, accumulate=False):
    lib = L.load()
    L.check(lib.miseg_permute3(_ptr(src), _ptr(dst), n[0], n[1], n[2], strides[0], strides[1], strides[2], int(accumulate), _stream()), "permute3")
    return dst


WGRAD_STREAM = None   # side HIP stream for the weight-gradient kernels (set by runtime/arena.py); None: launch in line
WGRAD_KINDS = ("gemm", "conv")
_WGRAD_KEEP = []      # operands of side-stream launches, kept alive until the join


class _Side:
    """`with ops.wgrad_side(x, dy): <launch a weight-gradient kernel>`: the launch goes to the side stream behind everything
    already queued on the current one.  Weight gradients only feed the gradient arena, so they are off the critical dX chain
    of the backward pass and fill the CUs its many small-grid kernels leave idle."""

    def __init__(self, keep, kind):
        self.side = WGRAD_STREAM if kind in WGRAD_KINDS else None
        if self.side is not None:
            _WGRAD_KEEP.extend(keep)

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            self.ctx.__exit__(*exc)
        return False


def wgrad_side(*keep, kind="gemm"):
    return _Side(keep, kind)


_BRANCH_STREAM = None
_MAIN_STREAM = None            # the stream a model's forward forked its side branch from (swin_unetr.py sets it)
BACKGROUND_LAUNCHES = 0        # running count of launches issued in background form (bench.py reports the per-step figure)
BACKGROUND_WORKGROUPS = 32     # cap of the branch's own background weight-gradient launch (each workgroup owns a CU).  Step against no branch,
                               # two boxes, before the deferral below: 256 (no cap) -0.4 %, 128 +1.0, 64 +1.4, 32 +2.0 / +2.4, 16 +2.0, 8 -3.8
DEFERRED_WORKGROUPS = 64       # cap of the main stream's deferred weight gradients (defer_to_branch).  (own cap, this cap) on one box, no branch =
                               # 132.6, branch alone 134.5: (24, 48) 136.9, (24, 64) 140.8, (24, 96) 139.8, (48, 64) 140.6, (64, 64) 140.7, (32, 32) 128.9
DEFERRED_WORKGROUPS_SPLIT = 96      # the same launches inside the first half of a split step (flush_deferred_on_branch)
GROUP_EARLY_WORKGROUPS = 0      # cap of the early (branch-stream) grouped weight-gradient launch; 0 = off
FLUSH_SMALL_ON_BRANCH = True      # (rounds 3 - 4 measured the alternatives; re-swept in round 5 after the grouped conv launch halved: 160.7 against 159.7 / 158.7 / 158.0
FLUSH_SIDE_ON_BRANCH = True       #  with this one / the next / the third off - module constants again)
FLUSH_MAIN_BEFORE_JOIN = True   # arena.end_backward: the main stream's grouped launches do not wait for the branch
DEFER_MIN_ROWS = 400000        # 96^3 layers only (the smaller ones are grouped into one launch at the end of the backward pass)
# (round 3: the queue of deferred launches itself lives on the training arena's StepQueues - `branch_deferred` - and is found through the
# gradient slot a launch accumulates into, like the other per-step queues: two models in one process no longer share it)


def branch_stream(device):
    """the stream of a model's side branch (SwinUNETR: the two image-resolution encoder blocks).  Autograd replays the branch's backward
    pass on it - beside the latency-bound launches of the deep stages - and `join_branch` joins it."""
    global _BRANCH_STREAM
    if _BRANCH_STREAM is None or _BRANCH_STREAM.device != device:
        _BRANCH_STREAM = torch.cuda.Stream(device=device)
    return _BRANCH_STREAM


def open_branch_deferral(params):
    """a model's forward opens the deferral for this step: full-size conv weight gradients of the MAIN stream (SwinUNETR: decoder1's two 96^3
    layers) wait for the side branch's backward pass and run at its head, in background form.  Needs a training arena (the queue lives on
    its StepQueues); returns False without one."""
    q = _arena_queues(params)
    if q is None:
        return False
    q.branch_deferred = []
    return True


def close_branch_deferral(params):
    q = _arena_queues(params)
    if q is not None:
        q.branch_deferred = None


def _arena_queues(params):
    for p in params:
        arena = getattr(p, "_miseg_arena", None)
        return QUEUES.get(arena._qkey) if arena is not None else None
    return None


def defer_to_branch(x, dy, slot, mode):
    """True when the conv weight gradient (x, dy) -> slot was queued for the head of the side branch's backward pass: leaf work of the main
    stream's full-size kernels (0.3 ms per step on the critical path) that then runs beside the small-grid launches instead"""
    q = _queues(slot)
    if q is None or q.branch_deferred is None or in_branch_backward() or rows(x)[1] < DEFER_MIN_ROWS:
        return False
    q.branch_deferred.append((x, dy, slot, mode))
    return True


def flush_branch_deferred(q, cap=None):
    """launch what `defer_to_branch` queued on the current stream (the branch's, in background form; join_branch's as a fallback)"""
    global BACKGROUND_WORKGROUPS
    if q is None or not q.branch_deferred:
        if q is not None:
            q.branch_deferred = None
        return
    items, q.branch_deferred = q.branch_deferred, None
    keep, BACKGROUND_WORKGROUPS = BACKGROUND_WORKGROUPS, (cap or DEFERRED_WORKGROUPS)
    try:
        for x, dy, slot, mode in items:
            conv3_wgrad(x, dy, dw=slot, accumulate=mode)
    finally:
        BACKGROUND_WORKGROUPS = keep
    _WGRAD_KEEP.extend(t for it in items for t in it[:2])      # alive until join_wgrad



def flush_deferred_on_branch(params):
    """(gradient hook) issue what defer_to_branch queued so far on the BRANCH stream, behind everything the main stream has issued (see
    SwinUNETR.forward: the first half of a split step)"""
    q = _arena_queues(params)
    if q is None or not q.branch_deferred or _BRANCH_STREAM is None:
        return
    cur = torch.cuda.current_stream()
    if cur == _BRANCH_STREAM:
        return
    _BRANCH_STREAM.wait_stream(cur)
    with torch.cuda.stream(_BRANCH_STREAM):
        # (alone on the branch stream, with ~0.8 ms of the main stream's decoder chain to hide behind: 96 workgroups finish in time where
        # the one-graph step's 64 left the first half waiting - single-rank RCCL step 135.4 / 138.5 / 138.1 / 137.1 patches/s at 64 / 96 / 128 / 192)
        flush_branch_deferred(q, cap=DEFERRED_WORKGROUPS_SPLIT)


def early_group_flush(q):
    """at the TAIL of the side branch's backward pass (its last node calls this): the small layers' weight gradients queued so far go out on
    the branch stream as one grouped launch in background form, beside what the main stream has left of its small-grid chain - instead of
    after it, at the very end of the step.  (At the HEAD of the branch the same launch made the branch the critical path: 145.4 -> 134.7
    patches/s with 64 workgroups, 141.1 with 128.)  GROUP_EARLY_WORKGROUPS = 0 switches it off.
    Round 3, with the wait on the main stream this launch needs for its operands: 146.5 -> 114.0 patches/s uncapped, 111.1 with 128 - the
    device-clock stamps show the WHOLE branch starting 2.3 ms later (its head at 5.9 ms instead of 3.6): a second edge main -> branch inside
    the captured step makes the hipGraph executor run the branch's segment behind the main stream's.  The same happened to a third stream
    forked for this launch at the point where the main stream enters the Swin stages' backward pass (147 -> 118: branch and third stream
    shared one queue, GPU_MAX_HW_QUEUES=8 changed nothing).  One fork and one join per side stream is what replays concurrently."""
    if q is None or not q.conv_wgrad or not GROUP_EARLY_WORKGROUPS or not in_branch_backward() or _MAIN_STREAM is None:
        return
    # the queued layers' operands were produced on the MAIN stream: everything the host has issued there so far (autograd issues the nodes of
    # both streams in one order) is what the launch needs.  -1: no cap (the branch's chain is over, the launch is alone on its stream)
    torch.cuda.current_stream().wait_stream(_MAIN_STREAM)
    stamp("early_group_begin")
    _flush_conv_wgrads(q.conv_wgrad, background=max(GROUP_EARLY_WORKGROUPS, 0), keep=True)
    stamp("early_group_end")


def join_branch(flush_deferred=True, queues=None, flush_main=False):
    """the current stream waits for the branch stream: before anything that consumes what the branch's backward produced (the side queue's
    launches of arena.end_backward, the optimiser, the end of a hipGraph capture).  queues: the arena's StepQueues whose
    deferred launches are issued here when the branch's backward never ran (nothing in it needed a gradient).
    flush_main: the main stream's own queued launches go out BEFORE the wait (they read nothing of the branch's)."""
    global _FLUSHING
    if flush_deferred:
        for q in ([queues] if queues is not None else list(QUEUES.values())):
            if q.branch_deferred:
                flush_branch_deferred(q)
    stamp("main_chain_end")
    _FLUSHING = True
    if flush_main and queues is not None:
        cur = torch.cuda.current_stream() if _BRANCH_STREAM is not None else None      # (no side stream: the CPU ranks of the gloo tests)
        if queues.side is not None and _BRANCH_STREAM is not None and FLUSH_SIDE_ON_BRANCH and cur != _BRANCH_STREAM:
            # what the branch's backward queued goes out on the BRANCH stream, behind its last kernel: the branch ends ~0.3 ms before the main
            # chain (step stamps), so these launches run beside the main stream's last kernels instead of behind the join.  The main stream's
            # grouped GEMM weight gradients, partial-tile sums and column sums follow them there (after a wait for the main chain), beside
            # the main stream's grouped conv weight gradients: that launch holds one 96 KB workgroup per CU and leaves the rest of the CU idle
            with torch.cuda.stream(_BRANCH_STREAM):
                queues.side.flush()
            if FLUSH_SMALL_ON_BRANCH and (queues.gemm_tn or queues.tn_reduce or queues.colsum or queues.tiny_wgrad):
                for lst in (queues.gemm_tn, queues.tn_reduce, queues.colsum, queues.tiny_wgrad):      # operands of the main stream, read on the branch: alive until join_wgrad
                    _WGRAD_KEEP.extend(t for it in lst for t in it if isinstance(t, torch.Tensor))
                _BRANCH_STREAM.wait_stream(cur)
                with torch.cuda.stream(_BRANCH_STREAM):
                    # (MISEG_TINY_WGRAD_AT=flush: the tiny-volume conv weight gradients of encoder10 / decoder5 - write-bound, 176 MB in four ~20 us
                    # launches - go out here, beside the main stream's grouped conv weight gradients; measured slower than inline, see TINY_WGRAD_AT)
                    for x_, dy_, dw_, acc_ in queues.tiny_wgrad:
                        _conv3_wgrad_now(x_, dy_, dw_, acc_)
                    queues.tiny_wgrad.clear()
                    _flush_gemm_tn(queues.gemm_tn, queues.writes)
                    _flush_tn_reduces(queues.tn_reduce)
                    _flush_colsums(queues.colsum)
            if queues.on_branch_end is not None and cur != _BRANCH_STREAM:
                # round 5 (runtime/graph.py::GraphedTrainStep): work that only needs what the main CHAIN has produced - the optimiser update of the
                # parameters whose gradients were written inline (StepQueues.inline_final) - goes out on the branch stream here, beside the
                # main stream's grouped launches below, instead of behind the whole pass
                _BRANCH_STREAM.wait_stream(cur)
                with torch.cuda.stream(_BRANCH_STREAM):
                    queues.on_branch_end()
        queues.flush(side=False)
        stamp("main_flushed")
    if _BRANCH_STREAM is not None:
        if STAMPS is not None and not torch.cuda.current_stream() == _BRANCH_STREAM:
            stamp("branch_end", _BRANCH_STREAM)
        torch.cuda.current_stream().wait_stream(_BRANCH_STREAM)
        stamp("joined")
    _FLUSHING = False


def in_branch_backward():
    return _BRANCH_STREAM is not None and torch._C._current_graph_task_id() >= 0 and torch.cuda.current_stream() == _BRANCH_STREAM


def _background():
    """> 0 (the workgroup cap) when this launch belongs to the branch's BACKWARD pass: the 3x3x3 convolution kernels then run in their
    background form (miseg_conv3_params.background, miseg_conv3_wgrad_params.max_workgroups) so that the main stream's small-grid
    launches keep finding free CUs.  The branch's forward runs right in front of its join, beside nothing: normal form."""
    if not in_branch_backward():
        return 0
    global BACKGROUND_LAUNCHES
    BACKGROUND_LAUNCHES += 1
    return BACKGROUND_WORKGROUPS


def join_wgrad():
    """the current stream waits for the side streams; call once after the backward pass (arena.end_backward)."""
    join_branch()
    if WGRAD_STREAM is not None:
        torch.cuda.current_stream().wait_stream(WGRAD_STREAM)
    _WGRAD_KEEP.clear()


class StepQueues:
    """launch-bound tails of ONE model's backward pass, queued by the host and issued as grouped launches at the end of it
    (runtime/arena.py::end_backward): bias-gradient column sums, the small weight-gradient GEMMs, the partial-tile sums of the
    streaming ones, the conv weight gradients of the 48^3-and-smaller layers.  One instance per training arena, found through the
    storage of the gradient slot a kernel accumulates into - two models (two arenas) in one process do not share anything."""

    def __init__(self, side=True, writes=None):
        self.colsum, self.gemm_tn, self.tn_reduce, self.conv_wgrad, self.tiny_wgrad = [], [], [], [], []
        self.unzeroed = set()        # data_ptr of arena slots the step's fill left out (their weight-gradient kernel overwrites them whole: arena.begin_step);
                                     # conv3_wgrad zero-fills one first if the launch it is about to issue would read or only add to it
        self.inline_final = []       # arena slots whose ONLY write of the step happened inline, as a plain store (the tiny-volume conv weight gradients):
                                     # final as soon as the main chain of the backward pass is through
        self.on_branch_end = None    # callable issued on the branch stream where join_branch sends the end-of-pass small launches (see there)
        # gradient slot (data_ptr) -> how many weight-gradient GEMMs of this step write it, direct launches and both queues together: a
        # grouped launch may STORE into a slot ("known zero", miseg_gemm_tn_desc.zeroed) only when it is the slot's one writer of the step -
        # a tied weight's second use, wherever it was issued, would otherwise be overwritten or overwrite
        self.writes = {} if writes is None else writes
        self.branch_deferred = None      # while the model's side branch is open: [(x, dy, slot, mode)], see defer_to_branch
        # what the side branch's backward pass queues (it reads tensors the BRANCH stream produced) is kept apart: the main stream issues
        # its own grouped launches as soon as its chain ends - beside the branch's last full-size kernels, which run in background form on
        # a fraction of the CUs - and only the branch's few wait for the join (arena.end_backward)
        self.side = StepQueues(side=False, writes=self.writes) if side else None

    def lists(self):
        """the queue a launch issued NOW belongs to"""
        return self.side if (self.side is not None and in_branch_backward()) else self

    def flush_small(self):
        """the queued GEMM weight gradients, partial-tile sums and column sums only (the grouped conv weight gradients keep waiting): what
        completes the small parameters of a range whose conv weights were written inline (tiny-volume layers), cheaply, mid-chain.  With
        MISEG_TINY_WGRAD_AT=flush those conv weights are queued too: they go out here, or the range would be reduced without them"""
        for x, dy, dw, acc in self.tiny_wgrad:
            _conv3_wgrad_now(x, dy, dw, acc)
        self.tiny_wgrad.clear()
        _flush_gemm_tn(self.gemm_tn, self.writes)
        _flush_tn_reduces(self.tn_reduce)
        _flush_colsums(self.colsum)

    def flush(self, side=True):
        for x, dy, dw, acc in self.tiny_wgrad:
            _conv3_wgrad_now(x, dy, dw, acc)
        self.tiny_wgrad.clear()
        _flush_conv_wgrads(self.conv_wgrad)
        _flush_gemm_tn(self.gemm_tn, self.writes)
        _flush_tn_reduces(self.tn_reduce)
        _flush_colsums(self.colsum)
        if side and self.side is not None:
            self.side.flush()


QUEUES = {}             # data_ptr of an arena's flat gradient storage -> its StepQueues while a step of that arena is open
DEFAULT_QUEUES = None   # tests / micro-benchmarks: queue accumulate-mode launches whose destination belongs to no arena


def _queues(out):
    if out is None:
        return None
    if QUEUES:
        q = QUEUES.get(out.untyped_storage().data_ptr())
        if q is not None:
            return q
    return DEFAULT_QUEUES


def _flush_tn_reduces(q):
    if not q:
        return
    lib = L.load()
    if os.environ.get("MISEG_DEBUG_QUEUES"):
        import sys
        print("tn_reduce queue:", [(it[3], it[4], it[5]) for it in q], file=sys.stderr)
    for i in range(0, len(q), 32):
        chunk = q[i:i + 32]
        descs = (L.TnReduceDesc * len(chunk))()
        for j, (ws, out, ldc, M, N, splits, regroup) in enumerate(chunk):
            descs[j] = L.TnReduceDesc(_ptr(ws), _ptr(out), ldc, M, N, splits, 0, regroup, 0)
        L.check(lib.miseg_gemm_tn_reduce_batch(descs, len(chunk), _stream()), "gemm_tn_reduce_batch")
        _stamp_launch("gemm_tn_reduce_batch")
    q.clear()


def _flush_gemm_tn(q, writes=None):
    if not q:
        return
    lib = L.load()
    if os.environ.get("MISEG_DEBUG_QUEUES"):      # measurement aid: what the grouped launch holds (M, N, K, zeroed)
        import sys
        print("gemm_tn queue:", [(rows(it[0])[2], rows(it[1])[2], rows(it[0])[1], int(it[3]), it[4]) for it in q], file=sys.stderr)
    once = writes
    if once is None:      # (no step-wide count: at least the problems of this launch)
        once = {}
        for it in q:      # a slot written by two problems of the same launch (a shared weight) is never "known zero" for either
            once[it[2].data_ptr()] = once.get(it[2].data_ptr(), 0) + 1
    for dt in {it[0].dtype for it in q}:
        items = [it for it in q if it[0].dtype == dt]
        for i in range(0, len(items), 24):
            chunk = items[i:i + 24]
            descs = (L.GemmTnDesc * len(chunk))()
            for j, (a, b, out, zeroed, regroup) in enumerate(chunk):
                lda, K, M = rows(a)
                ldb, _, N = rows(b)
                descs[j] = L.GemmTnDesc(_ptr(a), lda, _ptr(b), ldb, _ptr(out), N, M, N, K, 1 if (zeroed and once.get(out.data_ptr(), 2) == 1) else 0, regroup, 0)
            L.check(lib.miseg_gemm_tn_group(descs, len(chunk), _dt(chunk[0][0]), _stream()), "gemm_tn_group")
            _stamp_launch("gemm_tn_group")
    q.clear()


def _flush_colsums(q):
    """issue the queued accumulate-mode column sums in batches of lib.MISEG_COLSUM_BATCH (one launch each)."""
    if not q:
        return
    lib = L.load()
    if os.environ.get("MISEG_DEBUG_QUEUES"):
        import sys
        print("colsum queue:", [(rows(t)[1], rows(t)[2]) for t, _ in q], file=sys.stderr)
    for dt in {t.dtype for t, _ in q}:
        items = [(t, o) for t, o in q if t.dtype == dt]
        for i in range(0, len(items), 32):
            chunk = items[i:i + 32]
            descs = (L.ColsumDesc * len(chunk))()
            for j, (t, o) in enumerate(chunk):
                ld, n, Cc = rows(t)
                descs[j] = L.ColsumDesc(_ptr(t), ld, n, _ptr(o), Cc, 0)
            L.check(lib.miseg_colsum_batch(descs, len(chunk), _dt(chunk[0][0]), _stream()), "colsum_batch")
            _stamp_launch("colsum_batch")
    q.clear()


def colsum(x, out=None, accumulate=False):
    ld, n, Cc = rows(x)
    q = _queues(out) if (out is not None and accumulate) else None
    if q is not None:
        q.lists().colsum.append((x, out))          # keeps x alive until the flush
        return out
    if out is None:
        out = torch.empty(Cc, dtype=torch.float32, device=x.device)
        accumulate = False
    _call("miseg_colsum", L.Colsum(_ptr(x), ld, n, Cc, _dt(x), _ptr(out), int(accumulate)))
    return out


def cast_matrix(w, dtype, transpose=False, regroup=None):
    """fp32 [R, C] parameter -> compute dtype, optionally transposed to [C, R].  regroup=(inner, outer) additionally
    renumbers the C index c -> (c % inner) * outer + c // inner (ConvTranspose3d k2s2 weights: (co, tap) -> (tap, co)).
    With a training arena (runtime/arena.py) the result is the copy refreshed by the step's batched kernel."""
    inner, outer = regroup if regroup is not None else (1, 1)
    if dtype == torch.float32 and not transpose and inner == 1:
        return _fp32(w).reshape(w.shape[0], -1)
    arena = getattr(w, "_miseg_arena", None)
    if arena is not None and arena.dtype == dtype:
        sh = arena.shadow(w, transpose, inner, outer)
        if sh is not None:
            return sh
    w2 = _fp32(w).reshape(w.shape[0], -1)
    R, Cc = w2.shape
    if inner != 1:
        # one-off path (first step / no arena): regroup in fp32 with the permute kernel, then cast
        tmp = torch.empty((Cc, R) if transpose else (R, Cc), dtype=torch.float32, device=w.device)
        if transpose:   # tmp[(j, o)][r] = w2[r][o * inner + j]
            permute3(w2, tmp, (inner, outer, R), (1, inner, Cc))
        else:           # tmp[r][(j, o)] = w2[r][o * inner + j]
            permute3(w2, tmp, (R, inner, outer), (Cc, 1, inner))
        if dtype == torch.float32:
            return tmp
        w2, transpose = tmp, False
        R, Cc = w2.shape
    out = torch.empty((Cc, R) if transpose else (R, Cc), dtype=dtype, device=w.device)
    _call("miseg_cast_matrix", L.Cast(_ptr(w2), _ptr(out), R, Cc, L.F32 if dtype == torch.float32 else L.BF16, int(transpose)))
    return out


# ------------------------------------------------------------------------------------------ conv 3x3x3
def _round_up(a, b):
    return (a + b - 1) // b * b


def pack_conv3(w, dtype, want_fwd=True, want_bwd=True):
    """packs of a 3x3x3 weight for the forward (x side) and data-gradient (dy side) kernels; with a training arena the
    packs refreshed by the step's batched kernel (runtime/arena.py)."""
    arena = getattr(w, "_miseg_arena", None)
    if arena is not None and arena.dtype == dtype:
        pk = arena.conv_packs(w)
        if pk is not None:
            return pk
    Cout, Cin = w.shape[0], w.shape[1]
    lib = L.load()
    dt = L.F32 if dtype == torch.float32 else L.BF16
    fwd = torch.empty(lib.miseg_pack_conv3_elems(Cin, Cout, dt, 0), dtype=dtype, device=w.device) if want_fwd else None
    bwd = torch.empty(lib.miseg_pack_conv3_elems(Cin, Cout, dt, 1), dtype=dtype, device=w.device) if want_bwd else None
    _call("miseg_pack_conv3_weight", L.PackConv3(_ptr(_fp32(w)), _ptr(fwd), _ptr(bwd), Cin, Cout, dt))
    return fwd, bwd


def _vol(x):
    assert x.dim() == 5, "expected [B, D, H, W, C]"
    return x.shape[0], x.shape[1], x.shape[2], x.shape[3]


def _pending():
    """this thread's table: key of a data gradient whose split convolution left its slabs to the norm backward that consumes it -> PendingSlabs"""
    d = getattr(_TLS, "pending", None)
    if d is None:
        d = _TLS.pending = {}
        with _ALL_PENDING_LOCK:
            _ALL_PENDING.append(d)
    return d


def pending_dx_put(dx, pend):
    _pending()[(dx.data_ptr(), dx.numel(), dx.dtype)] = pend


def pending_dx_take(dy):
    d = getattr(_TLS, "pending", None)
    return d.pop((dy.data_ptr(), dy.numel(), dy.dtype), None) if d else None


def check_no_pending():
    """every deferred slab sum must have been taken by its consumer (the instance-norm backward right behind the data-gradient convolution):
    a gradient tensor that reached anything else would have been read unwritten"""
    with _ALL_PENDING_LOCK:
        n = sum(len(d) for d in _ALL_PENDING)
        for d in _ALL_PENDING:
            d.clear()
    if n:
        raise RuntimeError(f"{n} deferred data-gradient slab sum(s) were never consumed (conv3(..., dx_to_norm=True) in front of something that is no instance norm)")


FOLD_SHORTCUT = os.environ.get("MISEG_NO_SC_FOLD") is None      # A/B switch of round 5 (conv3_fwd(sc=))


def conv3_fuses_shortcut(x, Cout, Csc):
    """can conv3_fwd(x, ..., Cout, sc=(g [.., Csc], w [Cout, Csc])) take the 1x1x1 term along (miseg_conv3_params.sc_x)?"""
    if not FOLD_SHORTCUT or x.dtype != torch.bfloat16:
        return False
    B, D, H, W = _vol(x)
    return bool(L.load().miseg_conv3_fuses_shortcut(B, D, H, W, rows(x)[2], Cout, Csc, _dt(x)))


FOLD_S2C = os.environ.get("MISEG_NO_S2C_FOLD") is None      # A/B switch of round 5 (conv3_fwd(s2c=))


def conv3_fuses_s2c(x, Cout, C_left):
    """can conv3_fwd(x, ..., Cout, s2c=...) store its first C_left output channels in space-to-channel order (miseg_conv3_params.s2c_out)?"""
    if not FOLD_S2C:
        return False
    B, D, H, W = _vol(x)
    return bool(L.load().miseg_conv3_fuses_s2c(B, D, H, W, rows(x)[2], Cout, C_left, _dt(x)))


FOLD_FWD_SHORTCUT = os.environ.get("MISEG_NO_FS_FOLD") is None      # A/B switch of round 5 (conv3_fwd(fs=))


def conv3_fuses_fwd_shortcut(x, Cout):
    """can conv3_fwd(x, ..., Cout, fs=...) produce the 1x1x1 convolution of x as a second output (miseg_conv3_params.fs_w)?"""
    if not FOLD_FWD_SHORTCUT or x.dtype != torch.bfloat16:
        return False
    B, D, H, W = _vol(x)
    ld, _, Cin = rows(x)
    return bool(L.load().miseg_conv3_fuses_fwd_shortcut(B, D, H, W, Cin, Cout, _dt(x))) and x.data_ptr() % 16 == 0 and ld % 8 == 0


def conv3_fwd(x, wpk, Cout, out=None, res=None, want_stat=False, defer=False, sc=None, s2c=None, fs=None):
    """x [B,D,H,W,Cin] rows view; wpk [Cout][27][CinP].  res: rows view added to the result in the epilogue (falls back to a separate
    add where the kernel path cannot fuse it).  want_stat: returns (out, stat) with stat the instance-norm statistics of `out`
    ([16, B, Cout, 2] fp64, from the kernel's epilogue) or None where that is not available (the caller's norm then computes them).
    sc = (g, w) (after conv3_fuses_shortcut said yes): out += g @ w^T, g a rows view over the same voxels, w [Cout, Csc] contiguous in x's
    dtype - the 1x1x1 shortcut term of a residual block's data gradient.
    s2c = tensor [B, D/2, H/2, W/2, 8 * C_left] (after conv3_fuses_s2c said yes): the first C_left output channels are stored THERE, in
    space-to-channel order (block j = 4 (d&1) + 2 (h&1) + (w&1)), and not in `out`.
    fs = (w [Cout, Cin] contiguous in x's dtype, want_stat2) (after conv3_fuses_fwd_shortcut said yes; not with defer): the launch also produces
    y2 = x @ w^T - a residual block's 1x1x1 shortcut convolution - and the return value is (out, stat | None, y2, stat2 | None)."""
    B, D, H, W = _vol(x)
    ld, n, Cin = rows(x)
    if out is None:
        out = torch.empty(B, D, H, W, Cout, dtype=x.dtype, device=x.device)
    lib = L.load()
    wsb = lib.miseg_conv3_fwd_workspace_bytes(B, D, H, W, Cin, Cout, _dt(x))
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device) if wsb else None
    flops = 2.0 * B * D * H * W * 27 * Cin * Cout
    fast = lib.miseg_conv3_k96(Cin, _dt(x)) != 0      # 96-byte chunks (padded where the rows are wide enough to pay for it)
    bg = 1 if (fast and _background()) else 0
    tiny_k = fast and not bg and x.data_ptr() % 16 == 0 and ld % 8 == 0 and bool(lib.miseg_conv3_fwd_tiny(B, D, H, W, Cin, Cout, _dt(x)))
    name = (f"conv3_fwd{'_tiny' if tiny_k else '96' if fast else ''}_kernel<{'bf16' if x.dtype == torch.bfloat16 else 'f32'}>"
            + (" (background)" if bg else ""))
    fuse_res = res is not None and fast and res.dtype == x.dtype       # (the roofline leg times the launches exactly as the step issues them)
    # algorithmic bytes: x read once, y written once, the weight pack, the fused residual read once
    nbytes = float(x.element_size()) * (B * D * H * W * (Cin + Cout + (Cout if fuse_res else 0)) + wpk.numel())
    stat = None
    # want_stat == "defer" (the caller's next op is an instance norm that can take the partial slabs: instnorm_fwd_slabs): a split launch
    # over <= 2048 rows per sample stops after its slabs and the norm's ONE launch sums them, writes `out`, and normalises
    defer_req = bool(defer) or want_stat == "defer"      # (defer without want_stat: the data-gradient direction, returns (out, PendingSlabs | None))
    nsplit = lib.miseg_conv3_fwd_splits(B, D, H, W, Cin, Cout, _dt(x)) if (defer_req and fast and res is None and ws is not None) else 1
    defer = nsplit > 1 and D * H * W <= lib.miseg_instnorm_fused_max_rows() and rows(out)[0] == Cout
    if want_stat and fast and not defer:      # (a split reduction computes them in its second launch)
        stat = STAT_POOL.take(lib.miseg_instnorm_stat_bytes(B, Cout) // 8, x.device).view(-1, B, Cout, 2)
    scx, scw, ldsc, Csc = None, None, 0, 0
    if sc is not None:
        scx, scw = sc
        ldsc, nsc, Csc = rows(scx)
        assert fast and nsc == n and scw.is_contiguous() and tuple(scw.shape) == (Cout, Csc) and scw.dtype == x.dtype == scx.dtype
        flops += 2.0 * n * Csc * Cout
        nbytes += float(x.element_size()) * (n * Csc + scw.numel())
    fsw = y2 = stat2 = None
    if fs is not None:
        fsw, want2 = fs
        assert fast and not defer_req and fsw.is_contiguous() and tuple(fsw.shape) == (Cout, Cin) and fsw.dtype == x.dtype
        y2 = torch.empty(B, D, H, W, Cout, dtype=x.dtype, device=x.device)
        if want2:
            stat2 = STAT_POOL.take(lib.miseg_instnorm_stat_bytes(B, Cout) // 8, x.device).view(-1, B, Cout, 2)
        flops += 2.0 * n * Cin * Cout
        nbytes += float(x.element_size()) * (n * Cout + fsw.numel())
    mk = lambda st: L.Conv3(_ptr(x), ld, _ptr(out), rows(out)[0], _ptr(wpk), B, D, H, W, Cin, Cout, _dt(x), _ptr(ws),
                            _ptr(res) if fuse_res else None, rows(res)[0] if fuse_res else 0, _ptr(st), bg, 1 if defer else 0,
                            _ptr(scx), ldsc, _ptr(scw), Csc, _ptr(s2c), (s2c.shape[-1] // 8) if s2c is not None else 0,
                            _ptr(fsw), _ptr(y2), Cout if y2 is not None else 0, _ptr(stat2))
    _call("miseg_conv3_fwd", mk(stat), prof=(name, flops, nbytes))
    if fs is not None:
        assert res is None or fuse_res
        return out, stat, y2, stat2
    if res is not None and not fuse_res:
        out = add(out, res)
    if defer:
        return out, PendingSlabs(ws, nsplit, B * D * H * W * Cout)
    if defer_req and not want_stat:
        return out, None
    return (out, stat) if want_stat else out


class PendingSlabs:
    """what conv3_fwd(..., want_stat="defer") returns in place of the statistics: its output tensor is NOT written yet - the fp32 partial
    slabs of the split launch wait in `ws` for instnorm_fwd_slabs"""
    __slots__ = ("ws", "n", "stride")

    def __init__(self, ws, n, stride):
        self.ws, self.n, self.stride = ws, n, stride


# where the tiny-volume weight gradients are launched (conv3_wgrad): "inline" (default) = where the backward pass reaches the layer; "flush" =
# queued for the end of the backward pass - with a side branch they then run on the BRANCH stream beside the main stream's grouped launch
# (join_branch), without one (and in the two-graph data-parallel step, whose first half must finish these layers) in front of it.
# Round 5, same box: inline 155.2 / 154.5 patches/s, on the branch at the end 154.2 / 154.2 - four write-bound launches (176 MB) beside the
# grouped conv weight gradients lengthen the tail by more than the 80 us they take off the chain
TINY_WGRAD_AT = os.environ.get("MISEG_TINY_WGRAD_AT", "inline")
CONV_WGRAD_GROUP_VOXELS = 48 ** 3   # layers up to this many voxels are queued: alone they fill a fraction of the chip for 40-85 us each


def _flush_conv_wgrads(q, background=0, keep=False):
    """background > 0: the grouped launch walks its units with that many workgroups (miseg_conv3_wgrad_params.max_workgroups of the first
    descriptor) - issued on the side-branch stream beside the main stream's small-grid launches; operands and workspace stay alive until
    join_wgrad (the queue that held them is cleared here, and the allocator knows nothing about the branch stream's reads)"""
    if not q:
        return
    lib = L.load()
    for dt in {it[0].dtype for it in q}:
        items = [it for it in q if it[0].dtype == dt]
        for i in range(0, len(items), 24):
            chunk = items[i:i + 24]
            descs = (L.Conv3Wgrad * len(chunk))()
            for j, (x, dy, dw, acc) in enumerate(chunk):
                B, D, H, W = _vol(x)
                ldx, _, Cin = rows(x)
                lddy, _, Cout = rows(dy)
                descs[j] = L.Conv3Wgrad(_ptr(x), ldx, _ptr(dy), lddy, _ptr(dw), B, D, H, W, Cin, Cout, _dt(x), acc, None, int(background) if j == 0 else 0)
            wsb = lib.miseg_conv3_wgrad_group_workspace_bytes(descs, len(chunk))
            ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=chunk[0][0].device)
            fl = sum(2.0 * 27 * it[0].shape[-1] * it[1].shape[-1] * (it[0].numel() // it[0].shape[-1]) for it in chunk)
            with _ProfRegion("conv3_wgrad_group_kernel" + (" (background)" if background else ""), fl):
                L.check(lib.miseg_conv3_wgrad_group(descs, len(chunk), _ptr(ws), _stream()), "conv3_wgrad_group")
            _stamp_launch("conv3_wgrad_group")
            if background or keep:
                _WGRAD_KEEP.append(ws)
                _WGRAD_KEEP.extend(t for it in chunk for t in it[:2])
    q.clear()


def _conv3_wgrad_now(x, dy, dw, accumulate):
    """a queued single-layer weight gradient, launched now.  The tiny-volume kernel stores straight into dw; the workspace is still sized
    by the library, so that a layer it sends to the slab kernels after all (operands it finds misaligned) has its slabs"""
    B, D, H, W = _vol(x)
    ldx, _, Cin = rows(x)
    lddy, _, Cout = rows(dy)
    ws = torch.empty(max(L.load().miseg_conv3_wgrad_workspace_bytes(B, D, H, W, Cin, Cout) // 4, 1), dtype=torch.float32, device=x.device)
    _call("miseg_conv3_wgrad", L.Conv3Wgrad(_ptr(x), ldx, _ptr(dy), lddy, _ptr(dw), B, D, H, W, Cin, Cout, _dt(x), int(accumulate), _ptr(ws), 0),
          prof=("conv3_wgrad_tiny_kernel", 2.0 * B * D * H * W * 27 * Cin * Cout))


def conv3_wgrad(x, dy, dw=None, accumulate=False):
    """accumulate: False / True, or 2 = `dw` is known to hold zeros (a fresh arena slot): single-producer layers then store instead
    of read-modify-write and the others skip their zero fill."""
    B, D, H, W = _vol(x)
    ldx, n, Cin = rows(x)
    lddy, n2, Cout = rows(dy)
    assert n == n2
    if dw is None:
        dw = torch.empty(Cout, Cin, 3, 3, 3, dtype=torch.float32, device=x.device)
        accumulate = False
    q = _queues(dw) if accumulate else None
    # (narrow bf16 layers - 16 / 32 channels on both sides - have a kernel of their own that finishes a 48^3 layer in ~10 us: never queued)
    narrow = x.dtype == torch.bfloat16 and Cin in (16, 32) and Cout in (16, 32)
    lib = L.load()
    # tiny volumes (3^3 / 6^3, hundreds of channels: encoder10 / decoder5): the write-bound kernel of their own, launched where the backward
    # pass reaches them (TINY_WGRAD_AT = "inline") or with the queued launches at its end ("flush": in front of the grouped launch)
    # (the library's own test also wants 16-byte aligned operands and row strides that are multiples of 8 elements: the same test here)
    tiny = (bool(lib.miseg_conv3_wgrad_tiny(B, D, H, W, Cin, Cout, _dt(x))) and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0
            and ldx % 8 == 0 and lddy % 8 == 0)
    if q is not None and q.unzeroed and dw.data_ptr() in q.unzeroed:
        q.unzeroed.discard(dw.data_ptr())
        if not (tiny and int(accumulate) == 2):      # anything but the overwriting launch reads (or only adds to) the slot: it gets its zeros now
            fill32(dw)
    if tiny and q is not None and TINY_WGRAD_AT != "inline":
        q.lists().tiny_wgrad.append((x, dy, dw, int(accumulate)))
        return dw
    if q is not None and B * D * H * W <= CONV_WGRAD_GROUP_VOXELS and not narrow and not tiny:
        q.lists().conv_wgrad.append((x, dy, dw, int(accumulate)))      # keeps x and dy alive until the flush
        return dw
    if tiny and q is not None and int(accumulate) == 2:
        q.inline_final.append(dw)      # written here, once, by plain stores: nothing queued will touch the slot again
    ws = torch.empty(lib.miseg_conv3_wgrad_workspace_bytes(B, D, H, W, Cin, Cout) // 4, dtype=torch.float32, device=x.device)
    bg = _background()
    _call("miseg_conv3_wgrad", L.Conv3Wgrad(_ptr(x), ldx, _ptr(dy), lddy, _ptr(dw), B, D, H, W, Cin, Cout, _dt(x), int(accumulate), _ptr(ws), bg),
          prof=(f"conv3_wgrad_kernel<{'bf16' if x.dtype == torch.bfloat16 else 'f32'}>" + (" (background)" if bg else ""), 2.0 * B * D * H * W * 27 * Cin * Cout))
    return dw


def conv3_thin_fwd(x_ncdhw, w, dtype):
    B, Cin, D, H, W = x_ncdhw.shape
    assert x_ncdhw.dtype == torch.float32 and x_ncdhw.is_contiguous()
    Cout = w.shape[0]
    y = torch.empty(B, D, H, W, Cout, dtype=dtype, device=x_ncdhw.device)
    _call("miseg_conv3_thin_fwd", L.Conv3Thin(_ptr(x_ncdhw), _ptr(y), Cout, _ptr(_fp32(w)), B, Cin, D, H, W, Cout, _dt(y)))
    return y


def conv3_thin_wgrad(x_ncdhw, dy, dw):
    B, Cin, D, H, W = x_ncdhw.shape
    Cout = dy.shape[-1]
    p = L.Conv3ThinWgrad(_ptr(x_ncdhw), _ptr(dy), rows(dy)[0], _ptr(dw), B, Cin, D, H, W, Cout, _dt(dy), None)
    nb = L.load().miseg_conv3_thin_wgrad_workspace_bytes(C.byref(p))
    ws = torch.empty(nb // 4, dtype=torch.float32, device=dy.device) if nb else None
    p.workspace = _ptr(ws)
    _call("miseg_conv3_thin_wgrad", p)
    return dw


def ncdhw_to_rows(x_ncdhw, dtype):
    """NCDHW fp32 image -> channels-last rows with one 16-byte vector per voxel (channels >= Cin zero)."""
    B, Cin, D, H, W = x_ncdhw.shape
    assert x_ncdhw.dtype == torch.float32 and x_ncdhw.is_contiguous()
    CP = 16 // torch.empty(0, dtype=dtype).element_size()
    y = torch.empty(B, D, H, W, CP, dtype=dtype, device=x_ncdhw.device)
    lib = L.load()
    L.check(lib.miseg_ncdhw_to_rows(_ptr(x_ncdhw), _ptr(y), B, Cin, D * H * W, CP, _dt(y), _stream()), "ncdhw_to_rows")
    return y


def resample2(x, up, fine_shape=None):
    """up=False: keep the even voxels of x [B,D,H,W,C] (the stride-2 pick); up=True: zero insertion into a grid of
    fine_shape = (D, H, W) (ceil(./2) must be x's grid)."""
    B, d, h, w = _vol(x)
    ld, n, Cc = rows(x)
    if up:
        D, H, W = fine_shape
        assert ((D + 1) // 2, (H + 1) // 2, (W + 1) // 2) == (d, h, w), (fine_shape, x.shape)
        out = torch.empty(B, D, H, W, Cc, dtype=x.dtype, device=x.device)
    else:
        D, H, W = d, h, w
        out = torch.empty(B, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, Cc, dtype=x.dtype, device=x.device)
    _call("miseg_resample2", L.Resample2(_ptr(x), ld, _ptr(out), rows(out)[0], B, D, H, W, Cc, _dt(x), int(up)))
    return out


def rowbias_add(x, bias):
    ld, n, Cc = rows(x)
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _call("miseg_rowbias_add", L.Rowbias(_ptr(x), ld, _ptr(_fp32(bias)), _ptr(y), rows(y)[0], n, Cc, _dt(x)))
    return y


def prelu_fwd(x, slope):
    ld, n, Cc = rows(x)
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _call("miseg_prelu_fwd", L.PreluFwd(_ptr(x), ld, _ptr(_fp32(slope)), _ptr(y), rows(y)[0], n, Cc, _dt(x)))
    return y


def prelu_bwd(dy, x, slope, dslope):
    ld, n, Cc = rows(x)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    scratch = STAT_POOL.take(2, x.device) if dslope is not None else None      # zeroed fp64 (sum, arrival ticket) of the slope-gradient reduction
    _call("miseg_prelu_bwd", L.PreluBwd(_ptr(dy), rows(dy)[0], _ptr(x), ld, _ptr(_fp32(slope)), _ptr(dx), rows(dx)[0], _ptr(dslope), n, Cc, _dt(x), _ptr(scratch)))
    return dx


def rows_to_ncdhw(x):
    """channels-last [B,D,H,W,C] in the compute dtype -> NCDHW fp32 (the network output layout)."""
    B, D, H, W = _vol(x)
    ld, n, Cc = rows(x)
    y = torch.empty(B, Cc, D, H, W, dtype=torch.float32, device=x.device)
    lib = L.load()
    L.check(lib.miseg_layout_ncdhw(_ptr(x), ld, _ptr(y), B, Cc, D * H * W, _dt(x), 0, _stream()), "layout_ncdhw")
    return y


def ncdhw_to_rows_exact(g, dtype):
    """NCDHW fp32 -> channels-last rows of exactly C channels in `dtype` (gradient of rows_to_ncdhw)."""
    B, Cc, D, H, W = g.shape
    assert g.dtype == torch.float32 and g.is_contiguous()
    y = torch.empty(B, D, H, W, Cc, dtype=dtype, device=g.device)
    lib = L.load()
    L.check(lib.miseg_layout_ncdhw(_ptr(y), Cc, _ptr(g), B, Cc, D * H * W, _dt(y), 1, _stream()), "layout_ncdhw")
    return y


def im2col3(x, adjoint=False, C_out=None):
    """forward: x [B,D,H,W,C] -> col [B,D,H,W,27*C];  adjoint: col -> [B,D,H,W,C]."""
    B, D, H, W = _vol(x)
    ld, n, Cc = rows(x)
    if not adjoint:
        out = torch.empty(B, D, H, W, 27 * Cc, dtype=x.dtype, device=x.device)
        _call("miseg_im2col3", L.Im2col3(_ptr(x), ld, _ptr(out), 27 * Cc, B, D, H, W, Cc, _dt(x)))
    else:
        Co = Cc // 27
        out = torch.empty(B, D, H, W, Co, dtype=x.dtype, device=x.device)
        _call("miseg_col2im3", L.Im2col3(_ptr(x), ld, _ptr(out), Co, B, D, H, W, Co, _dt(x)))
    return out


# ------------------------------------------------------------------------------------------ attention
def winattn_params(qkv, out, qkv_bias, table, lse, heads, window, shift, tw, scale, drop=None):
    """drop = (p, (seed, stream_id, step_dev)) puts dropout on the attention probabilities; the key is DROP.next_key's"""
    B, D, H, W = _vol(qkv)
    ldq, n, C3 = rows(qkv)
    Cc = C3 // 3
    dp, (seed, sid, step) = drop if drop else (0.0, (0, 0, None))
    return L.Winattn(_ptr(qkv), ldq, _ptr(out), rows(out)[0], _ptr(qkv_bias), _ptr(table), _ptr(lse), B, D, H, W, Cc, heads, _dt(qkv),
                     window[0], window[1], window[2], shift[0], shift[1], shift[2], tw, scale, float(dp), seed, sid, _ptr(step))


def winattn_fwd(qkv, qkv_bias, table, heads, window, shift, tw, scale, drop=None):
    B, D, H, W = _vol(qkv)
    Cc = qkv.shape[-1] // 3
    out = torch.empty(B, D, H, W, Cc, dtype=qkv.dtype, device=qkv.device)
    nw = B * -(-D // window[0]) * -(-H // window[1]) * -(-W // window[2])
    lse = torch.empty(nw, heads, window[0] * window[1] * window[2], dtype=torch.float32, device=qkv.device)
    n = window[0] * window[1] * window[2]
    _call("miseg_winattn_fwd", winattn_params(qkv, out, qkv_bias, table, lse, heads, window, shift, tw, scale, drop),
          prof=("winattn", 4.0 * nw * heads * n * n * (Cc // heads), _nb(qkv, out, lse)))
    return out, lse


def winattn_bwd(qkv, out, lse, dout, qkv_bias, table, heads, window, shift, tw, scale, dqkv_bias, dtable, drop=None):
    dqkv = torch.empty(qkv.shape, dtype=qkv.dtype, device=qkv.device)
    f = winattn_params(qkv, out, qkv_bias, table, lse, heads, window, shift, tw, scale, drop)
    mk = lambda dqb, dtb: L.WinattnBwd(f, _ptr(dout), rows(dout)[0], _ptr(dqkv), rows(dqkv)[0], _ptr(dqb), _ptr(dtb))
    B, D, H, W = _vol(qkv)
    n = window[0] * window[1] * window[2]
    nw = B * -(-D // window[0]) * -(-H // window[1]) * -(-W // window[2])
    sq, st_ = _prof_scratch(dqkv_bias), _prof_scratch(dtable)      # (roofline leg: the repeats accumulate the bias / table gradients into scratch)
    _call("miseg_winattn_bwd", mk(dqkv_bias, dtable), prof=("winattn", 10.0 * nw * heads * n * n * (qkv.shape[-1] // 3 // heads), _nb(qkv, out, lse, dout, dqkv)),
          prof_params=mk(sq, st_) if (sq is not None or st_ is not None) else None)
    return dqkv


# ------------------------------------------------------------------------------------------ elementwise / movement
def add(a, b, out=None):
    lda, n, Cc = rows(a)
    y = out if out is not None else torch.empty(a.shape, dtype=a.dtype, device=a.device)
    _call("miseg_add", L.Add(_ptr(a), lda, _ptr(b), rows(b)[0], _ptr(y), rows(y)[0], n, Cc, _dt(a)))
    return y


def copy2d(src, dst):
    lds, n, Cc = rows(src)
    ldd, n2, C2 = rows(dst)
    assert n == n2 and Cc == C2
    _call("miseg_copy2d", L.Copy2d(_ptr(src), lds, _dt(src), _ptr(dst), ldd, _dt(dst), n, Cc))
    return dst


def gelu_fwd(x):
    ld, n, Cc = rows(x)
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _call("miseg_gelu_fwd", L.GeluFwd(_ptr(x), ld, _ptr(y), rows(y)[0], n, Cc, _dt(x)))
    return y


def gelu_bwd(dy, x):
    ld, n, Cc = rows(x)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _call("miseg_gelu_bwd", L.GeluBwd(_ptr(dy), rows(dy)[0], _ptr(x), ld, _ptr(dx), rows(dx)[0], n, Cc, _dt(x)))
    return dx


def _offs(offsets):
    arr = (C.c_int8 * 24)()
    flat = [v for o in offsets for v in o]
    assert len(flat) == 24
    for i, v in enumerate(flat):
        arr[i] = v
    return arr


def space_to_channel(fine, offsets, out=None):
    """fine [B,D,H,W,C] -> coarse [B,ceil(D/2),ceil(H/2),ceil(W/2),8C] with block j taken at offsets[j]."""
    B, D, H, W = _vol(fine)
    ld, n, Cc = rows(fine)
    if out is None:
        out = torch.empty(B, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, 8 * Cc, dtype=fine.dtype, device=fine.device)
    _call("miseg_space_to_channel", L.S2C(_ptr(fine), ld, _ptr(out), rows(out)[0], B, D, H, W, Cc, _dt(fine), _offs(offsets)))
    return out


def channel_to_space(coarse, offsets, fine_shape, out=None):
    """adjoint of space_to_channel: coarse [B,D2,H2,W2,8C] -> fine [B,D,H,W,C] (sum over blocks referencing a voxel)."""
    B, D, H, W, Cc = fine_shape
    if out is None:
        out = torch.empty(B, D, H, W, Cc, dtype=coarse.dtype, device=coarse.device)
    _call("miseg_channel_to_space", L.S2C(_ptr(coarse), rows(coarse)[0], _ptr(out), rows(out)[0], B, D, H, W, Cc, _dt(coarse), _offs(offsets)))
    return out


def patch_embed_fwd(x_ncdhw, w, bias, dtype):
    B, Cin, D, H, W = x_ncdhw.shape
    assert x_ncdhw.dtype == torch.float32 and x_ncdhw.is_contiguous()
    Cout = w.shape[0]
    y = torch.empty(B, D // 2, H // 2, W // 2, Cout, dtype=dtype, device=x_ncdhw.device)
    _call("miseg_patch_embed_fwd", L.PatchEmbed(_ptr(x_ncdhw), _ptr(y), Cout, _ptr(_fp32(w)), _ptr(_fp32(bias)), B, Cin, D, H, W, Cout, _dt(y)))
    return y


def patch_embed_bwd(x_ncdhw, dy, dw, dbias):
    B, Cin, D, H, W = x_ncdhw.shape
    Cout = dy.shape[-1]
    p = L.PatchEmbedBwd(_ptr(x_ncdhw), _ptr(dy), rows(dy)[0], _ptr(dw), _ptr(dbias), B, Cin, D, H, W, Cout, _dt(dy), None)
    ws = torch.empty(L.load().miseg_patch_embed_bwd_workspace_bytes(C.byref(p)) // 4, dtype=torch.float32, device=dy.device)
    p.workspace = ws.data_ptr()
    _call("miseg_patch_embed_bwd", p)


def head_fwd(x, w, bias):
    B, D, H, W = _vol(x)
    ld, n, Cin = rows(x)
    Cout = w.shape[0]
    y = torch.empty(B, Cout, D, H, W, dtype=torch.float32, device=x.device)
    _call("miseg_head_fwd", L.Head(_ptr(x), ld, _ptr(y), _ptr(_fp32(w)), _ptr(_fp32(bias)), B, D * H * W, Cin, Cout, _dt(x)))
    return y


def head_bwd(x, dy_ncdhw, w, dw, dbias, want_dx=True):
    B, D, H, W = _vol(x)
    ld, n, Cin = rows(x)
    Cout = w.shape[0]
    assert dy_ncdhw.dtype == torch.float32 and dy_ncdhw.is_contiguous()
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_dx else None
    _call("miseg_head_bwd", L.HeadBwd(_ptr(x), ld, _ptr(dy_ncdhw), _ptr(dx), rows(dx)[0] if dx is not None else 0, _ptr(_fp32(w)), _ptr(dw), _ptr(dbias),
                                      B, D * H * W, Cin, Cout, _dt(x)))
    return dx


def fill32(t, word=0):
    lib = L.load()
    assert t.is_contiguous() and t.element_size() == 4
    L.check(lib.miseg_fill32(_ptr(t), word, t.numel(), _stream()), "fill32")
    return t


def fill32_ranges(t, ranges, word=0):
    """t[off : off + n] = word for every (off, n) of `ranges` (element offsets into the contiguous 4-byte tensor t, multiples of 4) in one launch
    per L.FILL_RANGES ranges"""
    lib = L.load()
    assert t.is_contiguous() and t.element_size() == 4
    ranges = [(int(o), int(n)) for o, n in ranges if n > 0]
    for i in range(0, len(ranges), L.FILL_RANGES):
        chunk = ranges[i:i + L.FILL_RANGES]
        arr = (C.c_uint64 * (2 * len(chunk)))(*[v for r in chunk for v in r])
        L.check(lib.miseg_fill32_ranges(_ptr(t), word, arr, len(chunk), _stream()), "fill32_ranges")
    return t


def zeros_f32(shape, device):
    return fill32(torch.empty(shape, dtype=torch.float32, device=device))


# ------------------------------------------------------------------------------------------ after the path: loss, metric, optimiser, stitching
def _label(label):
    """class-id labels [B, 1, ...] (any of float32 / int32 / int64 / uint8, as the data pipeline delivers them) -> (tensor, MISEG_LABEL_*)"""
    kinds = {torch.float32: L.LABEL_F32, torch.int32: L.LABEL_I32, torch.int64: L.LABEL_I64, torch.uint8: L.LABEL_U8}
    if label.dtype not in kinds:
        label = label.to(torch.int32)
    return label.contiguous(), kinds[label.dtype]


class SegLossCfg:
    """DiceFocalLoss / DiceCELoss hyper-parameters as LitMonai builds them (reference lightning_monai.py:48-65)"""

    def __init__(self, kind, include_background, squared_pred, smooth_nr, smooth_dr, gamma=2.0, lambda_dice=1.0, lambda_other=1.0):
        self.kind, self.include_background, self.squared_pred = kind, bool(include_background), bool(squared_pred)
        self.smooth_nr, self.smooth_dr, self.gamma, self.lambda_dice, self.lambda_other = float(smooth_nr), float(smooth_dr), float(gamma), float(lambda_dice), float(lambda_other)


def _seg_loss_params(logits, label, cfg, sums, ws):
    if not logits.is_cuda:
        raise L.MisegHipError("miseg ops need CUDA/HIP tensors: the MI355X path has no CPU fallback")
    if logits.dtype != torch.float32 or not logits.is_contiguous() or logits.dim() < 3:
        raise ValueError("seg_loss: logits must be contiguous float32 [B, C, ...]")
    B, Cc = logits.shape[0], logits.shape[1]
    S = logits.numel() // (B * Cc)
    if label.numel() != B * S:
        raise ValueError(f"seg_loss: label {tuple(label.shape)} does not match logits {tuple(logits.shape)} ([B, 1, ...] class ids)")
    lab, ldt = _label(label)
    p = L.SegLoss(C.sizeof(L.SegLoss), cfg.kind, _ptr(logits), _ptr(lab), ldt, B, Cc, S, int(cfg.include_background), int(cfg.squared_pred), cfg.smooth_nr,
                  cfg.smooth_dr, cfg.gamma, cfg.lambda_dice, cfg.lambda_other, _ptr(ws), _ptr(sums), None, None, None)
    return p, lab, (B, Cc, S)


def seg_loss_fwd(logits, label, cfg):
    """-> (loss: 0-dim fp32 device tensor, sums: fp64 [3 B C + 1] saved for the backward pass)"""
    B, Cc = logits.shape[0], logits.shape[1]
    S = logits.numel() // (B * Cc)
    lib = L.load()
    ws = torch.empty(lib.miseg_seg_loss_workspace_bytes(B, Cc, S) // 8, dtype=torch.float64, device=logits.device)
    sums = torch.empty(3 * B * Cc + 1, dtype=torch.float64, device=logits.device)
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    p, lab, _ = _seg_loss_params(logits, label, cfg, sums, ws)
    p.loss = _ptr(loss)
    _call("miseg_seg_loss_fwd", p)
    return loss, sums


def seg_loss_bwd(logits, label, cfg, sums, gscale=None):
    """d(loss)/d(logits) * gscale (a 0-dim fp32 device tensor or None)"""
    dlogits = torch.empty_like(logits)
    ws = torch.empty(1, dtype=torch.float64, device=logits.device)      # not used by the backward pass
    p, lab, _ = _seg_loss_params(logits, label, cfg, sums, ws)
    if gscale is not None:
        gscale = gscale.to(torch.float32).contiguous()
        p.gscale = _ptr(gscale)
    p.dlogits = _ptr(dlogits)
    _call("miseg_seg_loss_bwd", p)
    return dlogits


def dice_metric(logits, label):
    """[B, C] Dice per class after argmax (NaN where the class is absent from the label)"""
    if logits.dtype != torch.float32 or not logits.is_contiguous():
        raise ValueError("dice_metric: logits must be contiguous float32 [B, C, ...]")
    B, Cc = logits.shape[0], logits.shape[1]
    S = logits.numel() // (B * Cc)
    if label.numel() != B * S:
        raise ValueError("dice_metric: label does not match logits")
    lab, ldt = _label(label)
    counts = torch.empty(B * Cc * 3, dtype=torch.int64, device=logits.device)
    dice = torch.empty(B, Cc, dtype=torch.float32, device=logits.device)
    _call("miseg_dice_metric", L.DiceMetric(C.sizeof(L.DiceMetric), _ptr(logits), _ptr(lab), ldt, B, Cc, S, _ptr(counts), _ptr(dice)))
    return dice


def stitch_windows(win, out, starts, roi, count=None, slab=None):
    """win fp32 [nd*nh*nw, C, rd, rh, rw] (the windows of the nd depth layers `starts[0]`, all resident), out fp32 [C, D, H, W];
    starts = (list_d, list_h, list_w).  slab = (d_begin, d_count): write only these depths of `out` from the resident layers (which then
    need not cover the whole depth axis, miseg_stitch_params)."""
    nd, nh, nw = (len(s) for s in starts)
    Cc, D, H, W = out.shape
    assert win.dtype == torch.float32 and out.dtype == torch.float32 and win.is_contiguous() and out.is_contiguous()
    if tuple(win.shape) != (nd * nh * nw, Cc) + tuple(roi):
        raise ValueError(f"stitch_windows: window buffer {tuple(win.shape)} does not match {nd}x{nh}x{nw} windows of {tuple(roi)} x {Cc} channels")
    arr = [(C.c_int32 * len(s))(*s) for s in starts]
    if count is not None:
        assert count.dtype == torch.int16 and count.is_contiguous() and tuple(count.shape) == (D, H, W)
    d0, dn = slab if slab is not None else (0, 0)
    _call("miseg_stitch_windows", L.Stitch(C.sizeof(L.Stitch), _ptr(win), _ptr(out), _ptr(count), Cc, D, H, W, roi[0], roi[1], roi[2], nd, nh, nw,
                                           C.cast(arr[0], C.c_void_p), C.cast(arr[1], C.c_void_p), C.cast(arr[2], C.c_void_p), int(d0), int(dn)))
    return out


# ------------------------------------------------------------------------------------------ dropout / stochastic depth
class _DropState:
    """keys of the counter-based dropout masks: `seed` (host, torch.initial_seed() unless set), a call counter that tells the call sites
    of one step apart (reset by begin_step / begin_forward) and a DEVICE step counter advanced once per step - under hipGraph replay the
    first two are baked into the graph, the third is what makes every replay draw new masks."""

    def __init__(self):
        self.seed, self.calls, self.step_dev = None, 0, None

    def next_key(self, device):
        if self.seed is None:
            self.seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        if self.step_dev is None or self.step_dev.device != device:
            self.step_dev = torch.zeros(1, dtype=torch.int64, device=device)
        self.calls += 1
        # the step counter is SNAPSHOT per call: the backward pass of this forward may run after a later forward advanced the counter
        snap = torch.empty(1, dtype=torch.int64, device=device)
        L.check(L.load().miseg_counter_copy(_ptr(snap), _ptr(self.step_dev), _stream()), "counter_copy")
        return self.seed, self.calls, snap

    def advance(self):
        """once per training step (arena.begin_step / GraphedStep._run / the nets' forward when nobody manages the step)"""
        self.calls = 0
        if self.step_dev is not None:
            L.check(L.load().miseg_counter_add(_ptr(self.step_dev), 1, _stream()), "counter_add")


DROP = _DropState()


def dropout_apply(x, p, key, rows_per_sample=0):
    """y = x * mask(key) / (1 - p); key = (seed, stream_id, step_dev) from DROP.next_key: the same key reproduces the same mask (backward)"""
    ld, n, Cc = rows(x)
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    seed, sid, step = key
    _call("miseg_dropout", L.Dropout(C.sizeof(L.Dropout), _ptr(x), ld, _ptr(y), rows(y)[0], n, Cc, _dt(x), rows_per_sample, float(p), seed, sid, _ptr(step)))
    return y
