"""torch.autograd glue over the C-ABI kernels.  Every forward AND backward arithmetic op below is a HIP kernel
from csrc/ (see hip/ops.py); torch supplies tensors, the caching allocator, streams and the autograd tape only.

Layout contract: activations are channels-last ([B, D, H, W, C] or [B, L, C]) in the compute dtype (float32 for the
parity mode, bfloat16 for throughput); parameters and their gradients are float32.
"""
import os

import torch
from torch.autograd import Function

from . import lib as L
from . import ops

STD_OFFSETS = tuple((a, b, c) for a in range(2) for b in range(2) for c in range(2))
"""(dz,dy,dx) of block j = 4a+2b+c: ConvTranspose3d k2 s2 taps and PatchMergingV2's itertools.product order."""
MERGE_V1_OFFSETS = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1))
"""reference networks/blocks/patch_merging.py:120-127 (v0.9 PatchMerging; slots 5,6 duplicate 2,3)."""


def _rv(t):
    """make a gradient usable as a channels-last row view (autograd may hand us expanded / permuted tensors)."""
    if t is None:
        return None
    try:
        ops.rows(t)
        return t
    except ValueError:
        return t.contiguous()


def _slot(p):
    """the arena slot (runtime/arena.py) that the weight-gradient kernels add into for parameter p, or None."""
    g = getattr(p, "_miseg_grad", None) if p is not None else None
    if g is not None:
        p._miseg_used = True
    return g


def _slot_first(p):
    """(slot, accumulate mode): 2 when this is the first write of the step into the slot (it still holds the zeros of
    arena.begin_step, which resets `_miseg_used` together with the fill), else 1."""
    g = getattr(p, "_miseg_grad", None) if p is not None else None
    if g is None:
        return None, 0
    first = not p._miseg_used
    p._miseg_used = True
    return g, 2 if first else 1


# ----------------------------------------------------------------------------------------------------------------
class _Fork(Function):
    """y1 = y2 = x with the gradient sum done by our add kernel instead of the autograd engine's."""

    @staticmethod
    def forward(ctx, x):
        # an output nobody differentiates (the ViT's unused hidden states) must arrive as None, not as a zero tensor autograd fills in for us:
        # that was a fill + an add launch per unused fork (9 of each per C-UNETR step)
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None:
            return g2
        if g2 is None:
            return g1
        return ops.add(_rv(g1), _rv(g2))


def fork(x):
    return _Fork.apply(x)


class _Mark(Function):
    """identity whose backward calls `fn()` (host code; no kernel) when the backward pass gets here: placed at the INPUT of a sub-network it
    fires right behind that sub-network's backward pass - autograd runs the ready node with the highest sequence number first - where a
    tensor hook would only fire in front of the input's own producer, which may be much later (runtime/graph.py: leaf-graph points)"""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.fn()
        return g, None


def backward_mark(x, fn):
    if fn is None or not x.requires_grad:
        return x
    y = _Mark.apply(x, fn)
    for a in ("_miseg_stat", "_miseg_cat"):      # tags the producer hung on the tensor travel with it
        if hasattr(x, a):
            setattr(y, a, getattr(x, a))
    return y


class _InnerTape(Function):
    """outputs computed EARLIER on a tape of their own (from detached leaves) enter the outer tape HERE.  Autograd runs the ready node with the
    highest sequence number first, so where a node is taped decides when its backward pass is issued; a side branch whose forward should
    run early (beside one launch-bound part of the step) and whose backward should run early too (beside another) cannot be taped where its
    forward runs.  forward: aliases of the inner outputs; backward: the inner tape's backward pass (a nested autograd call on the stream
    this node was taped on) and the gradients its leaves collected."""

    @staticmethod
    def forward(ctx, holder, *outer):
        ctx.holder = holder
        return tuple(o.detach() for o in holder["outs"])

    @staticmethod
    def backward(ctx, *g):
        h = ctx.holder
        pairs = [(o, gi) for o, gi in zip(h["outs"], g) if gi is not None and o.requires_grad]
        if pairs:
            torch.autograd.backward([o for o, _ in pairs], [gi for _, gi in pairs])
        grads = []
        for leaf in h["leaves"]:
            grads.append(leaf.grad if leaf is not None else None)
            if leaf is not None:
                leaf.grad = None
        h["outs"] = h["leaves"] = None          # the inner tape is spent
        return (None,) + tuple(grads)


def inner_tape_leaf(t):
    """a leaf that starts an inner tape at `t` (see _InnerTape); tags travel with it"""
    l = t.detach().requires_grad_(t.requires_grad)
    for a in ("_miseg_stat", "_miseg_cat"):
        if hasattr(t, a):
            setattr(l, a, getattr(t, a))
    return l


def inner_tape_join(outs, outer, leaves):
    """outs: tensors computed from `leaves` (= inner_tape_leaf(o) for o in outer; None where the outer tensor needs no gradient).  Returns
    aliases of `outs` that are taped at the CALLER's position, on the caller's current stream."""
    holder = {"outs": list(outs), "leaves": list(leaves)}
    res = _InnerTape.apply(holder, *outer)
    for r, o in zip(res, outs):
        for a in ("_miseg_stat", "_miseg_cat"):
            if hasattr(o, a):
                setattr(r, a, getattr(o, a))
    return res


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


class _Dropout(Function):
    """nn.Dropout / DropPath on channels-last rows: the mask is a counter-based hash of (seed, call site, step, index) - nothing is stored,
    the backward pass is the same kernel call on the gradient (swin_transformer_block.py:90-97,205,247)."""

    @staticmethod
    def forward(ctx, x, p, per_sample):
        ctx.key = ops.DROP.next_key(x.device)
        ctx.p = p
        ctx.rps = (ops.rows(x)[1] // x.shape[0]) if per_sample else 0
        return ops.dropout_apply(x, p, ctx.key, ctx.rps)

    @staticmethod
    def backward(ctx, g):
        return ops.dropout_apply(_rv(g), ctx.p, ctx.key, ctx.rps), None, None


def dropout(x, p, training=True):
    """elementwise dropout (identity in eval mode or at p == 0)"""
    return _Dropout.apply(x, float(p), False) if (training and p > 0.0) else x


def drop_path(x, p, training=True):
    """stochastic depth: the whole residual branch of a sample is dropped with probability p (MONAI DropPath, scale_by_keep)"""
    return _Dropout.apply(x, float(p), True) if (training and p > 0.0) else x


# ----------------------------------------------------------------------------------------------------------------
_PENDING_OUT = None      # see _InstNorm.forward
_LAST_STAT = None          # statistics buffer of the _InstNorm.forward that has just run (instance_norm(fork=True))
FORK_KEEPS_STAT = os.environ.get("MISEG_NO_FORK_STAT") is None      # A/B switch of round 5


class _InstNorm(Function):
    """(conditional) instance norm over all rows of each sample, + optional residual, + optional LeakyReLU.
    reference: networks/norms/conditional_instance_norm.py:59-68, dynunet_block.py:100-126."""

    @staticmethod
    def forward(ctx, x, res, styles_dev, styles_host, num_styles, affine, act, slope, eps, fork, stat_in, *params):
        """fork=True additionally returns x itself (the skip branch of `x + f(norm(x))`): the backward then receives both
        gradients at once and the fan-out sum rides in the norm-backward kernel instead of a separate add."""
        B = x.shape[0]
        S = ops.rows(x)[1] // B
        gammas = list(params[0::2]) if affine else None
        betas = list(params[1::2]) if affine else None
        # out: a caller-provided rows view to write into (the skip half of a decoder's concat buffer: no copy later).  It travels
        # beside the autograd inputs: to autograd the result is an ordinary fresh output, not an aliased / in-place-modified input
        global _PENDING_OUT
        out, _PENDING_OUT = _PENDING_OUT, None
        if isinstance(stat_in, ops.PendingSlabs):      # x is the not-yet-summed output of a split convolution: one launch finishes both
            y, stat = ops.instnorm_fwd_slabs(x, stat_in, B, S, styles_dev, gammas, betas, res=res, act=act, slope=slope, eps=eps, out=out)
        elif stat_in is not None:      # statistics already produced by the epilogue of the kernel that wrote x
            stat = stat_in
            y = ops.instnorm_apply(x, B, S, stat, styles_dev, gammas, betas, res=res, act=act, slope=slope, eps=eps, out=out)
        else:
            y, stat = ops.instnorm_fwd(x, B, S, styles_dev, gammas, betas, res=res, act=act, slope=slope, eps=eps, out=out)
        global _LAST_STAT
        _LAST_STAT = stat if isinstance(stat, torch.Tensor) else None      # (instance_norm(fork=True) hangs it on the skip branch)
        ctx.meta = (B, S, styles_host, num_styles, affine, act, slope, res is not None, eps)
        ctx.params = params
        # y is kept only where a residual entered the activation: otherwise the backward kernels recompute the LeakyReLU's sign from x
        keep_y = act != L.ACT_NONE and res is not None
        ctx.nbeta = len(betas) if (betas and act != L.ACT_NONE and not keep_y) else 0
        ctx.save_for_backward(x, y if keep_y else None, stat, styles_dev, *(gammas or []), *(betas[:ctx.nbeta] if ctx.nbeta else []))
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, gskip=None):
        B, S, styles_host, num_styles, affine, act, slope, has_res, eps = ctx.meta
        x, y, stat, styles_dev, *gb = ctx.saved_tensors
        gammas, betas = (gb[:len(gb) - ctx.nbeta], gb[len(gb) - ctx.nbeta:]) if ctx.nbeta else (gb, None)
        dy = _rv(dy)
        C = x.shape[-1]
        present = sorted(set(styles_host)) if styles_host is not None else [0]
        dgam = dbet = None
        in_arena = affine and getattr(ctx.params[0], "_miseg_grad", None) is not None
        if in_arena:
            dgam = [_slot(ctx.params[2 * s]) if s in present else None for s in range(num_styles)]
            dbet = [_slot(ctx.params[2 * s + 1]) if s in present else None for s in range(num_styles)]
        elif affine:
            buf = ops.zeros_f32((num_styles, 2, C), x.device)
            # parameters of a style absent from the batch get no gradient (reference: grad is None)
            dgam = [buf[s, 0] if s in present else None for s in range(num_styles)]
            dbet = [buf[s, 1] if s in present else None for s in range(num_styles)]
        dx, dres = ops.instnorm_bwd(dy, y, x, B, S, stat, styles_dev, gammas if affine else None, dgam, dbet, act=act, slope=slope, eps=eps,
                                    want_dres=has_res and ctx.needs_input_grad[1], gadd=_rv(gskip), betas=betas,
                                    pending=ops.pending_dx_take(dy))
        pg = []
        if affine:
            for s in range(num_styles):
                pg += [None, None] if in_arena else [dgam[s], dbet[s]]
        return (dx, dres, None, None, None, None, None, None, None, None, None, *pg)


class _ResNormPair(Function):
    """y = LeakyReLU(norm_a(xa) + norm_b(xb)): the tail of a UnetResBlock with a 1x1x1 shortcut conv (dynunet_block.py:118-124) in ONE
    apply pass - the shortcut branch is normalised on the fly instead of being written and read back.  Backward: the two ordinary
    instance-norm backward passes (the first hands the activation-masked gradient to the second)."""

    @staticmethod
    def forward(ctx, xa, xb, w1, styles_dev, styles_host, num_styles, affine, slope, eps_a, eps_b, stat_a, stat_b, *params):
        """w1 (rank-1 mode, one sample): xb is the ONE-channel image [1, D, H, W, 1] and w1 the weight [C, 1, 1, 1, 1] of the 1x1x1 shortcut
        convolution in front of norm_b (the stem block): the convolution's output round(xb * w1[c]) is formed inside the norm kernels and
        never stored, neither is its gradient - the backward kernel reduces the weight gradient itself."""
        B = xa.shape[0]
        S = ops.rows(xa)[1] // B
        global _PENDING_OUT
        out, _PENDING_OUT = _PENDING_OUT, None
        na = 2 * num_styles if affine else 0
        pa, pb = params[:na], params[na:]
        ga, ba = (list(pa[0::2]), list(pa[1::2])) if affine else (None, None)
        gb, bb = (list(pb[0::2]), list(pb[1::2])) if affine else (None, None)
        if eps_a != eps_b:
            raise ValueError("the two norms of a residual pair must share eps")
        sa = stat_a if stat_a is not None else ops.instnorm_stats(xa, B, S)
        if w1 is not None:
            if B != 1 or xb.shape[-1] != 1 or xb.requires_grad:
                raise ValueError("rank-1 shortcut: one sample, a one-channel image without gradient")
            wb = ops.cast_matrix(w1, xa.dtype)
            sb = ops.rank1_stats(xb, wb)
            y = ops.instnorm_apply(xa, B, S, sa, styles_dev, ga, ba, act=L.ACT_LEAKY, slope=slope, eps=eps_a, out=out, res_stat=sb,
                                   res_gammas=gb, res_betas=bb, r1=(xb, wb))
        else:
            sb = stat_b if stat_b is not None else ops.instnorm_stats(xb, B, S)
            y = ops.instnorm_apply(xa, B, S, sa, styles_dev, ga, ba, res=xb, act=L.ACT_LEAKY, slope=slope, eps=eps_a, out=out, res_stat=sb,
                                   res_gammas=gb, res_betas=bb)
        ctx.meta = (B, S, styles_host, num_styles, affine, slope, eps_a)
        ctx.params = params
        ctx.w1 = w1
        # y is not kept: the backward kernels recompute the LeakyReLU's sign from xa / xb (the forward's own expression), hence the betas
        ctx.save_for_backward(xa, xb, sa, sb, styles_dev, *(ga or []), *(gb or []), *(ba or []), *(bb or []))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, S, styles_host, num_styles, affine, slope, eps = ctx.meta
        xa, xb, sa, sb, styles_dev, *gg = ctx.saved_tensors
        ga, gb, ba, bb = (gg[i * num_styles:(i + 1) * num_styles] for i in range(4)) if affine else (None, None, None, None)
        dy = _rv(dy)
        C = xa.shape[-1]
        present = sorted(set(styles_host)) if styles_host is not None else [0]
        na = 2 * num_styles if affine else 0
        pa, pb = ctx.params[:na], ctx.params[na:]

        def grads(ps):
            if not affine:
                return None, None, False, None
            in_arena = getattr(ps[0], "_miseg_grad", None) is not None
            if in_arena:
                return ([_slot(ps[2 * s]) if s in present else None for s in range(num_styles)],
                        [_slot(ps[2 * s + 1]) if s in present else None for s in range(num_styles)], True, None)
            buf = ops.zeros_f32((num_styles, 2, C), xa.device)
            return ([buf[s, 0] if s in present else None for s in range(num_styles)],
                    [buf[s, 1] if s in present else None for s in range(num_styles)], False, buf)

        dga, dba, arena_a, _ = grads(pa)
        dgb, dbb, arena_b, _ = grads(pb)
        dw1 = None
        if ctx.w1 is not None:
            slot = _slot(ctx.w1)
            dwb = slot if slot is not None else ops.zeros_f32((C,), xa.device)
            dxa, dxb = ops.instnorm_pair_bwd(dy, None, xa, None, B, S, sa, sb, styles_dev, ga, gb, dga, dba, dgb, dbb, slope=slope, eps=eps, betas_a=ba,
                                             betas_b=bb, r1=(xb, ops.cast_matrix(ctx.w1, xa.dtype), dwb))
            dw1 = None if slot is not None else dwb.view(ctx.w1.shape)
        else:
            dxa, dxb = ops.instnorm_pair_bwd(dy, None, xa, xb, B, S, sa, sb, styles_dev, ga, gb, dga, dba, dgb, dbb, slope=slope, eps=eps, betas_a=ba,
                                             betas_b=bb)
        pg = []
        if affine:
            for dg_, db_, ar in ((dga, dba, arena_a), (dgb, dbb, arena_b)):
                for s in range(num_styles):
                    pg += [None, None] if ar else [dg_[s], db_[s]]
        return (dxa, dxb, dw1, None, None, None, None, None, None, None, None, None, *pg)


def _carried_stat(x):
    """statistics the GEMM that produced x left on it (linear / mlp / conv1 with want_stat): valid for exactly this tensor"""
    st = getattr(x, "_miseg_stat", None)
    if st is not None and x.shape[0] == 1 and st.shape[1] == 1 and st.shape[2] == x.shape[-1]:
        return st
    return None


def res_norm_pair(xa, xb, params_a, params_b, styles_dev=None, styles_host=None, slope=0.01, eps_a=1e-5, eps_b=1e-5, stat_a=None, out=None, w1=None):
    """LeakyReLU(norm_a(xa) + norm_b(xb)); params_*: None (both affine-less) or lists of (gamma, beta) pairs, one per style.
    w1: LeakyReLU(norm_a(xa) + norm_b(conv1x1x1(xb; w1))) with a one-channel xb (see _ResNormPair.forward)."""
    global _PENDING_OUT
    flat, n = [], 1
    if params_a is not None:
        n = len(params_a)
        for ps in (params_a, params_b):
            for g, b in ps:
                flat += [g, b]
    _PENDING_OUT = out
    return _ResNormPair.apply(xa, xb, w1, styles_dev, styles_host, n, params_a is not None, slope, eps_a, eps_b, stat_a,
                              _carried_stat(xb) if w1 is None else None, *flat)


def instance_norm(x, params=None, styles_dev=None, styles_host=None, res=None, act=L.ACT_NONE, slope=0.01, eps=1e-5, fork=False, stat=None, out=None):
    """params: None (no affine) | [(gamma, beta)] (plain) | [(g0,b0),(g1,b1),...] (conditional, one pair per style).
    fork=True returns (norm(x), x): see _InstNorm.forward."""
    flat = []
    n = 1
    if params is not None:
        n = len(params)
        for g, b in params:
            flat += [g, b]
    global _PENDING_OUT
    _PENDING_OUT = out
    if stat is None:
        stat = _carried_stat(x)
    global _LAST_STAT
    _LAST_STAT = None
    r = _InstNorm.apply(x, res, styles_dev, styles_host, n, params is not None, act, slope, eps, fork, stat, *flat)
    if fork and res is None:
        # the skip branch is x itself: its statistics - left on it by its producer, or summed by this norm's own statistics pass (round 5: the
        # affine-less norm of a Swin stage's returned feature map and norm1 of the stage's first block read the SAME tensor; the second
        # statistics launch, 4.7 us of the un-overlapped forward chain per stage, is gone) - stay valid for the next norm of x
        st = stat if isinstance(stat, torch.Tensor) else (_LAST_STAT if FORK_KEEPS_STAT else None)
        if st is not None:
            r[1]._miseg_stat = st
    _LAST_STAT = None
    return r


class _GroupStatNorm(Function):
    """GroupNorm / BatchNorm on channels-last rows out of the instance-norm kernels (reference networks/layers/factories.py:219-257: the
    `batch` / `group` entries of the norm factory, reachable through --encoder_norm_name / --decoder_norm_name, networks/norms/utils.py:11-14).
    Statistics are means over (rows of a sample) x (the `cg` channels of a group) - GroupNorm - or over (all rows of the batch) per channel -
    BatchNorm: the caller passes the batch as ONE sample and cg = 1.  Forward: per-channel sums (miseg_instnorm_stats), the group means as a
    handful of [B, C] float64 operations, miseg_instnorm_apply with the per-channel affine.  Backward: per-channel (sum dy, sum dy xhat)
    (miseg_instnorm_bwd_reduce) give d gamma, d beta directly and, weighted by gamma and summed over the group, the two means of
    dx = rs (gamma dy - mean(gamma dy) - xhat mean(gamma dy xhat)), which one pass evaluates as P dy + R x + Q (miseg_affine2)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cg, eps):
        B, Cc = x.shape[0], x.shape[-1]
        S = ops.rows(x)[1] // B
        G = Cc // cg
        raw = ops.instnorm_stats(x, B, S)                                   # [R, B, C, 2] float64
        tot = raw.sum(0).view(B, G, cg, 2).sum(2, keepdim=True) / cg        # group means of the per-channel sums
        stat = torch.zeros_like(raw)
        stat[0] = tot.expand(B, G, cg, 2).reshape(B, Cc, 2)
        y = ops.instnorm_apply(x, B, S, stat, None, [weight] if weight is not None else None, [bias] if bias is not None else None, eps=eps)
        ctx.save_for_backward(x, stat, weight)
        ctx.meta = (B, S, Cc, cg, eps)
        ctx.params = (weight, bias)
        ctx.mark_non_differentiable(stat)
        return y, stat

    @staticmethod
    def backward(ctx, dy, _dstat=None):
        x, stat, weight = ctx.saved_tensors
        B, S, Cc, cg, eps = ctx.meta
        G = Cc // cg
        dy = _rv(dy)
        ds = ops.instnorm_bwd_reduce(dy, x, B, S, stat, eps).sum(0)        # [B, C, 2]: (sum dy, sum dy xhat) per (sample, channel), float64
        gam = weight.double() if weight is not None else torch.ones(Cc, dtype=torch.float64, device=x.device)
        mu = stat[0, :, :, 0] / S
        rs = 1.0 / torch.sqrt((stat[0, :, :, 1] / S - mu * mu).clamp_min(0.0).float() + eps).double()
        n = float(cg * S)
        A = ((gam * ds[..., 0]).view(B, G, cg).sum(2, keepdim=True) / n).expand(B, G, cg).reshape(B, Cc)
        Bq = ((gam * ds[..., 1]).view(B, G, cg).sum(2, keepdim=True) / n).expand(B, G, cg).reshape(B, Cc)
        coef = torch.stack([rs * gam, -rs * rs * Bq, -rs * A + rs * rs * mu * Bq], dim=-1).float().contiguous()
        dx = ops.affine2(dy, x, coef, B, S) if ctx.needs_input_grad[0] else None
        dg = db = None
        pw, pb = ctx.params
        if pw is not None and ctx.needs_input_grad[1]:
            v = ds[..., 1].sum(0).float()
            slot = _slot(pw)
            if slot is not None:
                slot.add_(v)
            else:
                dg = v
        if pb is not None and ctx.needs_input_grad[2]:
            v = ds[..., 0].sum(0).float()
            slot = _slot(pb)
            if slot is not None:
                slot.add_(v)
            else:
                db = v
        return dx, dg, db, None, None


def group_norm(x, weight, bias, num_groups, eps=1e-5):
    """nn.GroupNorm(num_groups, C) on a channels-last tensor [B, ..., C]"""
    Cc = x.shape[-1]
    if Cc % num_groups != 0:
        raise ValueError("num_channels must be divisible by num_groups")
    return _GroupStatNorm.apply(x, weight, bias, Cc // num_groups, eps)[0]


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5, num_batches_tracked=None):
    """nn.BatchNorm3d on a channels-last tensor [B, ..., C]: batch statistics in training mode (the running estimates are updated like torch:
    momentum mix, unbiased variance; momentum None = cumulative average), the running estimates in eval mode (forward only)."""
    B, Cc = x.shape[0], x.shape[-1]
    if training or running_mean is None:
        xb = x.reshape((1, -1, Cc))                                           # the whole batch is one "sample": per-channel batch statistics
        y, stat = _GroupStatNorm.apply(xb, weight, bias, 1, eps)
        if training and running_mean is not None:
            with torch.no_grad():
                n = xb.shape[1]
                mean = stat[0, 0, :, 0] / n
                var = (stat[0, 0, :, 1] / n - mean * mean).clamp_min(0.0)
                if num_batches_tracked is not None:
                    num_batches_tracked += 1
                m = momentum if momentum is not None else 1.0 / float(num_batches_tracked)
                running_mean.mul_(1.0 - m).add_(mean.to(running_mean.dtype), alpha=m)
                running_var.mul_(1.0 - m).add_((var * (n / max(n - 1, 1))).to(running_var.dtype), alpha=m)
        return y.view(x.shape)
    if torch.is_grad_enabled() and x.requires_grad:
        raise NotImplementedError("BatchNorm in eval mode (running statistics) is forward-only on the MI355X path")
    n = ops.rows(x)[1]
    stat = torch.zeros(L.load().miseg_instnorm_stat_bytes(1, Cc) // 8, dtype=torch.float64, device=x.device).view(-1, 1, Cc, 2)
    rm, rv = running_mean.double(), running_var.double()
    stat[0, 0, :, 0] = rm * n
    stat[0, 0, :, 1] = (rv + rm * rm) * n
    xb = x.reshape((1, -1, Cc))
    return ops.instnorm_apply(xb, 1, n, stat, None, [weight] if weight is not None else None, [bias] if bias is not None else None, eps=eps).view(x.shape)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.has_beta = beta is not None
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = _rv(dy)
        C = x.shape[-1]
        sg, sb = _slot(ctx.params[0]), _slot(ctx.params[1])
        dg = sg if sg is not None else (ops.zeros_f32((C,), x.device) if gamma is not None else None)
        db = sb if sb is not None else (ops.zeros_f32((C,), x.device) if ctx.has_beta else None)
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db)
        return dx, None if sg is not None else dg, None if sb is not None else db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, eps)


# ----------------------------------------------------------------------------------------------------------------
class _Linear(Function):
    """y = x W^T + b (+ res) on the matrix cores (NT GEMM, the residual add rides in its epilogue); dX via the transposed
    pack, dW via the TN GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, want_stat=False):
        w = ops.cast_matrix(weight, x.dtype)
        y = ops.gemm_nt(x, w, bias, res=res, want_stat=want_stat)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _rv(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = ops.cast_matrix(weight, dy.dtype, transpose=True)      # [K, N]
            dx = ops.gemm_nt(dy, wt)
        dw, db = _wb_grads_into(ctx.params[0], ctx.params[1] if ctx.has_bias else None, dy, x, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw, db, (dy if ctx.needs_input_grad[3] else None), None


def _norm_grad_slots(params, affine, num_styles, styles_host, Cc, device):
    """where the affine gradients of a (conditional) instance norm go: (dgamma rows, dbeta rows, in_arena, builder of the returned gradients).
    Rows of a style absent from the batch stay None (reference: grad is None)."""
    present = sorted(set(styles_host)) if styles_host is not None else [0]
    if not affine:
        return None, None, False, lambda: []
    in_arena = getattr(params[0], "_miseg_grad", None) is not None
    if in_arena:
        dgam = [_slot(params[2 * s]) if s in present else None for s in range(num_styles)]
        dbet = [_slot(params[2 * s + 1]) if s in present else None for s in range(num_styles)]
        return dgam, dbet, True, lambda: [None, None] * num_styles
    buf = ops.zeros_f32((num_styles, 2, Cc), device)
    dgam = [buf[s, 0] if s in present else None for s in range(num_styles)]
    dbet = [buf[s, 1] if s in present else None for s in range(num_styles)]

    def build():
        pg = []
        for s in range(num_styles):
            pg += [dgam[s], dbet[s]]
        return pg
    return dgam, dbet, False, build


def _wgrad_into(p, g, act):
    """dW = g^T act of a linear layer: into the parameter's arena slot (returns None) or as a fresh tensor"""
    slot, mode = _slot_first(p)
    if slot is not None:
        with ops.wgrad_side(g, act):
            ops.gemm_tn(g, act, out=slot, accumulate=mode)      # (2: the slot still holds the step's zeros - a store, not a read-modify-write)
        return None
    return ops.gemm_tn(g, act).view(p.shape)


def _wb_grads_into(pw, pb, g, act, need_w=True, need_b=True):
    """(dW, db) of a linear layer from its output gradient g and input act: dW = g^T act, db = column sums of g.  With both parameters in a
    training arena the bias gradient rides in the weight-gradient product's launch where that takes the streaming path (round 5:
    ops.gemm_tn(colsum_out=)); returns None for what went into a slot."""
    need_b = need_b and pb is not None
    if need_w and need_b and getattr(pw, "_miseg_grad", None) is not None and getattr(pb, "_miseg_grad", None) is not None:
        slot, mode = _slot_first(pw)
        with ops.wgrad_side(g, act):
            ops.gemm_tn(g, act, out=slot, accumulate=mode, colsum_out=_slot(pb))
        return None, None
    return (_wgrad_into(pw, g, act) if need_w else None), (_bgrad_into(pb, g) if need_b else None)


def _bgrad_into(p, g):
    slot = _slot(p)
    if slot is not None:
        ops.colsum(g, out=slot, accumulate=True)
        return None
    return ops.colsum(g)


def _norm_bwd_from(dxn, x, S, stat, dstat, styles_dev, gammas, dgam, dbet, eps, gadd):
    """the instance-norm backward behind a data-gradient GEMM: apply only when the GEMM's epilogue left the sums (dstat), else both halves"""
    if dstat is not None:
        return ops.instnorm_bwd_apply(dxn, x, 1, S, stat, dstat, styles_dev, gammas, dgam, dbet, eps=eps, gadd=gadd)
    return ops.instnorm_bwd(dxn, None, x, 1, S, stat, styles_dev, gammas, dgam, dbet, eps=eps, gadd=gadd)[0]


class _NormLinear(Function):
    """y = norm(x) W^T + b with the (conditional) instance norm's apply pass folded into the GEMM's operand load (ONE sample, bf16, the
    tall-skinny GEMM path): the Swin block's norm1 -> qkv (swin_transformer_block.py:103, window_attention.py:101).  fork: second output = x, the
    skip branch of `x + f(norm(x))`; its gradient is added inside the norm-backward apply pass.  Backward: the data-gradient GEMM leaves the
    norm's backward sums in its epilogue, so the norm backward is ONE launch (apply) instead of two."""

    @staticmethod
    def forward(ctx, x, styles_dev, styles_host, num_styles, affine, eps, fork, stat_in, weight, bias, *params):
        S = ops.rows(x)[1]
        gammas = list(params[0::2]) if affine else None
        betas = list(params[1::2]) if affine else None
        stat = stat_in if stat_in is not None else ops.instnorm_stats(x, 1, S)
        keep = any(ctx.needs_input_grad)
        ref = ops.NormRef(stat, styles_dev, gammas, betas, eps)
        r = ops.gemm_nt(x, ops.cast_matrix(weight, x.dtype), bias, anorm=ref, anorm_out=keep)
        y, xn = r if keep else (r, None)
        ctx.save_for_backward(x, xn, stat, styles_dev, weight, *(gammas or []))
        ctx.meta = (S, styles_host, num_styles, affine, eps)
        ctx.params = (weight, bias) + tuple(params)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, gskip=None):
        x, xn, stat, styles_dev, weight, *gammas = ctx.saved_tensors
        S, styles_host, num_styles, affine, eps = ctx.meta
        pw, pb = ctx.params[0], ctx.params[1]
        dy = _rv(dy)
        dgam, dbet, _, build = _norm_grad_slots(ctx.params[2:], affine, num_styles, styles_host, x.shape[-1], x.device)
        dx = None
        if ctx.needs_input_grad[0]:
            wt = ops.cast_matrix(weight, dy.dtype, transpose=True)      # [K, N]
            if ops.gemm_nt_folds(dy, wt, bstat_x=x):
                dxn = ops.gemm_nt(dy, wt, bstat=(x, stat, eps))
                dstat = ops.pop_gemm_stat(dxn)
            else:
                dxn, dstat = ops.gemm_nt(dy, wt), None
            dx = _norm_bwd_from(dxn, x, S, stat, dstat, styles_dev, gammas if affine else None, dgam, dbet, eps, _rv(gskip))
        dw, db = _wb_grads_into(pw, pb, dy, xn, ctx.needs_input_grad[8], ctx.needs_input_grad[9])
        return (dx, None, None, None, None, None, None, None, dw, db, *build())


def norm_linear(x, params, styles_dev, styles_host, eps, weight, bias=None, fork=False):
    """linear(instance_norm(x)) [, x] with the norm folded into the GEMM (see _NormLinear), or None where that form does not apply (more
    than one sample, fp32, a shape outside the tall-skinny path): the caller then composes instance_norm + linear.
    params: None | [(gamma, beta)] | one pair per style."""
    if not (_one_sample(x) and x.dtype == torch.bfloat16):
        return None
    n = len(params) if params is not None else 1
    st = _carried_stat(x)
    ref = ops.NormRef(st if st is not None else x, styles_dev, [g for g, _ in params] if params is not None else None,
                      [b for _, b in params] if params is not None else None, eps)
    if not ops.gemm_nt_folds(x, weight, anorm=ref):
        return None
    flat = []
    if params is not None:
        for g, b in params:
            flat += [g, b]
    return _NormLinear.apply(x, styles_dev, styles_host, n, params is not None, eps, fork, st, weight, bias, *flat)


class _NormMlp(Function):
    """y = x + W2 gelu(W1 norm(x) + b1) + b2: the second half of a Swin block (swin_transformer_block.py:176-205,251) with the (conditional)
    instance norm's apply pass folded into the token load of the fused MLP kernel / of the fc1 GEMM, the residual in the fc2 epilogue, and -
    backward - the norm's sums in the epilogue of the kernel that produces the gradient of its output (ONE sample, bf16)."""

    @staticmethod
    def forward(ctx, x, styles_dev, styles_host, num_styles, affine, eps, want_stat, stat_in, w1, b1, w2, b2, *params):
        S = ops.rows(x)[1]
        gammas = list(params[0::2]) if affine else None
        betas = list(params[1::2]) if affine else None
        stat = stat_in if stat_in is not None else ops.instnorm_stats(x, 1, S)
        keep = any(ctx.needs_input_grad)
        ref = ops.NormRef(stat, styles_dev, gammas, betas, eps)
        ctx.fused = ops.mlp_fused(x, w1.shape[0])
        w1c, w2c = ops.cast_matrix(w1, x.dtype), ops.cast_matrix(w2, x.dtype)
        h = a = None
        if ctx.fused:
            r = ops.mlp_fwd(x, w1c, b1, w2c, b2, res=x, want_stat=want_stat, anorm=ref, anorm_out=keep)
            y, xn = r if keep else (r, None)
        else:
            h = torch.empty(x.shape[:-1] + (w1.shape[0],), dtype=x.dtype, device=x.device) if keep else None
            r = ops.gemm_nt(x, w1c, b1, act=L.ACT_GELU, preact_out=h, anorm=ref, anorm_out=keep)
            a, xn = r if keep else (r, None)
            y = ops.gemm_nt(a, w2c, b2, res=x, want_stat=want_stat)
        ctx.save_for_backward(x, xn, h, a, stat, styles_dev, w1, w2, *(gammas or []))
        ctx.meta = (S, styles_host, num_styles, affine, eps)
        ctx.params = (w1, b1, w2, b2) + tuple(params)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xn, h, a, stat, styles_dev, w1, w2, *gammas = ctx.saved_tensors
        S, styles_host, num_styles, affine, eps = ctx.meta
        pw1, pb1, pw2, pb2 = ctx.params[:4]
        dy = _rv(dy)
        dgam, dbet, _, build = _norm_grad_slots(ctx.params[4:], affine, num_styles, styles_host, x.shape[-1], x.device)
        w1t, w2t = ops.cast_matrix(w1, dy.dtype, transpose=True), ops.cast_matrix(w2, dy.dtype, transpose=True)
        if ctx.fused:      # z recomputed from norm(x); dz and h are written once, for the two weight-gradient products below
            dh, a, dxn, dstat = ops.mlp_bwd(xn, dy, ops.cast_matrix(w1, dy.dtype), pb1, w2t, w1t, need_dx=True, bstat=(x, stat, eps))
        else:
            dh = ops.gemm_nt(dy, w2t, gelu_grad_of=h)
            if ops.gemm_nt_folds(dh, w1t, bstat_x=x):
                dxn = ops.gemm_nt(dh, w1t, bstat=(x, stat, eps))
                dstat = ops.pop_gemm_stat(dxn)
            else:
                dxn, dstat = ops.gemm_nt(dh, w1t), None
        # the skip branch's gradient is dy itself: added inside the norm-backward apply pass
        dx = _norm_bwd_from(dxn, x, S, stat, dstat, styles_dev, gammas if affine else None, dgam, dbet, eps, dy)
        out = [dx, None, None, None, None, None, None, None, None, None, None, None]
        out[8], out[9] = _wb_grads_into(pw1, pb1, dh, xn, ctx.needs_input_grad[8], ctx.needs_input_grad[9])
        out[10], out[11] = _wb_grads_into(pw2, pb2, dy, a, ctx.needs_input_grad[10], ctx.needs_input_grad[11])
        return (*out, *build())


def norm_mlp(x, params, styles_dev, styles_host, eps, w1, b1, w2, b2, want_stat=False):
    """x + mlp(instance_norm(x)) with the norm folded into the MLP's first product (see _NormMlp), or None where that form does not apply."""
    if not (_one_sample(x) and x.dtype == torch.bfloat16):
        return None
    n = len(params) if params is not None else 1
    st = _carried_stat(x)
    ref = ops.NormRef(st if st is not None else x, styles_dev, [g for g, _ in params] if params is not None else None,
                      [b for _, b in params] if params is not None else None, eps)
    if not ops.FOLD_NORMS:
        return None
    if not ops.mlp_fused(x, w1.shape[0]) and not ops.gemm_nt_folds(x, w1, anorm=ref, act=L.ACT_GELU):
        return None
    flat = []
    if params is not None:
        for g, b in params:
            flat += [g, b]
    y = _NormMlp.apply(x, styles_dev, styles_host, n, params is not None, eps, want_stat, st, w1, b1, w2, b2, *flat)
    return _tag_stat(y) if want_stat else y


def _one_sample(x):
    return x.dim() >= 3 and x.shape[0] == 1


def _tag_stat(y):
    st = ops.pop_gemm_stat(y)
    if st is not None:
        y._miseg_stat = st
    return y


def linear(x, weight, bias=None, res=None, want_stat=False):
    """want_stat: the output feeds an instance norm - where the GEMM can (one sample, tall-skinny bf16 path) its epilogue leaves the
    statistics on the result (`_miseg_stat`), which instance_norm() picks up instead of running its statistics pass."""
    want = want_stat and _one_sample(x)
    y = _Linear.apply(x, weight, bias, res, want)
    return _tag_stat(y) if want else y


class _Mlp(Function):
    """y = W2 gelu(W1 x + b1) + b2 (+ res)   (MONAI MLPBlock, swin_transformer_block.py:97 / transformer_block.py:58).
    Forward: the first GEMM writes the pre-activation h and gelu(h) from one epilogue, the second adds the residual in its
    epilogue.  Backward: the GELU derivative is the epilogue of the data-gradient GEMM behind it (dh = (dy W2) * gelu'(h))."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res, want_stat=False):
        ctx.fused = ops.mlp_fused(x, w1.shape[0])
        if ctx.fused:      # 48 -> 192 -> 48 on >= 4096 tokens (stage 1 of the headline net): one launch, hidden activations never stored
            y = ops.mlp_fwd(x, ops.cast_matrix(w1, x.dtype), b1, ops.cast_matrix(w2, x.dtype), b2, res=res, want_stat=want_stat)
            ctx.save_for_backward(x, None, None, w1, w2)
            ctx.params = (w1, b1, w2, b2)
            return y
        # the pre-activation is only the backward pass's input (gelu'): an inference forward (no_grad) does not write it
        h = torch.empty(x.shape[:-1] + (w1.shape[0],), dtype=x.dtype, device=x.device) if any(ctx.needs_input_grad) else None
        a = ops.gemm_nt(x, ops.cast_matrix(w1, x.dtype), b1, act=L.ACT_GELU, preact_out=h)
        y = ops.gemm_nt(a, ops.cast_matrix(w2, x.dtype), b2, res=res, want_stat=want_stat)
        ctx.save_for_backward(x, h, a, w1, w2)
        ctx.params = (w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, a, w1, w2 = ctx.saved_tensors
        pw1, pb1, pw2, pb2 = ctx.params
        dy = _rv(dy)
        if ctx.fused:      # z recomputed from x; dz and h are written once, for the two weight-gradient products below
            dh, a, dx = ops.mlp_bwd(x, dy, ops.cast_matrix(w1, dy.dtype), pb1, ops.cast_matrix(w2, dy.dtype, transpose=True),
                                    ops.cast_matrix(w1, dy.dtype, transpose=True), need_dx=ctx.needs_input_grad[0])
        else:
            dh = ops.gemm_nt(dy, ops.cast_matrix(w2, dy.dtype, transpose=True), gelu_grad_of=h)
            dx = ops.gemm_nt(dh, ops.cast_matrix(w1, dy.dtype, transpose=True)) if ctx.needs_input_grad[0] else None
        out = [dx, None, None, None, None, dy if ctx.needs_input_grad[5] else None, None]
        for i, (pw, pb, act, g) in enumerate(((pw1, pb1, x, dh), (pw2, pb2, a, dy))):
            out[1 + 2 * i], out[2 + 2 * i] = _wb_grads_into(pw, pb, g, act, ctx.needs_input_grad[1 + 2 * i], ctx.needs_input_grad[2 + 2 * i])
        return tuple(out)


def mlp(x, w1, b1, w2, b2, res=None, want_stat=False):
    want = want_stat and _one_sample(x)
    y = _Mlp.apply(x, w1, b1, w2, b2, res, want)
    return _tag_stat(y) if want else y


class _Gelu(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(_rv(dy), x)


def gelu(x):
    return _Gelu.apply(x)


# ----------------------------------------------------------------------------------------------------------------
class _Conv3(Function):
    """3x3x3 / s1 / p1 / no bias (dynunet_block.py:295-326) as implicit GEMM; dgrad = same kernel, mirrored pack.
    want_stat: second output = the instance-norm statistics of y from the kernel's epilogue (None where the kernel path has none).
    fork: last output = x again (the residual branch of a UnetResBlock); the gradient arriving there is added in the epilogue of the
    data-gradient kernel instead of by a separate pass."""

    @staticmethod
    def forward(ctx, x, weight, want_stat=False, fork=False, dx_to_norm=False, s2c_left=0):
        """dx_to_norm: x is the output of a (conditional) instance norm that nothing else reads, so the data gradient computed here goes
        straight to that norm's backward pass - a split launch over a small stage may then leave its partial slabs to it (ops.pending_dx_put).
        s2c_left = C > 0: x is the concat buffer of upconv_cat whose first C channels came from a ConvTranspose3d(k2, s2); the data gradient
        of those channels leaves the kernel in the layout that layer's backward reads (ops.conv3_fwd(s2c=); handed over by ops.pending_dx_put)"""
        need_dx = ctx.needs_input_grad[0]
        ctx.dx_to_norm = bool(dx_to_norm)
        ctx.s2c_left = int(s2c_left)
        fwdp, bwdp = ops.pack_conv3(weight, x.dtype, True, need_dx)
        # want_stat = "defer": the statistics slot may come back as ops.PendingSlabs - y is then written by the instance norm that consumes it
        y, stat = ops.conv3_fwd(x, fwdp, weight.shape[0], want_stat=want_stat) if want_stat else (ops.conv3_fwd(x, fwdp, weight.shape[0]), None)
        ctx.save_for_backward(x, bwdp)
        ctx.wshape = weight.shape
        ctx.params = (weight,)
        ctx.layout = (bool(want_stat), fork)
        # the statistics output never gets a gradient: without this autograd would materialise a zero tensor for it (a fill launch)
        ctx.set_materialize_grads(False)
        outs = [y]
        if want_stat:
            if isinstance(stat, torch.Tensor):
                ctx.mark_non_differentiable(stat)
            outs.append(stat)
        if fork:
            outs.append(x.view_as(x))
        return tuple(outs) if len(outs) > 1 else y

    @staticmethod
    def backward(ctx, dy, *rest):
        x, bwdp = ctx.saved_tensors
        dy = _rv(dy)
        gskip = _rv(rest[-1]) if ctx.layout[1] and rest[-1] is not None else None
        dx = None
        # the gradient of the forked input may be a promise (_Conv1.backward, fold_dx): (g3, W3^T) of the block's 1x1x1 shortcut convolution,
        # whose data gradient g3 W3 this launch then forms itself (ops.conv3_fwd(sc=)) - the tensor behind `gskip` was never written
        sc = ops.pending_dx_take(gskip) if gskip is not None else None
        if sc is not None:
            assert isinstance(sc, tuple) and sc[0] == "sc", "the gradient of a forked conv input arrived as partial slabs"
            sc, gskip = sc[1:], None
        if sc is not None and not (ctx.needs_input_grad[0] and ops.conv3_fuses_shortcut(dy, ctx.wshape[1], sc[0].shape[-1])):
            gskip, sc = ops.gemm_nt(sc[0], sc[1]), None          # (the promise cannot be kept by this launch after all: the plain product)
        if ctx.needs_input_grad[0]:
            if ctx.dx_to_norm and gskip is None and sc is None:
                dx, pend = ops.conv3_fwd(dy, bwdp, ctx.wshape[1], defer=True)
                if pend is not None:
                    ops.pending_dx_put(dx, pend)
            else:
                dy8 = None
                if ctx.s2c_left and ops.conv3_fuses_s2c(dy, ctx.wshape[1], ctx.s2c_left):
                    B_, D_, H_, W_ = x.shape[:4]
                    dy8 = torch.empty(B_, D_ // 2, H_ // 2, W_ // 2, 8 * ctx.s2c_left, dtype=dy.dtype, device=dy.device)
                dx = ops.conv3_fwd(dy, bwdp, ctx.wshape[1], res=gskip, sc=sc, s2c=dy8)
                if dy8 is not None:
                    ops.pending_dx_put(dx, ("dy8", dy8))      # the first s2c_left channels of dx were NOT written: _UpCat.backward takes dy8
        dw = None
        if ctx.needs_input_grad[1]:
            slot, mode = _slot_first(ctx.params[0])
            if slot is not None and ops.defer_to_branch(x, dy, slot, mode):
                pass
            elif slot is not None:
                q = ops._queues(slot)
                if q is not None and q.branch_deferred and ops.in_branch_backward():
                    ops.stamp("branch_bwd_head", fine=True)
                    ops.flush_branch_deferred(q)
                with ops.wgrad_side(x, dy, kind="conv"):
                    ops.conv3_wgrad(x, dy, dw=slot, accumulate=mode)
            else:
                dw = ops.conv3_wgrad(x, dy)
        return dx, dw, None, None, None, None


class _Conv3Shortcut(Function):
    """(y, stat, y3, stat3) = (conv3x3x3(x, w1), conv1x1x1(x, w3)) - the first convolution of a residual block and the block's shortcut convolution
    (dynunet_block.py:87-97,100-126: both read the block's input) as ONE launch in both directions where the kernel can (round 5):
    forward, the shortcut's output is a second pass of the 3x3x3 launch over the centre tap (ops.conv3_fwd(fs=)); backward, its data gradient is
    one more centre tap of the 3x3x3 data-gradient launch (sc=) and, for a concat input, the transposed convolution's half of the result
    leaves in space-to-channel order (s2c=).  Elsewhere the separate GEMMs run."""

    @staticmethod
    def forward(ctx, x, w1, w3, want_stat, s2c_left):
        need_dx = ctx.needs_input_grad[0]
        fwdp, bwdp = ops.pack_conv3(w1, x.dtype, True, need_dx)
        Cout = w1.shape[0]
        w3c = ops.cast_matrix(w3, x.dtype).view(w3.shape[0], -1)
        if ops.conv3_fuses_fwd_shortcut(x, Cout) and w3c.is_contiguous():
            y, stat, y3, stat3 = ops.conv3_fwd(x, fwdp, Cout, want_stat=want_stat, fs=(w3c, True))
        else:
            y, stat = ops.conv3_fwd(x, fwdp, Cout, want_stat=want_stat) if want_stat else (ops.conv3_fwd(x, fwdp, Cout), None)
            y3 = ops.gemm_nt(x, w3c, want_stat=True)
            stat3 = ops.pop_gemm_stat(y3)
        ctx.save_for_backward(x, bwdp, w3)
        ctx.wshape = w1.shape
        ctx.params = (w1, w3)
        ctx.s2c_left = int(s2c_left)
        ctx.set_materialize_grads(False)
        for st in (stat, stat3):
            if isinstance(st, torch.Tensor):
                ctx.mark_non_differentiable(st)
        return y, stat, y3, stat3

    @staticmethod
    def backward(ctx, dy, _dstat, dy3, _dstat3):
        x, bwdp, w3 = ctx.saved_tensors
        dy, dy3 = _rv(dy), _rv(dy3)
        Cin = ctx.wshape[1]
        dx = None
        if ctx.needs_input_grad[0]:
            w3t = ops.cast_matrix(w3, dy.dtype, transpose=True)      # [Cin][Cout3]
            fold = (dy3.dtype == torch.bfloat16 and dy3.data_ptr() % 16 == 0 and ops.rows(dy3)[0] % 8 == 0 and w3t.is_contiguous()
                    and ops.conv3_fuses_shortcut(dy, Cin, dy3.shape[-1]))
            gskip = None if fold else ops.gemm_nt(dy3, w3t)
            dy8 = None
            if ctx.s2c_left and ops.conv3_fuses_s2c(dy, Cin, ctx.s2c_left):
                B_, D_, H_, W_ = x.shape[:4]
                dy8 = torch.empty(B_, D_ // 2, H_ // 2, W_ // 2, 8 * ctx.s2c_left, dtype=dy.dtype, device=dy.device)
            dx = ops.conv3_fwd(dy, bwdp, Cin, res=gskip, sc=(dy3, w3t) if fold else None, s2c=dy8)
            if dy8 is not None:
                ops.pending_dx_put(dx, ("dy8", dy8))
        dw1 = dw3 = None
        if ctx.needs_input_grad[1]:
            slot, mode = _slot_first(ctx.params[0])
            if slot is not None and ops.defer_to_branch(x, dy, slot, mode):
                pass
            elif slot is not None:
                with ops.wgrad_side(x, dy, kind="conv"):
                    ops.conv3_wgrad(x, dy, dw=slot, accumulate=mode)
            else:
                dw1 = ops.conv3_wgrad(x, dy)
        if ctx.needs_input_grad[2]:
            slot, mode = _slot_first(ctx.params[1])
            if slot is not None:
                with ops.wgrad_side(dy3, x):
                    ops.gemm_tn(dy3, x, out=slot, accumulate=mode)
            else:
                dw3 = ops.gemm_tn(dy3, x).view(w3.shape)
        return dx, dw1, dw3, None, None


def conv3_shortcut(x, w1, w3, want_stat=False):
    """(out, stat, shortcut output) of a residual block with a shortcut convolution; None where this form does not apply (the caller composes
    conv3(fork=True) + conv1)"""
    if not (x.dtype == torch.bfloat16 and x.dim() == 5 and x.requires_grad and _one_sample(x) and ops.conv3_fuses_fwd_shortcut(x, w1.shape[0])):
        return None
    y, st, y3, st3 = _Conv3Shortcut.apply(x, w1, w3, want_stat, getattr(x, "_miseg_upcat", 0))
    if st3 is not None:
        y3._miseg_stat = st3      # (what conv1(want_stat=True) leaves on its result: the pair norm behind it skips its statistics pass)
    return y, st, y3


def conv3(x, weight, want_stat=False, fork=False, dx_to_norm=False):
    r = _Conv3.apply(x, weight, want_stat, fork, dx_to_norm, getattr(x, "_miseg_upcat", 0) if x.requires_grad else 0)
    if fork and x.dtype == torch.bfloat16 and x.requires_grad and x.dim() == 5:
        # may a 1x1x1 convolution on the forked input leave its data gradient to this convolution's data-gradient launch? (conv1(fold_dx))
        B, D, H, W, Cin = x.shape
        Cout = weight.shape[0]
        if ops.FOLD_SHORTCUT and ops.L.load().miseg_conv3_fuses_shortcut(B, D, H, W, Cout, Cin, Cout, ops.L.BF16):
            r[-1]._miseg_sc_fold = Cout
    return r


class _Conv3T(Function):
    """stride-1 3x3x3 convolution with the weight of a ConvTranspose3d ([Cin][Cout][3][3][3]): the forward is the data-gradient
    kernel of the ordinary convolution whose weight tensor this is (mirrored taps, swapped channels), and vice versa."""

    @staticmethod
    def forward(ctx, x, weight):
        fwdp, bwdp = ops.pack_conv3(weight, x.dtype, ctx.needs_input_grad[0], True)
        ctx.save_for_backward(x, fwdp)
        ctx.wshape = weight.shape
        ctx.params = (weight,)
        return ops.conv3_fwd(x, bwdp, weight.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x, fwdp = ctx.saved_tensors
        dy = _rv(dy)
        dx = ops.conv3_fwd(dy, fwdp, ctx.wshape[0]) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            slot = _slot(ctx.params[0])
            if slot is not None:
                with ops.wgrad_side(x, dy, kind="conv"):
                    ops.conv3_wgrad(dy, x, dw=slot, accumulate=True)
            else:
                dw = ops.conv3_wgrad(dy, x)          # roles swapped: dw[Cin_t][Cout_t][27]
        return dx, dw


def conv3_transposed_weight(x, weight):
    return _Conv3T.apply(x, weight)


class _Resample2(Function):
    @staticmethod
    def forward(ctx, x, up, fine_shape):
        ctx.up, ctx.fine = up, tuple(x.shape[1:4]) if not up else tuple(fine_shape)
        return ops.resample2(x, up, fine_shape)

    @staticmethod
    def backward(ctx, g):
        return ops.resample2(_rv(g), not ctx.up, ctx.fine if not ctx.up else None), None, None


def subsample2(x):
    """x[:, ::2, ::2, ::2, :] -- with conv3 in front: the stride-2 convolution of convolutions.py:131-139."""
    return _Resample2.apply(x, False, None)


def upsample2_zero(x, fine_shape):
    return _Resample2.apply(x, True, tuple(fine_shape))


class _RowBias(Function):
    @staticmethod
    def forward(ctx, x, bias):
        ctx.params = (bias,)
        return ops.rowbias_add(x, bias)

    @staticmethod
    def backward(ctx, g):
        g = _rv(g)
        db = None
        if ctx.needs_input_grad[1]:
            slot = _slot(ctx.params[0])
            if slot is not None:
                ops.colsum(g, out=slot, accumulate=True)
            else:
                db = ops.colsum(g)
        return g, db


def rowbias(x, bias):
    return x if bias is None else _RowBias.apply(x, bias)


class _AddPosition(Function):
    """x [B, L, C] + the learned position table [1, L, C] (MONAI PatchEmbeddingBlock, reference patch_embedding.py:121): the fp32 parameter
    is added to the activations directly (one rounding), its gradient - the sum of dy over the batch - goes to the parameter's arena slot
    like every other weight gradient (a `.to(dtype)` copy made autograd write `p.grad` itself, which arena.publish() then dropped)."""

    @staticmethod
    def forward(ctx, x, pos):
        ctx.params = (pos,)
        B = x.shape[0]
        return ops.rowbias_add(x.reshape(B, -1), pos.reshape(-1)).view(x.shape)

    @staticmethod
    def backward(ctx, g):
        g = _rv(g)
        pos = ctx.params[0]
        dpos = None
        if ctx.needs_input_grad[1]:
            g2 = g.reshape(g.shape[0], -1)
            slot = _slot(pos)
            if slot is not None:
                ops.colsum(g2, out=slot.view(-1), accumulate=True)
            else:
                dpos = ops.colsum(g2).view(pos.shape)
        return g, dpos


def add_position(x, pos):
    return _AddPosition.apply(x, pos)


class _PReLU(Function):
    @staticmethod
    def forward(ctx, x, slope):
        ctx.save_for_backward(x, slope)
        ctx.params = (slope,)
        return ops.prelu_fwd(x, slope)

    @staticmethod
    def backward(ctx, g):
        x, slope = ctx.saved_tensors
        slot = _slot(ctx.params[0]) if ctx.needs_input_grad[1] else None
        ds = slot if slot is not None else (ops.zeros_f32(slope.shape, x.device) if ctx.needs_input_grad[1] else None)
        dx = ops.prelu_bwd(_rv(g), x, slope, ds)
        return dx, None if slot is not None else ds


def prelu(x, slope):
    return _PReLU.apply(x, slope)


_CONST_SLOPES = {}


def leaky_relu(x, slope=0.01):
    """LeakyReLU(slope) (slope 0: ReLU) as the PReLU kernels with a constant one-element slope (no gradient for it)"""
    key = (x.device, float(slope))
    t = _CONST_SLOPES.get(key)
    if t is None:
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            # a tensor created under capture lives in the graph's private pool and is only filled when the graph is replayed: cached here, a
            # later eager call could read it unfilled.  First use inside a capture gets a private copy; the cache is filled by an eager call
            return _PReLU.apply(x, torch.full((1,), float(slope), dtype=torch.float32, device=x.device))
        t = _CONST_SLOPES[key] = torch.full((1,), float(slope), dtype=torch.float32, device=x.device)
    return _PReLU.apply(x, t)


class _Cat2(Function):
    """cat([a, b], channel) with our strided-copy kernel; the backward hands out the two halves as views."""

    @staticmethod
    def forward(ctx, a, b):
        ca, cb = a.shape[-1], b.shape[-1]
        out = torch.empty(a.shape[:-1] + (ca + cb,), dtype=a.dtype, device=a.device)
        ops.copy2d(a, out[..., :ca])
        ops.copy2d(b, out[..., ca:])
        ctx.ca = ca
        return out

    @staticmethod
    def backward(ctx, g):
        g = _rv(g)
        return g[..., :ctx.ca], g[..., ctx.ca:]


def cat_channels(a, b):
    return _Cat2.apply(a, b)


def image_rows(x_ncdhw, dtype):
    """the NCDHW fp32 network input as channels-last rows [B, D, H, W, C] in the compute dtype (no gradient: the image is data) - the entry
    of multi-channel images (--in_channels > 1, utils/parser.py:11) into the ordinary convolution kernels"""
    return ops.ncdhw_to_rows_exact(x_ncdhw.detach().float().contiguous(), dtype)


class _ToNCDHW(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.dtype = x.dtype
        return ops.rows_to_ncdhw(x)

    @staticmethod
    def backward(ctx, g):
        return ops.ncdhw_to_rows_exact(g.contiguous().float(), ctx.dtype)


def to_ncdhw(x):
    return _ToNCDHW.apply(x)


class _Conv3Thin(Function):
    """stem conv straight from the NCDHW fp32 network input (Cin <= 4): the image becomes channels-last rows of one
    16-byte vector per voxel and goes through the implicit-GEMM kernels (the pack zero-pads Cin the same way)."""

    @staticmethod
    def forward(ctx, x_ncdhw, weight, dtype):
        ctx.wshape = weight.shape
        ctx.params = (weight,)
        # one image channel (CT / MR) and a multiple of 8 output channels: the VALU brick kernels (27 taps are no matrix-core K)
        ctx.stem = weight.shape[1] == 1 and weight.shape[0] % 8 == 0 and weight.shape[0] <= 56
        if ctx.stem:
            ctx.save_for_backward(x_ncdhw)
            return ops.conv3_thin_fwd(x_ncdhw, weight, dtype)
        xr = ops.ncdhw_to_rows(x_ncdhw, dtype)
        fwdp, _ = ops.pack_conv3(weight, dtype, True, False)
        ctx.save_for_backward(xr)
        return ops.conv3_fwd(xr, fwdp, weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (xr,) = ctx.saved_tensors
        slot = _slot(ctx.params[0])
        if ctx.stem:
            dy = _rv(dy)
            if slot is not None:
                ops.conv3_thin_wgrad(xr, dy, slot)
                ops.early_group_flush(ops._queues(slot))      # the stem's weight gradient is the last node of a side branch's backward pass
                return None, None, None
            return None, ops.conv3_thin_wgrad(xr, dy, ops.zeros_f32(ctx.wshape, dy.device)), None
        if slot is not None:
            dy = _rv(dy)
            with ops.wgrad_side(xr, dy, kind="conv"):
                dwp = ops.conv3_wgrad(xr, dy)                    # [Cout, CP, 3, 3, 3], channels >= Cin are zero
                slot.add_(dwp[:, : ctx.wshape[1]])
                ops._WGRAD_KEEP.append(dwp)
            return None, None, None
        dwp = ops.conv3_wgrad(xr, _rv(dy))
        return None, dwp[:, : ctx.wshape[1]].contiguous(), None


def conv3_thin(x_ncdhw, weight, dtype):
    return _Conv3Thin.apply(x_ncdhw, weight, dtype)


class _Conv1(Function):
    """1x1x1 conv, no bias (ResBlock shortcut dynunet_block.py:87-97) == Linear over the channel dim."""

    @staticmethod
    def forward(ctx, x, weight, want_stat=False, fold_dx=False):
        """fold_dx: x is the forked input of a 3x3x3 convolution (conv3(fork=True)) whose data-gradient launch can take this layer's data
        gradient along: backward then returns an UNWRITTEN tensor and leaves (dy, W^T) for that launch (ops.pending_dx_put)"""
        w = ops.cast_matrix(weight, x.dtype)
        ctx.save_for_backward(x, weight)
        ctx.params = (weight,)
        ctx.fold_dx = bool(fold_dx)
        return ops.gemm_nt(x, w, want_stat=want_stat)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _rv(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            wt = ops.cast_matrix(weight, dy.dtype, transpose=True)      # [Cin][Cout]
            if (ctx.fold_dx and dy.dtype == torch.bfloat16 and dy.data_ptr() % 16 == 0 and ops.rows(dy)[0] % 8 == 0 and wt.is_contiguous()
                    and wt.data_ptr() % 16 == 0):
                dx = torch.empty_like(x)
                ops.pending_dx_put(dx, ("sc", dy, wt))
            else:
                dx = ops.gemm_nt(dy, wt)
        dw = None
        if ctx.needs_input_grad[1]:
            slot, mode = _slot_first(ctx.params[0])
            if slot is not None:
                with ops.wgrad_side(dy, x):
                    ops.gemm_tn(dy, x, out=slot, accumulate=mode)
            else:
                dw = ops.gemm_tn(dy, x).view(weight.shape)
        return dx, dw, None, None


def conv1(x, weight, want_stat=False):
    want = want_stat and _one_sample(x)
    fold = getattr(x, "_miseg_sc_fold", None) == weight.shape[0]      # (the forked input of a conv3 with as many output channels: UnetResBlock)
    y = _Conv1.apply(x, weight, want, fold)
    return _tag_stat(y) if want else y


class _UpCat(Function):
    """ConvTranspose3d(k2, s2, no bias) followed by cat([up, skip], C)  (unetr_block.py:80-85): one GEMM
    [voxels, Cin] x [Cin, 8*Cout], a 2x2x2 scatter straight into the first half of the concat buffer, and a strided
    copy of the skip into the second half.  skip=None gives the bare transposed conv (UnetrPrUpBlock)."""

    @staticmethod
    def forward(ctx, x, skip, weight):
        Cin, Cout = weight.shape[0], weight.shape[1]
        B, d, h, w = x.shape[0], x.shape[1], x.shape[2], x.shape[3]
        dev = x.device
        wf = ops.cast_matrix(weight, x.dtype, transpose=True, regroup=(8, Cout))     # [(j,co)][ci]
        width = 2 * Cout if skip is not None else Cout
        # a skip that its producer already wrote into the right half of a concat buffer (`concat_buffer`) needs no copy
        cat = getattr(skip, "_miseg_cat", None) if skip is not None else None
        placed = (cat is not None and tuple(cat.shape) == (B, 2 * d, 2 * h, 2 * w, width) and cat.dtype == x.dtype
                  and skip.data_ptr() == cat.data_ptr() + Cout * cat.element_size() and skip.stride(-2) == width)
        if not placed:
            cat = torch.empty(B, 2 * d, 2 * h, 2 * w, width, dtype=x.dtype, device=dev)
        # the tall layers store the 2x2x2 scatter straight from the GEMM's epilogue; the others go through [voxels, 8*Cout]
        if not ops.gemm_nt_scatter(x, wf, cat[..., :Cout], (B, d, h, w)):
            ops.channel_to_space(ops.gemm_nt(x, wf), STD_OFFSETS, (B, 2 * d, 2 * h, 2 * w, Cout), out=cat[..., :Cout])
        if skip is not None and not placed:
            ops.copy2d(skip, cat[..., Cout:])
        ctx.save_for_backward(x, weight)
        ctx.has_skip = skip is not None
        ctx.params = (weight,)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        x, weight = ctx.saved_tensors
        Cin, Cout = weight.shape[0], weight.shape[1]
        dcat = _rv(dcat)
        dskip = dcat[..., Cout:] if ctx.has_skip and ctx.needs_input_grad[1] else None
        pend = ops.pending_dx_take(dcat)      # the producer of dcat may have stored these channels in this order itself (_Conv3.backward, s2c_left)
        if pend is not None:
            assert isinstance(pend, tuple) and pend[0] == "dy8" and pend[1].shape[-1] == 8 * Cout
            dy8 = pend[1]
        else:
            dy8 = ops.space_to_channel(dcat[..., :Cout], STD_OFFSETS)                 # [B,d,h,w,8*Cout]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dy8, ops.cast_matrix(weight, x.dtype, regroup=(8, Cout)))   # [ci][(j,co)]
        if ctx.needs_input_grad[2]:
            slot, mode = _slot_first(ctx.params[0])
            if slot is not None and ops.gemm_tn_regroups(x, dy8, slot):
                # dW[ci][co][j] = sum over voxels of x[v][ci] dy8[v][(j, co)]: the grouped launch / the batched partial-tile sum stores the product's
                # column (j, co) at (co, j) - the torch layout of the arena slot, no [(j, co)][ci] intermediate, no permute pass (round 5)
                with ops.wgrad_side(x, dy8):
                    ops.gemm_tn(x, dy8, out=slot.view(Cin, 8 * Cout), accumulate=mode, regroup=Cout)
            elif slot is not None:
                with ops.wgrad_side(dy8, x):
                    dwf = ops.gemm_tn(dy8, x)                                             # [(j,co)][ci]
                    ops.permute3(dwf, slot, (Cin, Cout, 8), (1, Cin, Cout * Cin), accumulate=True)
                    ops._WGRAD_KEEP.append(dwf)
            else:
                dwf = ops.gemm_tn(dy8, x)                                                 # [(j,co)][ci]
                dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
                ops.permute3(dwf, dw, (Cin, Cout, 8), (1, Cin, Cout * Cin))
        return dx, dskip, dw


def upconv_cat(x, skip, weight):
    cat = _UpCat.apply(x, skip, weight)
    if skip is not None:
        cat._miseg_upcat = weight.shape[1]      # (read by conv3(): the data gradient of these channels may leave its kernel in space-to-channel order)
    return cat


def concat_buffer(shape_bdhw, channels, dtype, device):
    """(buffer [B,D,H,W,2C], its right half as a rows view): hand the view to the block that produces a decoder's skip tensor
    (`out=`), tag the result with `tag_concat`, and upconv_cat finds the skip already in place."""
    cat = torch.empty(*shape_bdhw, 2 * channels, dtype=dtype, device=device)
    return cat, cat[..., channels:]


def tag_concat(skip, cat):
    skip._miseg_cat = cat
    return skip


# ----------------------------------------------------------------------------------------------------------------
class _S2C(Function):
    @staticmethod
    def forward(ctx, x, offsets):
        ctx.offsets, ctx.shape = offsets, tuple(x.shape)
        return ops.space_to_channel(x, offsets)

    @staticmethod
    def backward(ctx, g):
        return ops.channel_to_space(_rv(g), ctx.offsets, ctx.shape), None


def space_to_channel(x, offsets):
    return _S2C.apply(x, offsets)


class _WinAttn(Function):
    """fused pad/roll/partition + softmax(q k^T * scale + bias + mask) v + reverse/roll/crop
    (window_attention.py:99-119, swin_transformer_block.py:116-169)."""

    @staticmethod
    def forward(ctx, qkv, qkv_bias, table, heads, window, shift, tw, scale, drop_p):
        # attn_drop (swin_transformer_block.py:56-58,91): the mask is drawn inside the kernels from a key, the backward re-creates it
        ctx.drop = (drop_p, ops.DROP.next_key(qkv.device)) if drop_p > 0.0 else None
        out, lse = ops.winattn_fwd(qkv, qkv_bias, table, heads, window, shift, tw, scale, ctx.drop)
        ctx.save_for_backward(qkv, out, lse, qkv_bias, table)
        ctx.meta = (heads, window, shift, tw, scale)
        ctx.params = (qkv_bias, table)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, qkv_bias, table = ctx.saved_tensors
        heads, window, shift, tw, scale = ctx.meta
        sq = _slot(ctx.params[0]) if ctx.needs_input_grad[1] else None
        st = _slot(ctx.params[1]) if ctx.needs_input_grad[2] else None
        dqb = sq if sq is not None else (ops.zeros_f32(qkv_bias.shape, qkv.device) if qkv_bias is not None and ctx.needs_input_grad[1] else None)
        dtab = st if st is not None else (ops.zeros_f32(table.shape, qkv.device) if table is not None and ctx.needs_input_grad[2] else None)
        dqkv = ops.winattn_bwd(qkv, out, lse, _rv(dout), qkv_bias, table, heads, window, shift, tw, scale, dqb, dtab, ctx.drop)
        return dqkv, None if sq is not None else dqb, None if st is not None else dtab, None, None, None, None, None, None


def window_attention(qkv, qkv_bias, table, heads, window, shift, tw, scale, drop_p=0.0, training=True):
    return _WinAttn.apply(qkv, qkv_bias, table, heads, tuple(window), tuple(shift), tw, scale, float(drop_p) if training else 0.0)


# ----------------------------------------------------------------------------------------------------------------
class _PatchEmbed(Function):
    @staticmethod
    def forward(ctx, x_ncdhw, weight, bias, dtype):
        ctx.save_for_backward(x_ncdhw, weight, bias)
        ctx.params = (weight, bias)
        return ops.patch_embed_fwd(x_ncdhw, weight, bias, dtype)

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        sw, sb = _slot(ctx.params[0]), _slot(ctx.params[1])
        dw = sw if sw is not None else ops.zeros_f32(weight.shape, x.device)
        db = sb if sb is not None else (ops.zeros_f32(bias.shape, x.device) if bias is not None else None)
        ops.patch_embed_bwd(x, _rv(dy), dw, db)
        return None, None if sw is not None else dw, None if sb is not None else db, None


def patch_embed(x_ncdhw, weight, bias, dtype):
    return _PatchEmbed.apply(x_ncdhw, weight, bias, dtype)


class _Head(Function):
    """UnetOutBlock: 1x1x1 conv + bias, channels-last in -> NCDHW fp32 logits out (dynunet_block.py:273-292)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        return ops.head_fwd(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous().float()
        sw, sb = _slot(ctx.params[0]), _slot(ctx.params[1])
        dw = sw if sw is not None else ops.zeros_f32(weight.shape, x.device)
        db = sb if sb is not None else (ops.zeros_f32((weight.shape[0],), x.device) if ctx.has_bias else None)
        dx = ops.head_bwd(x, dy, weight, dw, db, want_dx=ctx.needs_input_grad[0])
        return dx, None if sw is not None else dw, None if sb is not None else db


def head(x, weight, bias):
    return _Head.apply(x, weight, bias)
