"""roofline leg of bench.py: per-launch device time of the conv3 implicit-GEMM kernels measured with HIP events on the
launch stream during one extra profiled forward+backward step, priced against the dense MFMA peak of the compute dtype
(MI355X_MICROARCH.md: bf16 ~2.5 PFLOP/s dense, fp32 matrix 157.3 TFLOP/s)."""
import torch

from ..hip import ops

PEAK_TFLOPS = {torch.bfloat16: 2500.0, torch.float32: 157.3}


def profile_step(step_fn):
    """run step_fn() with per-launch event timing of the hooked kernels; returns {kernel: [(ms, flops)]}"""
    rec = []
    ops.PROFILE_HOOK = rec
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops.PROFILE_HOOK = None
    out = {}
    for name, e0, e1, flops in rec:
        out.setdefault(name, []).append((e0.elapsed_time(e1), flops))
    return out


def summarize(prof, dtype):
    best = None
    for name, lst in prof.items():
        tot_ms = sum(t for t, _ in lst)
        if best is None or tot_ms > best[1]:
            best = (name, tot_ms, lst)
    name, tot_ms, lst = best
    flops = sum(f for _, f in lst)
    achieved = flops / (tot_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype]
    return {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": None, "launches_per_step": len(lst), "avg_launch_ms": tot_ms / len(lst),
            "flops_per_step": flops, "all_kernels_ms": {k: sum(t for t, _ in v) for k, v in prof.items()}}
