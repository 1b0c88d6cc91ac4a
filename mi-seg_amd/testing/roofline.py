"""roofline leg of bench.py: per-launch device time of the conv3 implicit-GEMM kernels measured with HIP events on the
launch stream during one extra profiled forward+backward step (each launch repeated ops.PROFILE_REPS times between the two
events), priced against the dense MFMA peak of the compute dtype (MI355X_MICROARCH.md: bf16 ~2.5 PFLOP/s dense, fp32 matrix
157.3 TFLOP/s).  `traffic` is the HBM bytes per launch of the same kernel from the committed rocprofv3 PMC passes
(profiles/*_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled on gfx950), null when absent."""
import glob
import json
import os

import torch

from ..hip import ops

PEAK_TFLOPS = {torch.bfloat16: 2500.0, torch.float32: 157.3}
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s


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
    for name, e0, e1, flops, nbytes in rec:
        out.setdefault(name, []).append((e0.elapsed_time(e1) / ops.PROFILE_REPS, flops, nbytes))
    return out


def _pmc_traffic(kernel):
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles")
    files = sorted(glob.glob(os.path.join(root, "*_pmc_traffic.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    for k, v in data.get("kernels", {}).items():
        if kernel.startswith(k):
            return v["hbm_bytes_per_launch"], os.path.relpath(files[-1], os.path.dirname(root))
    return None, None


CLASSES = {      # hooked launch name (prefix) -> (class, roof that bounds it)
    "conv3_fwd": ("conv 3x3x3 forward / data gradient (implicit GEMM)", "mfma"),
    "conv3_wgrad": ("conv 3x3x3 weight gradient", "mfma"),
    "instnorm": ("instance norms (statistics, apply, backward; conditional and plain)", "hbm"),
    "gemm_nt": ("linears / 1x1x1 convs / ConvTranspose GEMMs (NT, activations streamed)", "hbm"),
    "winattn": ("window attention forward + backward (QK^T, softmax, AV on MFMA; softmax on the vector unit)", "mfma"),
}


def classes(prof, dtype):
    """per kernel class of the SAME profiled step: launches, summed device time, achieved rate against the roof that bounds the class -
    HBM classes in GB/s of algorithmic bytes (each tensor a launch touches counted once), MFMA classes in TFLOP/s"""
    agg = {}
    for name, lst in prof.items():
        key = next((k for k in CLASSES if name.startswith(k)), None)
        if key is None:
            continue
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += len(lst)
        a[1] += sum(t[0] for t in lst)
        a[2] += sum(t[1] for t in lst)
        a[3] += sum(t[2] for t in lst)
    out = []
    for key, (n, ms, flops, nbytes) in agg.items():
        label, bound = CLASSES[key]
        if bound == "mfma":
            ach, peak, unit = flops / (ms * 1e-3) / 1e12, PEAK_TFLOPS[dtype], "TFLOP/s"
        else:
            ach, peak, unit = nbytes / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        out.append({"class": label, "bound": bound, "launches": n, "ms": ms, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak})
    return sorted(out, key=lambda d: -d["ms"])


def summarize(prof, dtype):
    best = None
    for name, lst in prof.items():
        if not name.startswith("conv3_fwd"):      # the dominant kernel of every workload so far; the other classes are listed in `classes`
            continue
        tot_ms = sum(t[0] for t in lst)
        if best is None or tot_ms > best[1]:
            best = (name, tot_ms, lst)
    name, tot_ms, lst = best
    flops = sum(t[1] for t in lst)
    achieved = flops / (tot_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype]
    traffic, src = _pmc_traffic(name)
    alg_bytes = sum(t[2] for t in lst) / len(lst)
    avg_s = tot_ms * 1e-3 / len(lst)
    # the 3x3x3 conv is reported against BOTH roofs - the matrix-core roof that bounds it (frac) and the HBM roof north_star
    # names (hbm_frac = algorithmic bytes per launch / launch time / 8 TB/s; hbm_frac_counter = the same with the PMC-counted bytes)
    return {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "hbm_achieved_GBs": alg_bytes / avg_s / 1e9, "hbm_peak_GBs": PEAK_HBM_GBS, "hbm_frac": alg_bytes / avg_s / 1e9 / PEAK_HBM_GBS,
            "hbm_frac_counter": (traffic / avg_s / 1e9 / PEAK_HBM_GBS) if traffic else None,
            "traffic": traffic, "traffic_unit": "bytes per launch on the L2's memory side (PMC FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits are counted)", "traffic_source": src,
            "algorithmic_bytes_per_launch": sum(t[2] for t in lst) / len(lst), "launches_per_step": len(lst), "avg_launch_ms": tot_ms / len(lst),
            "flops_per_step": flops, "all_kernels_ms": {k: sum(t[0] for t in v) for k, v in prof.items()},
            "classes": classes(prof, dtype)
            # (the profiled step runs WITHOUT the model's side branch - networks/nets/swin_unetr.py: one stream, every launch in its normal
            # form, nothing beside anything - so that the figure covers the same 38 launches round after round; what the real step
            # throttles is reported by bench.py as `side_branch`)
            }
