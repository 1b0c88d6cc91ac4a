"""roofline leg of bench.py: per-launch device time of the conv3 implicit-GEMM kernels measured with HIP events on the
launch stream during one extra forward+backward step - every launch ONCE, in place, in the schedule that is timed (round 4;
csrc/common.cpp::__wrap_hipLaunchKernel of the measurement build libmiseg_hip_prof.so) -, priced against the dense MFMA peak of the compute dtype (MI355X_MICROARCH.md: bf16 ~2.5 PFLOP/s dense, fp32 matrix
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
    """run step_fn() - an eager step exactly as the timed one issues it: side branch, fused MLP, GEMM statistics epilogues and grouped weight
    gradients ON - with the library's in-situ timing armed around every hooked call (hip/ops.py::_ProfRegion): each kernel is timed ONCE,
    where it runs, by its own dispatch timestamps.  Returns {kernel: [(ms of the call's first kernel, flops, bytes, ms of all its kernels)]}"""
    import ctypes as C
    from ..hip import lib as L
    # the measurement build of the same objects (libmiseg_hip_prof.so: linked with --wrap=hipLaunchKernel); the product library has no hook
    with L.profiling_library() as lib:
        lib.miseg_prof_read(None, None, 0)          # forget anything recorded earlier
        rec = []
        ops.PROFILE_HOOK = rec
        try:
            step_fn()
            torch.cuda.synchronize()
        finally:
            ops.PROFILE_HOOK = None
            lib.miseg_prof_arm(-1)
        cap = 64 * max(1, len(rec))
        tags, ms = (C.c_int * cap)(), (C.c_float * cap)()
        n = lib.miseg_prof_read(tags, ms, cap)
    if n > cap:
        raise RuntimeError(f"roofline: {n} launches recorded, room for {cap}")
    per = {}
    for i in range(n):
        if ms[i] < 0:
            raise RuntimeError("roofline: a launch's timestamps could not be read")
        per.setdefault(tags[i], []).append(ms[i])
    out = {}
    for tag, (name, flops, nbytes) in enumerate(rec):
        d = per.get(tag)
        if not d:
            continue          # a call that launched nothing (empty problem)
        out.setdefault(name, []).append((d[0], flops, nbytes, sum(d)))
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
    "conv3_wgrad": ("conv 3x3x3 weight gradient (direct, grouped and their slab sums)", "mfma"),
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
        # the side branch's throttled launches (hip/ops.py::_background: one workgroup per CU / a few CUs) run BESIDE the main stream's chain and
        # are slow on purpose: a line of their own, so that the main-stream classes still add up to the step and these are not mistaken for them
        a = agg.setdefault((key, "(background)" in name), [0, 0.0, 0.0, 0.0])
        a[0] += len(lst)
        a[1] += sum(t[3] for t in lst)          # every kernel of the call (a split launch's slab sum, a norm's reduce + apply)
        a[2] += sum(t[1] for t in lst)
        a[3] += sum(t[2] for t in lst)
    out = []
    for (key, bg), (n, ms, flops, nbytes) in agg.items():
        label, bound = CLASSES[key]
        if bg:
            label += " - side-branch launches in background form (throttled, overlapping the main stream)"
        if bound == "mfma":
            ach, peak, unit = flops / (ms * 1e-3) / 1e12, PEAK_TFLOPS[dtype], "TFLOP/s"
        else:
            ach, peak, unit = nbytes / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        out.append({"class": label, "bound": bound, "launches": n, "ms": ms, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak})
    return sorted(out, key=lambda d: -d["ms"])


def summarize(prof, dtype, counters=True):
    """counters: the committed PMC file (profiles/*_pmc_traffic.json) was collected over THIS workload (bench.py's default: configs[1], bf16,
    training step); any other workload reports `traffic: null` rather than another workload's bytes"""
    # the dominant kernel of every workload so far: the 3x3x3 implicit GEMM.  All its launches of the step, the side branch's throttled
    # ("background") ones included - one kernel symbol, what the rocprofv3 average of the same command covers; the other classes: `classes`
    groups = {}
    for name, lst in prof.items():
        if name.startswith("conv3_fwd"):
            groups.setdefault(name.replace(" (background)", ""), []).extend((t, "(background)" in name) for t in lst)
    name, lst = max(groups.items(), key=lambda kv: sum(t[0][0] for t in kv[1]))
    nbg = sum(1 for _, bg in lst if bg)
    lst = [t for t, _ in lst]
    tot_ms = sum(t[0] for t in lst)
    flops = sum(t[1] for t in lst)
    achieved = flops / (tot_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype]
    traffic, src = _pmc_traffic(name) if counters else (None, None)
    alg_bytes = sum(t[2] for t in lst) / len(lst)
    avg_s = tot_ms * 1e-3 / len(lst)
    # the 3x3x3 conv is reported against BOTH roofs - the matrix-core roof that bounds it (frac) and the HBM roof north_star
    # names (hbm_frac = algorithmic bytes per launch / launch time / 8 TB/s; hbm_frac_counter = the same with the PMC-counted bytes)
    return {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "hbm_achieved_GBs": alg_bytes / avg_s / 1e9, "hbm_peak_GBs": PEAK_HBM_GBS, "hbm_frac": alg_bytes / avg_s / 1e9 / PEAK_HBM_GBS,
            "hbm_frac_counter": (traffic / avg_s / 1e9 / PEAK_HBM_GBS) if traffic else None,
            "traffic": traffic, "traffic_unit": "bytes per launch on the L2's memory side (PMC FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits are counted)", "traffic_source": src,
            "algorithmic_bytes_per_launch": alg_bytes, "launches_per_step": len(lst), "background_launches": nbg, "avg_launch_ms": tot_ms / len(lst),
            "timing": "each launch once, in place (side branch, fused MLP, statistics epilogues, grouped weight gradients on), by its own dispatch timestamps "
                      "(hipExtLaunchKernel start/stop events on the launch stream), eager step",
            "flops_per_step": flops, "all_kernels_ms": {k: sum(t[3] for t in v) for k, v in prof.items()},
            "classes": classes(prof, dtype)}
