"""Builds csrc/*.hip + *.cpp for gfx950 with hipcc (in-tree, incremental) into

  csrc/libmiseg_hip.so        the product library (include/miseg_hip.h + the experiment entry points of include/miseg_hip_debug.h)
  csrc/libmiseg_hip_prof.so   the measurement build of the SAME objects: common.cpp compiled with -DMISEG_PROF_WRAP and the link done with
                              -Wl,--wrap=hipLaunchKernel, so that every kernel launch of the library can be timed in place (bench.py's
                              roofline leg loads it for one eager step: hip/lib.py::profiling_library).  Nothing else differs.

Both carry the sha256 of the sources they were compiled from (miseg_source_digest); hip/lib.py::load() refuses a library that is older
than the sources beside it.  The .so files are git-ignored but travel to the GPU box with the gpurun snapshot.  Every build_all() writes
what it did - which objects were compiled, which were reused, the compiler, the digest - to profiles/build_record.json."""
import hashlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
         "-I", os.path.join(ROOT, "include"), f'-DMISEG_COMPILED_ARCH="{ARCH}"']
SOURCES = ["common.cpp", "graphsplit.cpp", "norm.hip", "elementwise.hip", "gemm.hip", "mlp.hip", "conv3d.hip", "attention.hip", "attention_global.hip", "training.hip"]


def lib_path(prof=False):
    return os.path.join(HERE, "libmiseg_hip_prof.so" if prof else "libmiseg_hip.so")


def _source_files():
    fs = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith((".hip", ".cpp", ".h"))]
    inc = os.path.join(ROOT, "include")
    fs += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    return fs


def source_digest():
    """sha256 over (file name, content) of every csrc/*.{hip,cpp,h} and include/*.h in name order"""
    h = hashlib.sha256()
    for f in _source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


def _deps(src):
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    inc = os.path.join(ROOT, "include")
    hdrs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return [src] + hdrs


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, verbose, log, obj_name=None, extra=(), force=False):
    obj = os.path.join(HERE, "build", (obj_name or os.path.basename(src)) + ".o")
    if not force and not _stale(obj, _deps(src)):
        log["reused"].append(os.path.basename(obj))
        return obj
    cmd = [HIPCC] + FLAGS + list(extra) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    log["compiled"].append({"object": os.path.basename(obj), "seconds": round(time.time() - t0, 1)})
    return obj


def _link(out, objs, verbose, log, extra=()):
    if not _stale(out, objs):
        log["reused"].append(os.path.basename(out))
        return
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}"] + list(extra) + ["-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    log["linked"].append(os.path.basename(out))


def build_all(verbose=False, jobs=4, force=None):
    """force (default: MISEG_FORCE_REBUILD=1): recompile every object even when it is newer than its sources"""
    force = bool(os.environ.get("MISEG_FORCE_REBUILD")) if force is None else force
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    log = {"compiled": [], "reused": [], "linked": []}
    digest = source_digest()
    stamp = os.path.join(HERE, "build", "source_digest.txt")
    # the two common.cpp objects carry the digest: they are recompiled whenever ANY source changed (seconds), the others only when they did
    digest_moved = not os.path.exists(stamp) or open(stamp).read().strip() != digest
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    common = os.path.join(HERE, "common.cpp")
    dig = [f'-DMISEG_SOURCE_DIGEST="{digest}"']
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose, log, extra=dig if s == common else (), force=force or (s == common and digest_moved)), srcs))
    prof_common = _compile(common, verbose, log, obj_name="common_prof.cpp", extra=dig + ["-DMISEG_PROF_WRAP"], force=force or digest_moved)
    _link(lib_path(), objs, verbose, log)
    # --wrap: every kernel launch of the measurement build goes through common.cpp::__wrap_hipLaunchKernel (in-situ timing, miseg_prof_arm)
    _link(lib_path(prof=True), [prof_common if o.endswith("common.cpp.o") else o for o in objs], verbose, log, extra=["-Wl,--wrap=hipLaunchKernel"])
    open(stamp, "w").write(digest)
    _record(log, digest, force)
    return lib_path()


def _record(log, digest, force):
    """profiles/build_record.json: what the last build_all() of this tree did (the driver's build() check runs it on the CPU box)"""
    try:
        ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.strip().splitlines()
        rec = {"when": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "hipcc": ver[0] if ver else "?", "arch": ARCH, "forced": bool(force),
               "source_digest": digest, "compiled": log["compiled"], "linked": log["linked"], "reused": sorted(log["reused"]),
               "libraries": {os.path.basename(p): {"bytes": os.path.getsize(p), "sha256_16": hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]}
                             for p in (lib_path(), lib_path(prof=True)) if os.path.exists(p)}}
        # keep the last record in which something was compiled beside the latest one: a later all-reused run does not erase the evidence
        path = os.path.join(ROOT, "profiles", "build_record.json")
        prev = {}
        if os.path.exists(path):
            try:
                prev = json.load(open(path))
            except Exception:
                prev = {}
        last_compile = rec if rec["compiled"] else prev.get("last_build_that_compiled", prev if prev.get("compiled") else None)
        if last_compile is not None and "last_build_that_compiled" in last_compile:
            last_compile = {k: v for k, v in last_compile.items() if k != "last_build_that_compiled"}
        rec["last_build_that_compiled"] = last_compile if last_compile is not rec else None
        os.makedirs(os.path.dirname(path), exist_ok=True)
        json.dump(rec, open(path, "w"), indent=1)
    except OSError:
        pass      # a read-only tree (the GPU box never builds)


if __name__ == "__main__":
    print(build_all(verbose=True))
