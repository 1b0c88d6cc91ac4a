"""Builds csrc/*.hip + *.cpp into csrc/libmiseg_hip.so for gfx950 with hipcc (in-tree, incremental).

The .so is git-ignored but travels to the GPU box with the gpurun snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
         "-I", os.path.join(ROOT, "include"), f'-DMISEG_COMPILED_ARCH="{ARCH}"']
SOURCES = ["common.cpp", "graphsplit.cpp", "norm.hip", "elementwise.hip", "gemm.hip", "mlp.hip", "conv3d.hip", "attention.hip", "attention_global.hip", "training.hip"]


def lib_path():
    return os.path.join(HERE, "libmiseg_hip.so")


def _deps(src):
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "miseg_hip.h"))
    return [src] + hdrs


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, verbose):
    obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
    if not _stale(obj, _deps(src)):
        return obj
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build_all(verbose=False, jobs=4):
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    out = lib_path()
    if _stale(out, objs):
        # --wrap: every kernel launch of the library goes through common.cpp::__wrap_hipLaunchKernel (in-situ timing, miseg_prof_arm)
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,--wrap=hipLaunchKernel", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build_all(verbose=True))
