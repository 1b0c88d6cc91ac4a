"""sweep of the (output-channel tiles per workgroup, chunk split) plan of the fast conv path on the small volumes (needs two experiment hooks at the end of
csrc/conv3d.hip::fwd96_plan that the shipped kernel does not carry:
    if (const char* e = getenv("MISEG_F96_NT")) { *nt = atoi(e); if (const char* k = getenv("MISEG_F96_KS")) *ksplit = min(atoi(k), nchunks); }
results of round 2: profiles/r02_f96_plan_sweep.txt)"""
import os, subprocess, sys
shapes = [(24, 96, 96), (24, 192, 96), (24, 96, 192), (12, 192, 192), (12, 384, 192), (12, 192, 384), (6, 384, 384), (6, 768, 384), (6, 384, 768), (3, 768, 768)]
code = r'''
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath("bench.py"))))
sys.path.insert(0, "scripts")
import torch
import bench_conv
S, Ci, Co = map(int, sys.argv[1:4])
bench_conv.run(S, Ci, Co, torch.bfloat16, what=("fwd",))
'''
for S, Ci, Co in shapes:
    for nt, ks in [(None, None), (1, 1), (1, 2), (1, 4), (1, 8), (1, 16), (3, 1), (3, 2), (3, 4), (3, 8), (3, 16)]:
        if ks is not None and ks > Ci // 48:
            continue
        env = dict(os.environ)
        if nt is not None:
            env["MISEG_F96_NT"], env["MISEG_F96_KS"] = str(nt), str(ks)
        r = subprocess.run([sys.executable, "-c", code, str(S), str(Ci), str(Co)], env=env, capture_output=True, text=True)
        out = [l for l in r.stdout.splitlines() if "fwd" in l]
        print(f"nt={nt} ks={ks}: {out[0] if out else r.stderr[-300:]}", flush=True)
