"""The open item of DESIGN.md section 3: twice in ~20 first-process runs on fresh boxes the arena-vs-plain test saw
bf16 logits of a 32^3 net 1-2 % apart in one step, never in-process afterwards.  This script starts N FRESH processes, each of which runs
the test body of tests/test_hip_modules.py::test_param_arena_matches_plain_autograd (scripts/debug/arena_repro.py: every net, both dtypes,
arena steps 0..2 against the plain steps, logits compared bitwise) as the first GPU work of that process, and counts mismatching steps.

    python scripts/debug/fresh_process_stress.py [N=50] [--env KEY=VALUE ...]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50
env = dict(os.environ)
for kv in sys.argv[1:]:
    if "=" in kv and not kv.startswith("-"):
        k, v = kv.split("=", 1)
        env[k] = v
bad, t0 = [], time.time()
for i in range(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts/debug/arena_repro.py"), "1"], capture_output=True, text=True, env=env)
    lines = [l for l in r.stdout.splitlines() if "max |logit diff|" in l]
    if r.returncode != 0 or lines:
        bad.append((i, r.returncode, lines, r.stderr[-400:] if r.returncode else ""))
    if i % 5 == 4:
        print(f"{i + 1} processes, {len(bad)} with a mismatch, {time.time() - t0:.0f} s", flush=True)
for b in bad:
    print("MISMATCH in process", *b)
print(f"{n} fresh processes: {len(bad)} with a mismatching step")
sys.exit(1 if bad else 0)
