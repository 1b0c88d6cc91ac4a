"""profiles/<round>_step_kernels.txt from the timeline scripts/trace_step.sh leaves in gpurun_out/<tag>_timeline.txt: the kernels of ONE replayed step
(one period between two arena fills) counted by name and listed in issue order.  python scripts/step_kernels.py <tag> <round> [ms without profiler]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    plain_ms = sys.argv[3] if len(sys.argv) > 3 else None
    lines = open(os.path.join(ROOT, "gpurun_out", f"{tag}_timeline.txt")).read().split("\n")
    gaps = [l for l in lines if " gap " in l]
    idx = [i for i, l in enumerate(gaps) if "fill_words_kernel" in l]
    if len(idx) > 2:                  # several replays in the file: the second one
        seg = gaps[idx[1]:idx[2]]
        period_us = float(gaps[idx[2]].split()[0]) - float(seg[0].split()[0])
    else:                             # one period (step_timeline.py --period): it may start inside the previous step's tail - rotate to the arena fill
        period_us = float(gaps[-1].split()[0]) + float(gaps[-1].split()[1]) - float(gaps[0].split()[0])
        seg = gaps[idx[0]:] + gaps[:idx[0]]
    names = [l.split("gap")[1].split()[1] for l in seg]
    c = collections.Counter(names)
    t0 = float(seg[0].split()[0])
    copies = c.get("__amd_rocclr_copyBuffer", 0)
    out = ["One replayed step of the headline bench (python bench.py, hipGraph), kernels in issue order from rocprofv3 --kernel-trace",
           f"(scripts/trace_step.sh {tag}; the profiler slows the replay: {period_us / 1000:.2f} ms per step under it" + (f", {plain_ms} without)." if plain_ms else ")."),
           f"{len(names)} kernels in the period, {len(names) - copies} of them nodes of the graph (the __amd_rocclr_copyBuffer is the input copy in front of the replay).",
           "rocprofv3 --stats of the whole bench process divided by its steps also counts the eager warm-up / capture / check steps and torch's own kernels of the checks.",
           "", "count  kernel"]
    out += ["%5d  %s" % (v, k) for k, v in c.most_common()]
    out += ["", "start_us  dur_us  overlap  kernel  grid"]
    for l in seg:
        f = l.split()
        out.append("%8.1f %7.1f %8s  %-34s %s" % ((float(f[0]) - t0) % period_us if float(f[0]) < t0 else float(f[0]) - t0, float(f[1]), f[3], f[4], f[6]))
    path = os.path.join(ROOT, "profiles", f"{rnd}_step_kernels.txt")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:3]))
    print("wrote", path)


if __name__ == "__main__":
    main()
