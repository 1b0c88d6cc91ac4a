"""one eager fwd+bwd step of the headline model under the torch profiler: which aten ops (copies, adds, fills) still run beside the
HIP library, with their input shapes.   gpurun -- python scripts/list_aten_ops.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ge.load_package()
    model = bench.build_model(torch.bfloat16)
    from mi_seg_amd.runtime.arena import ParamArena
    arena = ParamArena([p for p in model.parameters() if p.requires_grad], torch.bfloat16)
    x = torch.rand(1, 1, 96, 96, 96, device=dev)
    cot = torch.randn(1, 6, 96, 96, 96, device=dev)

    def step():
        arena.begin_step()
        model(x, [0]).backward(cot)
        arena.publish()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        step()
    torch.cuda.synchronize()
    rows = [(e.key, e.count) for e in prof.key_averages() if e.key.startswith("aten::")]
    for k, c in sorted(rows, key=lambda r: -r[1])[:30]:
        print(f"{c:5d} {k}")
    keys = ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::zero_", "aten::fill_", "aten::clone", "aten::cat", "aten::sum",
            "aten::_to_copy", "aten::contiguous", "aten::index_select", "aten::roll", "aten::pad", "aten::constant_pad_nd")
    print("--- by shape")
    for e in prof.key_averages(group_by_input_shape=True):
        if e.key in keys:
            print(f"{e.count:4d} {e.key:18s} {e.input_shapes}")


main()
