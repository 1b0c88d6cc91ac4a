"""Can a dist.all_reduce (RCCL) be CAPTURED as nodes of a hipGraph between two of our kernels?  (ProcessGroupNCCL joins its
stream to a capture with ordinary events - no external event needed.)  One rank, run on a 1-GPU box:
    python scripts/debug/rccl_capture_probe.py
Prints what happened: capture ok / error text, replay results against the eager sequence."""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops

n = 1 << 24
a = torch.randn(n, device="cuda")
b = torch.zeros(n, device="cuda")
c = torch.zeros(n, device="cuda")
# warm RCCL (communicator setup must not land inside the capture)
dist.all_reduce(a.clone())
torch.cuda.synchronize()


def seq(async_op):
    ops.add(a.view(-1, 4), a.view(-1, 4), out=b.view(-1, 4))          # b = 2a (our kernel)
    w = dist.all_reduce(b, op=dist.ReduceOp.AVG, async_op=async_op)      # one rank: b unchanged
    ops.add(a.view(-1, 4), a.view(-1, 4), out=c.view(-1, 4))          # independent kernel that may overlap the collective
    if async_op:
        w.wait()
    ops.add(b.view(-1, 4), c.view(-1, 4), out=c.view(-1, 4))          # c = 4a


for async_op in (False, True):
    b.zero_(); c.zero_()
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            seq(async_op)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            seq(async_op)
        print(f"async_op={async_op}: capture OK")
        for rep in range(3):
            a.mul_(1.5)
            b.zero_(); c.zero_()
            g.replay()
            torch.cuda.synchronize()
            print(f"  replay {rep}: c == 4a: {bool(torch.equal(c, 4 * a))}, b == 2a: {bool(torch.equal(b, 2 * a))}")
    except Exception:
        print(f"async_op={async_op}: capture FAILED")
        traceback.print_exc()
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
dist.destroy_process_group()
print("probe done")
