"""the grouped small weight-gradient GEMMs of one step (gemm_tn_group), timed per family"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops
dt = torch.bfloat16
fam = {
    "K=27": [(3072, 768, 27), (768, 3072, 27)],
    "K=216": [(1536, 384, 216)] * 3 + [(1152, 384, 216)] * 2 + [(384, 768, 216)] + [(384, 1536, 216)] * 3 + [(384, 384, 216)] * 2,
    "K=1728": [(768, 192, 1728)] * 3 + [(576, 192, 1728)] * 2 + [(192, 384, 1728)] + [(192, 768, 1728)] * 3 + [(192, 192, 1728)] * 2,
}
fam["all"] = fam["K=27"] + fam["K=216"] + fam["K=1728"]
fam["vit-b (C-UNETR, one launch of 24)"] = [(2304, 768, 216), (768, 768, 216), (3072, 768, 216), (768, 3072, 216)] * 6
for name, probs in fam.items():
    items = [(torch.randn(K, M, device="cuda").to(dt), torch.randn(K, N, device="cuda").to(dt), torch.zeros(M, N, device="cuda")) for M, N, K in probs]
    def run():
        ops.DEFAULT_QUEUES = ops.StepQueues()
        for a, b, o in items:
            ops.gemm_tn(a, b, out=o, accumulate=2 if "zeroed" in sys.argv else True)
        ops.DEFAULT_QUEUES.flush()
        ops.DEFAULT_QUEUES = None
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5): run()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5 * 1e6
    out_mb = sum(M * N for M, N, K in probs) * 4 / 1e6
    fl = sum(2.0 * M * N * K for M, N, K in probs) / 1e9
    print(f"{name:7s}: {len(probs):2d} problems {t:7.1f} us   outputs {out_mb:6.1f} MB  {fl:6.2f} GFLOP")
