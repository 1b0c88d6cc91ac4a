"""micro-benchmark of the GEMM kernels on the headline shapes (hipGraph-captured loops; algorithmic bytes / time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops

def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

dt = torch.bfloat16
if len(sys.argv) > 1 and sys.argv[1] == "small":      # the deep-stage / ViT linears (gemm_nt_small_kernel): weight-bound
    print("NT small  y[M,N] = x[M,K] w[N,K]^T")
    for M, N, K in [(216, 768, 768), (216, 2304, 768), (216, 3072, 768), (216, 768, 2304), (216, 768, 3072), (1728, 768, 192), (1728, 192, 768), (1728, 576, 192),
                    (216, 1536, 384), (216, 384, 1536), (216, 1152, 384), (27, 3072, 768), (27, 768, 3072), (27, 2304, 768)]:
        x = torch.randn(M, K, device="cuda").to(dt); w = torch.randn(N, K, device="cuda").to(dt)
        t = timed(lambda: ops.gemm_nt(x, w))
        print(f"  M {M:5d} N {N:5d} K {K:5d}: {t:7.1f} us  weights {2.0 * N * K / t / 1e6:5.2f} TB/s  {2.0*M*N*K/t/1e6:7.1f} TF", flush=True)
    sys.exit(0)
print("NT  y[M,N] = x[M,K] w[N,K]^T")
for M, N, K in [(884736, 48, 96), (884736, 96, 48), (110592, 192, 48), (110592, 48, 192), (110592, 144, 48), (110592, 48, 48), (110592, 384, 96),
                (13824, 384, 96), (13824, 96, 384), (13824, 288, 96), (1728, 768, 192), (1728, 192, 768), (216, 1536, 384), (216, 384, 1536), (216, 1152, 384), (27, 3072, 768)]:
    x = torch.randn(M, K, device="cuda").to(dt); w = torch.randn(N, K, device="cuda").to(dt)
    t = timed(lambda: ops.gemm_nt(x, w))
    by = 2.0 * (M * K + M * N + N * K)
    print(f"  M {M:7d} N {N:5d} K {K:5d}: {t:7.1f} us  {by/t/1e6:5.2f} TB/s  {2.0*M*N*K/t/1e6:7.1f} TF", flush=True)
print("TN  dw[M,N] = dy[K,M]^T x[K,N]")
for K, M, N in [(884736, 48, 96), (110592, 192, 48), (110592, 48, 192), (110592, 144, 48), (110592, 384, 96), (13824, 384, 96), (13824, 768, 192), (1728, 768, 192), (216, 1536, 384)]:
    a = torch.randn(K, M, device="cuda").to(dt); b = torch.randn(K, N, device="cuda").to(dt)
    t = timed(lambda: ops.gemm_tn(a, b))
    by = 2.0 * (K * M + K * N) + 4.0 * M * N
    print(f"  K {K:7d} M {M:5d} N {N:5d}: {t:7.1f} us  {by/t/1e6:5.2f} TB/s", flush=True)
