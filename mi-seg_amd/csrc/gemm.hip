// MFMA GEMMs for the linears / 1x1x1 convs / transposed-conv / tiny-grid conv paths.
//   NT:  C[M][N] = A[M][K] * B[N][K]^T        (activations x packed weights; k contiguous in both)
//   TN:  C[M][N] = A[K][M]^T * B[K][N]        (weight gradients: reduction over tokens, the row index of both)
// bf16 uses v_mfma_f32_16x16x32_bf16, fp32 the exact v_mfma_f32_16x16x4_f32 (same LDS byte layout: a lane's
// operand is always one 16-byte chunk of k).  The MFMA is issued with the operands swapped (weights as "A") so a
// lane ends up with 4 CONSECUTIVE output channels of one token -> one 8/16-byte store.
#include "common.h"

namespace miseg {

template <class T> struct Mma;
template <> struct Mma<bf16> {
  static constexpr int KPC = 8;  // k elements per 16-byte chunk
  __device__ static __forceinline__ void run(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KPC = 4;
  __device__ static __forceinline__ void run(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
  }
};

// Exact-erf GELU (MONAI MLPBlock, swin_transformer_block.py:176-205) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far
// below the fp32 parity tolerance): one v_rcp, one v_exp and five FMAs instead of the ~40-instruction library erff - the GELU epilogues
// of the 110592 x 192 linears were VALU-bound (48 us for 95 MB).  cdf(t) = 0.5 (1 + erf(t / sqrt 2)); e = exp(-t^2 / 2) is shared with
// the density term of the derivative.
__device__ __forceinline__ float gelu_cdf(float t, float& e) {
  const float a = fabsf(t) * 0.70710678118654752f;
  const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));      // v_rcp_f32 (1 ulp); __frcp_rn is the ten-instruction IEEE division
  float p = fmaf(1.061405429f, k, -1.453152027f);
  p = fmaf(p, k, 1.421413741f);
  p = fmaf(p, k, -0.284496736f);
  p = fmaf(p, k, 0.254829592f);
  e = __expf(-a * a);
  const float half_tail = 0.5f * p * k * e;          // 0.5 erfc(|t| / sqrt 2)
  return t >= 0.f ? 1.f - half_tail : half_tail;
}
__device__ __forceinline__ float gelu_erf(float t) { float e; return t * gelu_cdf(t, e); }
__device__ __forceinline__ float gelu_erf_grad(float t) { float e; const float c = gelu_cdf(t, e); return fmaf(t * 0.39894228040143268f, e, c); }
// fp32 parity mode (output type float): the library erff / expf, i.e. what torch's CPU kernels evaluate - the short forms above put the
// parity mode's logits 1.7x further from the float64 run than the reference's own fp32 run; bf16 keeps the short forms
template <class TO> __device__ __forceinline__ float gelu_fwd_t(float t) {
  if constexpr (sizeof(TO) == 4) return 0.5f * t * (1.f + erff(t * 0.70710678118654752f));
  else return gelu_erf(t);
}
template <class TO> __device__ __forceinline__ float gelu_grad_t(float t) {
  if constexpr (sizeof(TO) == 4) return fmaf(t * 0.39894228040143268f, expf(-0.5f * t * t), 0.5f * (1.f + erff(t * 0.70710678118654752f)));
  else return gelu_erf_grad(t);
}

// epilogue of the NT kernels: z = acc + bias ; mode 2: z *= gelu'(aux) ; mode 1: aux = z (the pre-activation, for the backward) ;
// act ; + res.  aux / res are [M][N] row views in the output dtype.
struct Epi {
  const float* bias; int act;
  const void* res; int64_t ldres;
  void* aux; int64_t ldaux;
  int mode;
  double* stat;      // streaming kernel, STAT instantiation: instance-norm statistics of the (rounded) output, one sample (M rows)
  int sd, sh, sw, sco;   // streaming kernel, SCAT instantiation: rows are the voxels of a [.., sd, sh, sw] grid, columns are (j, co) with
                         // j = 4 jd + 2 jh + jw: element (voxel, j, co) goes to row ((2d+jd), (2h+jh), (2w+jw)) of the doubled grid, column co
  // streaming kernel, STAT == 2 (round 5): the output is the gradient with respect to the OUTPUT of an instance norm whose raw input is
  // bx [M][N]; `stat` then receives that norm's backward sums (sum q, sum q * xhat), xhat = (bx - mean) * rstd from bstat_in (the norm's
  // forward statistics) - what instnorm_bwd_reduce_kernel computes from the stored tensor in a launch of its own
  const void* bx; int64_t ldbx; const double* bstat_in; float beps;
  // streaming kernel, ANORM (round 5): A is the RAW input of a (conditional) instance norm over its M rows (one sample) and is normalised as
  // it is loaded - fma(x, sc, sh) with the scale / shift of instnorm_apply_kernel, rounded to bf16 exactly as that kernel stores it; an_out
  // (optional) receives norm(A) once (column group 0), for the weight-gradient product of the backward pass
  const double* an_stat; const int32_t* an_styles; const float* an_gamma[MISEG_MAX_STYLES]; const float* an_beta[MISEG_MAX_STYLES]; float an_eps;
  void* an_out; int64_t ld_an_out;
};

// mean and 1 / sqrt(var + eps) of one channel from its fp64 (sum, sum of squares) over S rows: the arithmetic of norm.hip::mean_rstd, bit for bit
__device__ __forceinline__ void gemm_mean_rstd(double sum, double sq, double invS, float eps, float& m, float& rs) {
  const double mu = sum * invS;
  double var = fma(sq, invS, -mu * mu);
  if (var < 0.0) var = 0.0;
  m = (float)mu;
  rs = 1.0f / sqrtf((float)var + eps);
}
// sum over the 16 replicas of a one-sample statistics buffer [16][1][C][2]
__device__ __forceinline__ void gemm_gather_stat(const double* __restrict__ stat, int C, int ch, double& sum, double& sq) {
  sum = 0.0; sq = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) { sum += stat[((int64_t)r * C + ch) * 2]; sq += stat[((int64_t)r * C + ch) * 2 + 1]; }
}

template <class TO>
__device__ __forceinline__ float epi_one(float x, int m, int n, const Epi& e) {
  if (e.bias) x += e.bias[n];
  if (e.mode == 2) x *= gelu_grad_t<TO>(to_f32(reinterpret_cast<const TO*>(e.aux)[(int64_t)m * e.ldaux + n]));
  if (e.mode == 1) reinterpret_cast<TO*>(e.aux)[(int64_t)m * e.ldaux + n] = from_f32<TO>(x);
  if (e.act == MISEG_ACT_GELU) x = gelu_fwd_t<TO>(x);
  if (e.res) x += to_f32(reinterpret_cast<const TO*>(e.res)[(int64_t)m * e.ldres + n]);
  return x;
}

// 4 consecutive columns n..n+3 of row m (bf16 output, 8-byte aligned views): bias already added by the caller
__device__ __forceinline__ f32x4 epi_vec4_bf16(f32x4 v, int m, int n, const Epi& e, bool gelu) {
  if (e.mode == 2) {
    const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(e.aux) + (int64_t)m * e.ldaux + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_grad((float)h[r]);
  }
  if (e.mode == 1)
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(e.aux) + (int64_t)m * e.ldaux + n) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  if (gelu) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
  if (e.res) {
    const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(e.res) + (int64_t)m * e.ldres + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
  }
  return v;
}

// 16-byte chunk of k for row `row` starting at element k: zero-filled outside [0,rows) x [0,K)
template <class T>
__device__ __forceinline__ typename Vec16<T>::type load_chunk(const T* base, int64_t ld, int row, int rows, int k, int K, bool vec_ok) {
  typedef typename Vec16<T>::type VT;
  constexpr int N = Vec16<T>::N;
  VT v;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = from_f32<T>(0.f);
  if (row < rows && k < K) {
    const T* p = base + (int64_t)row * ld + k;
    if (vec_ok && k + N <= K) v = *reinterpret_cast<const VT*>(p);
    else {
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (k + i < K) v[i] = p[i];
    }
  }
  return v;
}

template <class TO>
__device__ __forceinline__ void store_out4(TO* C, int64_t ldc, int m, int n, int M, int N, f32x4 v, const Epi& e, int mode /*0 store,1 add,2 atomic*/) {
  if (m >= M) return;
  if constexpr (std::is_same<TO, float>::value) {
    // plain fp32 store / add of four consecutive columns as one 16-byte access (the weight-gradient tiles: no epilogue pieces)
    if (mode != 2 && !e.bias && !e.res && e.mode == 0 && e.act == MISEG_ACT_NONE && n + 3 < N && (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
      f32x4* p4 = reinterpret_cast<f32x4*>(C + (int64_t)m * ldc + n);
      *p4 = mode == 1 ? *p4 + v : v;
      return;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (n + r < N) {
      const float x = epi_one<TO>(v[r], m, n + r, e);
      TO* p = C + (int64_t)m * ldc + n + r;
      if constexpr (std::is_same<TO, float>::value) {
        if (mode == 2) atomicAdd(p, x);
        else if (mode == 1) *p += x;
        else *p = x;
      } else {
        *p = from_f32<TO>(x);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ NT
template <class T, class TO, int NT>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, TO* __restrict__ C, int64_t ldc,
                                                      int M, int N, int K, Epi epi, int mode, bool vec_a, bool vec_b,
                                                      int k_per_split) {
  typedef typename Vec16<T>::type VT;
  constexpr int KPC = Mma<T>::KPC;
  constexpr int BM = 128, BN = 16 * NT, CH = 8 /*16B chunks per row per stage*/, ROWB = CH * 16 + 16 /*bytes, padded*/;
  __shared__ __attribute__((aligned(16))) char lds[(BM + BN) * ROWB];
  char* lA = lds;
  char* lB = lds + BM * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
  f32x4 acc[2][NT], tot[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int A_ITERS = BM * CH / 256, B_CHUNKS = BN * CH, B_ITERS = (B_CHUNKS + 255) / 256;
  VT ra[A_ITERS], rb[B_ITERS];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int c = tid + i * 256, row = c / CH, ch = c % CH;
      ra[i] = load_chunk<T>(A, lda, m0 + row, M, k0 + ch * KPC, kend, vec_a);
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int c = tid + i * 256, row = c / CH, ch = c % CH;
      if (c < B_CHUNKS) rb[i] = load_chunk<T>(B, ldb, n0 + row, N, k0 + ch * KPC, kend, vec_b);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int c = tid + i * 256, row = c / CH, ch = c % CH;
      *reinterpret_cast<VT*>(lA + row * ROWB + ch * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int c = tid + i * 256, row = c / CH, ch = c % CH;
      if (c < B_CHUNKS) *reinterpret_cast<VT*>(lB + row * ROWB + ch * 16) = rb[i];
    }
  };
  const int fi = lane & 15, fq = lane >> 4;
  constexpr int KSTAGE = CH * KPC;
  gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += KSTAGE) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (k0 + KSTAGE < kend) gload(k0 + KSTAGE);
#pragma unroll
    for (int ks = 0; ks < CH / 4; ++ks) {
      VT af[2], bfr[NT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const VT*>(lA + (wave * 32 + mt * 16 + fi) * ROWB + (ks * 4 + fq) * 16);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bfr[nt] = *reinterpret_cast<const VT*>(lB + (nt * 16 + fi) * ROWB + (ks * 4 + fq) * 16);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Mma<T>::run(acc[mt][nt], bfr[nt], af[mt]);
    }
    if constexpr (std::is_same<T, float>::value) {
      // fp32 = the PARITY mode: blocked (two-level) accumulation - the 32 products of a stage join the running sum ONCE, as in the fp32
      // convolution (round 3).  One running fp32 sum over K = 4096 (C-UNETR's perceptron patch embedding, vit.py:101-110) / 3072 (its MLP)
      // put the net's logits 1.5 - 1.9 x further from the float64 run than the reference's own fp32 run on all three input seeds
      // (round 5, tests/test_hip_modules.py::test_vs_truth_over_seeds; torch's CPU GEMM sums in blocks too)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { tot[mt][nt] += acc[mt][nt]; acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
  }
  if (blockIdx.z != 0) epi.bias = nullptr;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      store_out4<TO>(C, ldc, m0 + wave * 32 + mt * 16 + fi, n0 + nt * 16 + fq * 4, M, N, std::is_same<T, float>::value ? tot[mt][nt] : acc[mt][nt], epi, mode);
}

// ------------------------------------------------------------------------------------------------ NT, K == 1
// The 1x1x1 shortcut convolution of the stem block (dynunet_block.py:89, one input channel): an outer product, one 16-byte store per
// lane.  The tiled kernel spent 92 us on it (a 64-wide k stage for one k), this is one pass over the output.
template <class T>
__global__ void __launch_bounds__(256) gemm_nt_k1_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw, T* __restrict__ C, int64_t ldc,
                                                         int M, int N, const float* __restrict__ bias) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::N;
  const int nv = N / VN;
  const int64_t total = (int64_t)M * nv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / nv;
    const int n = (int)(i - m * nv) * VN;
    const float a = to_f32(A[m * lda]);
    VT o;
#pragma unroll
    for (int e = 0; e < VN; ++e) o[e] = from_f32<T>(a * to_f32(W[(int64_t)(n + e) * ldw]) + (bias ? bias[n + e] : 0.f));
    *reinterpret_cast<VT*>(C + m * ldc + n) = o;
  }
}

// the same with the instance-norm statistics of the rounded output (one sample = all M rows, N <= 128, no bias needed by the caller but
// honoured): a thread keeps one 16-byte column group, walks rows, and its partial sums meet in LDS; one fp64 atomic per column and workgroup
template <class T>
__global__ void __launch_bounds__(256) gemm_nt_k1_stat_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw, T* __restrict__ C, int64_t ldc,
                                                              int M, int N, const float* __restrict__ bias, double* __restrict__ stat) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::N;
  __shared__ float red[2][256][VN + 1];
  const int nv = N / VN, rpt = 256 / nv;          // column groups per row, rows per block pass
  const int cg = threadIdx.x % nv, rl = threadIdx.x / nv;
  const bool live = rl < rpt;
  float wv[VN], bv[VN], ssum[VN], ssq[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) {
    wv[e] = to_f32(W[(int64_t)(cg * VN + e) * ldw]);
    bv[e] = bias ? bias[cg * VN + e] : 0.f;
    ssum[e] = ssq[e] = 0.f;
  }
  if (live) {
    for (int64_t m = (int64_t)blockIdx.x * rpt + rl; m < M; m += (int64_t)gridDim.x * rpt) {
      const float a = to_f32(A[m * lda]);
      VT o;
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        o[e] = from_f32<T>(a * wv[e] + bv[e]);
        const float q = to_f32(o[e]);
        ssum[e] += q;
        ssq[e] = fmaf(q, q, ssq[e]);
      }
      if (C) *reinterpret_cast<VT*>(C + m * ldc + cg * VN) = o;       // C == nullptr: statistics only (miseg_rank1_stats)
    }
  }
#pragma unroll
  for (int e = 0; e < VN; ++e) { red[0][threadIdx.x][e] = ssum[e]; red[1][threadIdx.x][e] = ssq[e]; }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * N; o += 256) {
    const int k = o / N, col = o - k * N, g = col / VN, e = col - g * VN;
    float tot = 0.f;
    for (int r = 0; r < rpt; ++r) tot += red[k][r * nv + g][e];
    atomicAdd(stat + ((int64_t)(blockIdx.x & 15) * N + col) * 2 + k, (double)tot);       // [16 replicas][B = 1][N][2]
  }
}

// ------------------------------------------------------------------------------------------------ NT, streaming
// Tall activations x small weight (the Swin linears of the high-resolution stages, 1x1x1 convs, the ConvTranspose GEMM):
// M ~ 1e5..1e6 rows, K <= 192, N <= ~400.  HBM-bound: every byte of A is read ONCE (all N columns are produced by the
// wave that loaded the rows), the whole weight sits in LDS for the lifetime of a persistent workgroup, a wave streams
// 32-row tiles with the next tile's operands already in flight.  A fragments go global -> registers directly (a lane's
// MFMA operand is 16 contiguous bytes of its row), so the only LDS traffic is the weight fragments.
typedef __attribute__((ext_vector_type(4))) short s16x4_g;

// STAT (N <= 16 * NCH, one sample): per-column sum / sum of squares of the rounded output accumulate in registers over all the tiles of
// a wave and are added once per workgroup to the replicated fp64 statistics buffer of the instance norm that consumes the output
// (proj + residual and fc2 + residual feed norm2 / the next block's norm1, the 1x1x1 shortcut conv feeds norm3): no separate pass.
// STAT == 2 / ANORM (round 5): see struct Epi - the norm-backward reduction in the data-gradient GEMM that produces the norm's output gradient,
// and the norm's apply pass folded into the operand load of the GEMM that consumes its output (Swin qkv / fc1: swin_transformer_block.py:103,176).
template <int K16, bool GELU, int NCH = 12 /* n-tiles per accumulator chunk: fewer for the long rows (K >= 288), whose A fragments fill the registers */,
          int STAT = 0, bool SCAT = false, bool ANORM = false>
__global__ void __launch_bounds__(256, 2) gemm_nt_stream_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ W, int64_t ldw,
                                                                bf16* __restrict__ C, int64_t ldc, int M, int Nfull, Epi epi, int N) {
  // blockIdx.y = column group of N columns (the mid-size token counts: 13,824 rows are 108 workgroups of 128 rows - with the whole weight
  // staged by each of them, 73 KB for fc1 at stage 1, the staging loop was most of the 33 us; column groups give 4x the workgroups, each
  // staging a quarter).  Everything below works on the group's slice: pointers advanced by n_off columns / weight rows.
  const int n_off = blockIdx.y * N;
  W += (int64_t)n_off * ldw;
  C += n_off;
  if (epi.bias) epi.bias += n_off;
  if (epi.res) epi.res = reinterpret_cast<const bf16*>(epi.res) + n_off;
  if (epi.aux) epi.aux = reinterpret_cast<bf16*>(epi.aux) + n_off;
  const float* bias = epi.bias;
  constexpr int K = K16 * 16, KS32 = K / 32, TAIL = K16 & 1;
  constexpr int ROWB = K * 2 + 16;   // weight row stride in LDS: conflict-free 16-byte fragment reads
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* lbias = reinterpret_cast<float*>(lds + (size_t)N * ROWB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, kg = lane >> 4;
  for (int c = tid; c < N * (K / 8); c += 256) {
    const int row = c / (K / 8), ch = c - row * (K / 8);
    *reinterpret_cast<bf16x8*>(lds + row * ROWB + ch * 16) = *reinterpret_cast<const bf16x8*>(W + (int64_t)row * ldw + ch * 8);
  }
  for (int n = tid; n < N; n += 256) lbias[n] = bias ? bias[n] : 0.f;
  float* lmean = lbias + N;          // STAT == 2: mean / rstd of the norm whose output gradient this GEMM produces, columns of this group
  float* lrstd = lmean + N;
  float* lsc = lbias + N + (STAT == 2 ? 2 * N : 0);      // ANORM: scale / shift per k
  float* lsh = lsc + K;
  if constexpr (STAT == 2) {
    for (int n = tid; n < N; n += 256) {
      double a, b;
      gemm_gather_stat(epi.bstat_in, Nfull, n_off + n, a, b);
      float m_, r_;
      gemm_mean_rstd(a, b, 1.0 / M, epi.beps, m_, r_);
      lmean[n] = m_;
      lrstd[n] = r_;
    }
  }
  if constexpr (ANORM) {
    // (a select chain: a run-time index into the by-value argument struct would send the whole struct through scratch memory)
    const int st = epi.an_styles ? epi.an_styles[0] : 0;
    const float* g = st == 0 ? epi.an_gamma[0] : st == 1 ? epi.an_gamma[1] : st == 2 ? epi.an_gamma[2] : epi.an_gamma[3];
    const float* be = st == 0 ? epi.an_beta[0] : st == 1 ? epi.an_beta[1] : st == 2 ? epi.an_beta[2] : epi.an_beta[3];
    for (int k = tid; k < K; k += 256) {
      double a, b;
      gemm_gather_stat(epi.an_stat, K, k, a, b);
      float m_, r_;
      gemm_mean_rstd(a, b, 1.0 / M, epi.an_eps, m_, r_);
      const float sc = r_ * (g ? g[k] : 1.f);
      lsc[k] = sc;
      lsh[k] = (be ? be[k] : 0.f) - m_ * sc;
    }
  }
  __syncthreads();
  const int ntiles = N / 16, mtiles = (M + 31) / 32, nwaves = gridDim.x * 4;
  bf16x8 cur[2][KS32 > 0 ? KS32 : 1], nxt[2][KS32 > 0 ? KS32 : 1];
  bf16x4 curt[2], nxtt[2];
  // ANORM: this lane's scale / shift (k = ks * 32 + 8 kg + e, tail k = KS32 * 32 + 4 kg + e) stay in registers
  float asc[ANORM ? (KS32 > 0 ? KS32 : 1) : 1][8], ash[ANORM ? (KS32 > 0 ? KS32 : 1) : 1][8], asct[4], asht[4];
  if constexpr (ANORM) {
#pragma unroll
    for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) { asc[ks][e] = lsc[ks * 32 + 8 * kg + e]; ash[ks][e] = lsh[ks * 32 + 8 * kg + e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { asct[e] = TAIL ? lsc[KS32 * 32 + 4 * kg + e] : 0.f; asht[e] = TAIL ? lsh[KS32 * 32 + 4 * kg + e] : 0.f; }
  }
  // normalise the fragments of one 32-row tile in place (and store them once: column group 0 of the grid)
  auto normA = [&](int tile_, bf16x8 (&f)[2][KS32 > 0 ? KS32 : 1], bf16x4 (&t)[2]) {
    if constexpr (ANORM) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = tile_ * 32 + mt * 16 + fi;
        bf16* orow = (epi.an_out && blockIdx.y == 0 && row < M) ? reinterpret_cast<bf16*>(epi.an_out) + (int64_t)row * epi.ld_an_out : nullptr;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
          bf16x8 v = f[mt][ks];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaf((float)v[e], asc[ks][e], ash[ks][e]);
          f[mt][ks] = v;
          if (orow) *reinterpret_cast<bf16x8*>(orow + ks * 32 + 8 * kg) = v;
        }
        if (TAIL) {
          bf16x4 v = t[mt];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (bf16)fmaf((float)v[e], asct[e], asht[e]);
          t[mt] = v;
          if (orow) *reinterpret_cast<bf16x4*>(orow + KS32 * 32 + 4 * kg) = v;
        }
      }
    }
  };
  auto loadA = [&](int tile, bf16x8 (&f)[2][KS32 > 0 ? KS32 : 1], bf16x4 (&t)[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = min(tile * 32 + mt * 16 + fi, M - 1);
      const bf16* p = A + (int64_t)row * lda;
#pragma unroll
      for (int ks = 0; ks < KS32; ++ks) f[mt][ks] = *reinterpret_cast<const bf16x8*>(p + ks * 32 + 8 * kg);
      if (TAIL) t[mt] = *reinterpret_cast<const bf16x4*>(p + KS32 * 32 + 4 * kg);
    }
  };
  float ssum[STAT ? NCH : 1][4], ssq[STAT ? NCH : 1][4];
#pragma unroll
  for (int j = 0; j < (STAT ? NCH : 1); ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[j][r] = 0.f; ssq[j][r] = 0.f; }
  int tile = blockIdx.x * 4 + wave;
  if (tile < mtiles) { loadA(tile, cur, curt); normA(tile, cur, curt); }
  for (; tile < mtiles; tile += nwaves) {
    if (tile + nwaves < mtiles) loadA(tile + nwaves, nxt, nxtt);
    for (int nc = 0; nc < ntiles; nc += NCH) {
      const int ncnt = min(NCH, ntiles - nc);
      f32x4 acc[2][NCH];
#pragma unroll
      for (int j = 0; j < NCH; ++j) acc[0][j] = acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        if (j < ncnt) {
          const char* wrow = lds + ((nc + j) * 16 + fi) * ROWB;
#pragma unroll
          for (int ks = 0; ks < KS32; ++ks) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + ks * 64 + kg * 16);
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, cur[0][ks], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, cur[1][ks], acc[1][j], 0, 0, 0);
          }
          if (TAIL) {
            const s16x4_g wt = *reinterpret_cast<const s16x4_g*>(wrow + KS32 * 64 + kg * 8);
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wt, __builtin_bit_cast(s16x4_g, curt[0]), acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wt, __builtin_bit_cast(s16x4_g, curt[1]), acc[1][j], 0, 0, 0);
          }
        }
      }
      // lane (fi = row, kg): columns (nc + j) * 16 + 4kg .. +3
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = tile * 32 + mt * 16 + fi;
        if (row < M) {
          bf16* crow = C + (int64_t)row * ldc;
          int64_t srow = 0;      // SCAT: row of voxel (2d, 2h, 2w) in the doubled grid
          if constexpr (SCAT) {
            const int w = row % epi.sw, t1 = row / epi.sw, h = t1 % epi.sh, t2 = t1 / epi.sh;      // t2 = b * sd + d
            srow = ((int64_t)t2 * 2 * (2 * epi.sh) + 2 * h) * (2 * epi.sw) + 2 * w;
          }
#pragma unroll
          for (int j = 0; j < NCH; ++j) {
            if (j < ncnt) {
              const int n = (nc + j) * 16 + 4 * kg;
              const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbias + n);
              const f32x4 v = epi_vec4_bf16(acc[mt][j] + b4, row, n, epi, GELU);
              const bf16x4 o4 = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
              if constexpr (SCAT) {
                const int jj = (n + n_off) / epi.sco, co = (n + n_off) - jj * epi.sco;
                const int64_t drow = srow + ((int64_t)(jj >> 2) * (2 * epi.sh) + ((jj >> 1) & 1)) * (2 * epi.sw) + (jj & 1);
                *reinterpret_cast<bf16x4*>(C - n_off + drow * ldc + co) = o4;
              } else {
                *reinterpret_cast<bf16x4*>(crow + n) = o4;
              }
              if constexpr (STAT == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float q = (float)o4[r]; ssum[j][r] += q; ssq[j][r] = fmaf(q, q, ssq[j][r]); }
              }
              if constexpr (STAT == 2) {      // the terms of instnorm_bwd_reduce_kernel: s += g, q = fma(g, (x - m) * rs, q)
                const bf16x4 x4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(epi.bx) + n_off + (int64_t)row * epi.ldbx + n);
                const f32x4 m4 = *reinterpret_cast<const f32x4*>(lmean + n), r4 = *reinterpret_cast<const f32x4*>(lrstd + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float q = (float)o4[r]; ssum[j][r] += q; ssq[j][r] = fmaf(q, ((float)x4[r] - m4[r]) * r4[r], ssq[j][r]); }
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int ks = 0; ks < KS32; ++ks) cur[mt][ks] = nxt[mt][ks];
      curt[mt] = nxtt[mt];
    }
    if (tile + nwaves < mtiles) normA(tile + nwaves, cur, curt);
  }
  if constexpr (STAT != 0) {
    // lane (fi = row, kg) holds columns j * 16 + 4 kg + r: every lane's partials go to LDS ([k][column][wave * 16 + fi], 65-float rows),
    // one thread per (column, k) adds the 64 of them; the weight image is dead once every wave is past its last tile
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
    constexpr int RS = 65;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (j < ntiles) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = j * 16 + kg * 4 + r;
          red[(0 * N + col) * RS + wave * 16 + fi] = ssum[j][r];
          red[(1 * N + col) * RS + wave * 16 + fi] = ssq[j][r];
        }
      }
    }
    __syncthreads();
    for (int o = tid; o < 2 * N; o += 256) {
      const int k = o / N, col = o - k * N;
      const float* rp = red + (k * N + col) * RS;
      float tot = 0.f;
#pragma unroll 16
      for (int i = 0; i < 64; ++i) tot += rp[i];
      atomicAdd(epi.stat + ((int64_t)(blockIdx.x & 15) * Nfull + n_off + col) * 2 + k, (double)tot);      // [16 replicas][B = 1][N][2]
    }
  }
}

// ------------------------------------------------------------------------------------------------ NT, small
// Linears of the deep stages (a few hundred to a few thousand tokens, K up to 3072): too little work per output tile to
// amortise an LDS pipeline, so the time is the dependent global->LDS->MFMA chain of each K stage.  Here a workgroup owns
// one 16 x (16*NTW) output tile, its four waves split K, and every lane's operands (16 contiguous bytes of a row each) go
// straight from global memory (L2) to registers in batches of four k-steps, all loads of a batch in flight together;
// the four partial tiles meet in LDS.
// Workgroup -> tile: the hardware deals consecutive workgroups to the 8 XCDs round-robin, each with its own L2.  With (m-tile, n-tile) =
// (blockIdx.x, blockIdx.y) the 14 row tiles that share a weight tile ran on 8 different XCDs and every XCD fetched the WHOLE weight
// matrix from HBM: 216 tokens x (768 .. 3072)^2 took 3.5 us + 3 us per MB of weights (0.33 TB/s of weight bytes).  XCD k owns the k-th
// contiguous eighth of the n-tiles instead (all row tiles of an n-tile side by side): the weights cross HBM once, the few activations
// (216 x K) are what every XCD reads.
// MTW x NTW MFMA tiles per workgroup.  The kernel is bound by the operand traffic L2 -> CU, not by a latency chain (a software-pipelined
// K loop with unconditional, clamped loads was SLOWER: 216 x 768 x 3072 18.8 -> 20.3 us, it over-fetches two batches per wave): 32 x 32
// outputs read 64 operand rows per k-step for 4 MFMAs where 16 x 32 read 48 for 2 (216 x 768 x 3072: 20.2 -> 14.2 us, 216 x 768 x 768:
// 8.4 -> 6.4), 32 x 64 read 96 for 8 (216 x 2304 x 768: 12.8 -> 8.9).  scripts/bench_gemm.py small, MISEG_GEMM_SMALL_TILE.
// Round 5 (the deep Swin stages, 1,728 / 216 tokens of ONE sample): STAT = 1 - instance-norm statistics of the rounded output in the epilogue
// (proj + residual, fc2 + residual: the consumer norm then has no statistics pass); STAT = 2 - the backward sums (sum q, sum q * xhat) of
// the norm whose output gradient this GEMM produces; ANORM - A is the raw input of an instance norm, normalised as it is loaded (scale /
// shift per k from LDS), norm(A) stored once by the workgroups of the first column tile.  With these the norm between two linears of a deep
// Swin block is no launch at all in the forward pass (it was one 5 - 8 us launch each: ~6 % of the stage's chain).
static constexpr int SM_BS = 4;      // k-steps whose loads are in flight together
template <int MTW, int NTW, bool GELU, int STAT = 0, bool ANORM = false>
__global__ void __launch_bounds__(256) gemm_nt_small_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ W, int64_t ldw, bf16* __restrict__ C,
                                                            int64_t ldc, int M, int N, int K, Epi epi, int gm) {
  const float* bias = epi.bias;
  constexpr int NTILE = MTW * NTW;
  __shared__ __attribute__((aligned(16))) float part[4][NTILE][64][4];
  __shared__ __attribute__((aligned(16))) float ncoef[(ANORM || STAT == 2) ? 2 * 384 : 4];      // ANORM: scale[K], shift[K] (K <= 384); STAT 2: mean / rstd of this tile's columns
  // wave index as a SCALAR: the k-range guards below must be real branches -- an MFMA ignores EXEC, so a guard the compiler
  // if-converts (it cannot know tid >> 6 is wave-uniform) would let the skipped k-steps accumulate garbage operands
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, kg = lane >> 4;
  int m0, n0;
  {
    const int nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, r = nb & 7;
    const int unit = xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3);
    const int nt_ = unit / gm;
    m0 = (unit - nt_ * gm) * 16 * MTW;
    n0 = nt_ * 16 * NTW;
  }
  const int ksteps = K / 32, kpw = (ksteps + 3) / 4;
  const int ks0 = wave * kpw, ks1 = min(ksteps, ks0 + kpw);
  if constexpr (ANORM) {
    const int st = epi.an_styles ? epi.an_styles[0] : 0;
    const float* g = st == 0 ? epi.an_gamma[0] : st == 1 ? epi.an_gamma[1] : st == 2 ? epi.an_gamma[2] : epi.an_gamma[3];
    const float* be = st == 0 ? epi.an_beta[0] : st == 1 ? epi.an_beta[1] : st == 2 ? epi.an_beta[2] : epi.an_beta[3];
    for (int k = tid; k < K; k += 256) {
      double a, b;
      gemm_gather_stat(epi.an_stat, K, k, a, b);
      float m_, r_;
      gemm_mean_rstd(a, b, 1.0 / M, epi.an_eps, m_, r_);
      const float sc = r_ * (g ? g[k] : 1.f);
      ncoef[k] = sc;
      ncoef[384 + k] = (be ? be[k] : 0.f) - m_ * sc;
    }
    __syncthreads();
  }
  if constexpr (STAT == 2) {
    for (int c = tid; c < 16 * NTW; c += 256) {
      float m_ = 0.f, r_ = 0.f;
      if (n0 + c < N) {
        double a, b;
        gemm_gather_stat(epi.bstat_in, N, n0 + c, a, b);
        gemm_mean_rstd(a, b, 1.0 / M, epi.beps, m_, r_);
      }
      ncoef[c] = m_;
      ncoef[384 + c] = r_;
    }
    // (visible to the epilogue's readers through the barrier behind the partial-tile stores)
  }
  const bf16* arow[MTW];
  const bf16* wrow[NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) arow[mt] = A + (int64_t)min(m0 + mt * 16 + fi, M - 1) * lda + 8 * kg;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) wrow[nt] = W + (int64_t)min(n0 + nt * 16 + fi, N - 1) * ldw + 8 * kg;
  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ks = ks0; ks < ks1; ks += SM_BS) {
    bf16x8 af[SM_BS][MTW], wf[SM_BS][NTW];
#pragma unroll
    for (int u = 0; u < SM_BS; ++u) {
      if (ks + u < ks1) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) af[u][mt] = *reinterpret_cast<const bf16x8*>(arow[mt] + (ks + u) * 32);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) wf[u][nt] = *reinterpret_cast<const bf16x8*>(wrow[nt] + (ks + u) * 32);
      }
    }
#pragma unroll
    for (int u = 0; u < SM_BS; ++u) {
      if (ks + u < ks1) {
        if constexpr (ANORM) {      // fma(x, sc, sh) rounded to bf16: the bits instnorm_apply_kernel would have stored
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(ncoef + (ks + u) * 32 + 8 * kg), s1 = *reinterpret_cast<const f32x4*>(ncoef + (ks + u) * 32 + 8 * kg + 4);
          const f32x4 h0 = *reinterpret_cast<const f32x4*>(ncoef + 384 + (ks + u) * 32 + 8 * kg), h1 = *reinterpret_cast<const f32x4*>(ncoef + 384 + (ks + u) * 32 + 8 * kg + 4);
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            bf16x8 v = af[u][mt];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (bf16)fmaf((float)v[e], s0[e], h0[e]); v[4 + e] = (bf16)fmaf((float)v[4 + e], s1[e], h1[e]); }
            af[u][mt] = v;
            const int row = m0 + mt * 16 + fi;
            if (epi.an_out && n0 == 0 && row < M) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(epi.an_out) + (int64_t)row * epi.ld_an_out + (ks + u) * 32 + 8 * kg) = v;
          }
        }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][nt], af[u][mt], acc[mt][nt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) *reinterpret_cast<f32x4*>(&part[wave][mt * NTW + nt][lane][0]) = acc[mt][nt];
  __syncthreads();
  // wave w finishes tiles w, w + 4, ...: lane (fi = row, kg): columns n0 + nt * 16 + 4kg .. +3
  for (int t = wave; t < NTILE; t += 4) {
    const int mt = t / NTW, nt = t - mt * NTW;
    f32x4 v = *reinterpret_cast<const f32x4*>(&part[0][t][lane][0]);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(&part[w][t][lane][0]);
    const int m = m0 + mt * 16 + fi, n = n0 + nt * 16 + 4 * kg;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < M && n < N) {
      if (bias) { v[0] += bias[n]; v[1] += bias[n + 1]; v[2] += bias[n + 2]; v[3] += bias[n + 3]; }
      v = epi_vec4_bf16(v, m, n, epi, GELU);
      const bf16x4 o4 = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      *reinterpret_cast<bf16x4*>(C + (int64_t)m * ldc + n) = o4;
      if constexpr (STAT == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s4[r] = (float)o4[r]; q4[r] = s4[r] * s4[r]; }
      }
      if constexpr (STAT == 2) {
        const bf16x4 x4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(epi.bx) + (int64_t)m * epi.ldbx + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s4[r] = (float)o4[r]; q4[r] = s4[r] * (((float)x4[r] - ncoef[nt * 16 + 4 * kg + r]) * ncoef[384 + nt * 16 + 4 * kg + r]); }
      }
    }
    if constexpr (STAT != 0) {
      // the 16 rows of the tile sit on the 16 lanes fi of each kg group: butterfly over fi, then one fp64 atomic per (column, which) from lane fi == 0
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s4[r] += __shfl_xor(s4[r], o, 64); q4[r] += __shfl_xor(q4[r], o, 64); }
      }
      if (fi == 0 && n < N) {
        double* dst = epi.stat + ((int64_t)(((m0 >> 4) + mt) & 15) * N + n) * 2;      // replica = row tile mod 16: [16][B = 1][N][2]
#pragma unroll
        for (int r = 0; r < 4; ++r) { atomicAdd(dst + 2 * r, (double)s4[r]); atomicAdd(dst + 2 * r + 1, (double)q4[r]); }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ TN
// A stored [K][M] (lda), B stored [K][N] (ldb).  Tile 64 x 64, 4 waves as 2(M) x 2(N), each 32 x 32.
template <class T> struct TnFrag;
template <> struct TnFrag<bf16> {
  static constexpr int BK = 32;  // k rows per stage
  // lane's 8 k-values of column (c0 + lane&15) for MFMA k-group lane>>4, via two transposed LDS reads
  __device__ static __forceinline__ bf16x8 load(const char* tile, int rowb, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3;
    const char* a1 = tile + (8 * g + qq) * rowb + (c0 + 4 * p) * 2;
    const char* a2 = a1 + 4 * rowb;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a2));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
};

// regroup > 0 (fp32 output): column n = j * regroup + c of the product is stored at column c * (N / regroup) + j - the weight gradient of a
// ConvTranspose3d(k2, s2) computed as x^T dy8 with dy8's columns in (j, co) order lands in the torch layout [Cin][Cout][2][2][2] directly
// (round 5: no [(j, co)][ci] intermediate, no permute launch, no fill)
// BKB (bf16; 0 = 128): k rows per stage.  The GROUPED launch runs beside the grouped conv weight gradients, whose one workgroup per CU holds
// 96 KB of LDS: with 128-row stages (36 KB) ONE of these workgroups fits beside it, with 96-row stages (27 KB) two (round 5)
template <class T, class TO, int BKB = 0>
__device__ __forceinline__ void gemm_tn_body(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, TO* __restrict__ C, int64_t ldc,
                                             int M, int N, int K, int mode, bool vec_a, bool vec_b, int k_per_split, int bx, int by, int bz, int regroup = 0) {
  typedef typename Vec16<T>::type VT;
  constexpr int KPC = Mma<T>::KPC;     // elements per 16-byte chunk (here along m / n)
  constexpr int BM = 64, BN = 64, BK = std::is_same<T, bf16>::value ? (BKB ? BKB : 128) : 64;   // k rows per stage
  constexpr int ROWB = BM * (int)sizeof(T) + 16;   // bytes per k-row of a tile (padded)
  __shared__ __attribute__((aligned(16))) char lds[2 * BK * ROWB];
  char* lA = lds;
  char* lB = lds + BK * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = bx * BM, n0 = by * BN;
  const int kbeg = bz * k_per_split, kend = min(K, kbeg + k_per_split);
  f32x4 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int CPR = BM / KPC;                 // chunks per k-row
  constexpr int ITERS = (BK * CPR + 255) / 256;
  VT ra[ITERS], rb[ITERS];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int c = tid + i * 256, kr = c / CPR, ch = c % CPR;
      if (c < BK * CPR) {
        // rows are k here, columns m / n: reuse load_chunk with (row=k, col=m)
        ra[i] = load_chunk<T>(A, lda, k0 + kr, kend, m0 + ch * KPC, M, vec_a);
        rb[i] = load_chunk<T>(B, ldb, k0 + kr, kend, n0 + ch * KPC, N, vec_b);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int c = tid + i * 256, kr = c / CPR, ch = c % CPR;
      if (c < BK * CPR) {
        *reinterpret_cast<VT*>(lA + kr * ROWB + ch * 16) = ra[i];
        *reinterpret_cast<VT*>(lB + kr * ROWB + ch * 16) = rb[i];
      }
    }
  };
  const int fi = lane & 15, fq = lane >> 4;
  gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);
    if constexpr (std::is_same<T, bf16>::value) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          af[t] = TnFrag<bf16>::load(lA + ks * 32 * ROWB, ROWB, wm * 32 + t * 16, lane);
          bfr[t] = TnFrag<bf16>::load(lB + ks * 32 * ROWB, ROWB, wn * 32 + t * 16, lane);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float af[2], bfr[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          af[t] = *reinterpret_cast<const float*>(lA + (kk * 4 + fq) * ROWB + (wm * 32 + t * 16 + fi) * 4);
          bfr[t] = *reinterpret_cast<const float*>(lB + (kk * 4 + fq) * ROWB + (wn * 32 + t * 16 + fi) * 4);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
      }
      // fp32 parity mode: blocked accumulation as in gemm_nt_kernel (one stage of reduction rows joins the running sum once)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { tot[mt][nt] += acc[mt][nt]; acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
  }
  if constexpr (std::is_same<TO, float>::value) {
    if (regroup > 0) {
      const int per = N / regroup;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 v = std::is_same<T, float>::value ? tot[mt][nt] : acc[mt][nt];
          const int m = m0 + wm * 32 + mt * 16 + fi;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * 32 + nt * 16 + fq * 4 + r;
            if (m < M && n < N) {
              float* p = C + (int64_t)m * ldc + (n % regroup) * per + n / regroup;
              if (mode == 2) atomicAdd(p, v[r]);
              else if (mode == 1) *p += v[r];
              else *p = v[r];
            }
          }
        }
      return;
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      store_out4<TO>(C, ldc, m0 + wm * 32 + mt * 16 + fi, n0 + wn * 32 + nt * 16 + fq * 4, M, N, std::is_same<T, float>::value ? tot[mt][nt] : acc[mt][nt],
                     Epi{nullptr, MISEG_ACT_NONE, nullptr, 0, nullptr, 0, 0, nullptr}, mode);
}

template <class T, class TO>
__global__ void __launch_bounds__(256) gemm_tn_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, TO* __restrict__ C, int64_t ldc,
                                                      int M, int N, int K, int mode, bool vec_a, bool vec_b, int k_per_split) {
  gemm_tn_body<T, TO>(A, lda, B, ldb, C, ldc, M, N, K, mode, vec_a, vec_b, k_per_split, blockIdx.x, blockIdx.y, blockIdx.z);
}

static constexpr int TN_GROUP_BK = 96;      // bf16 stage depth of the grouped launch (see gemm_tn_body)
// grouped form: a list of independent TN problems (the weight gradients of the deep stages, a few dozen workgroups each)
// in ONE launch; descriptors travel in the kernel arguments
struct TnGroup {
  struct P { const void* A; int64_t lda; const void* B; int64_t ldb; float* C; int64_t ldc; int M, N, K, kps, gx, gy, block0, mode, regroup; bool va, vb; } p[MISEG_GEMM_GROUP];
  int n;
};

template <class T>
__global__ void __launch_bounds__(256) gemm_tn_group_kernel(TnGroup g) {
  int k = 0;
  while (k + 1 < g.n && g.p[k + 1].block0 <= (int)blockIdx.x) ++k;
  const TnGroup::P& q = g.p[k];
  const int local = blockIdx.x - q.block0;
  const int bx = local % q.gx, by = (local / q.gx) % q.gy, bz = local / (q.gx * q.gy);
  gemm_tn_body<T, float, TN_GROUP_BK>((const T*)q.A, q.lda, (const T*)q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, q.mode, q.va, q.vb, q.kps, bx, by, bz, q.regroup);
}

// ------------------------------------------------------------------------------------------------ TN, streaming
// Weight gradients of the tall layers: C[M][N] (+)= sum over ~1e5..1e6 tokens of A[t][m] B[t][n] with M, N multiples of 48.
// HBM-bound on the two token streams.  A workgroup owns a (wm x wn) arrangement of 48x48 wave tiles and a contiguous
// token range; 64-token stages are double-buffered in LDS (one barrier per stage, the next stage's 16-byte chunks in
// registers while the current one feeds the MFMAs through transposed LDS reads), few token splits so that the fp32
// atomics of the epilogue stay a small fraction of the stream.
__global__ void __launch_bounds__(256, 2) gemm_tn_stream_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb,
                                                                float* __restrict__ C, int64_t ldc, int M, int N, int T, int wm, int wn, int tps, int mode,
                                                                float* __restrict__ partial, float* __restrict__ colsum) {
  constexpr int BK = 64, MAXC = 8;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int BM = wm * 48, BN = wn * 48, RA = BM * 2 + 16, RB = BN * 2 + 16;
  const int stage_bytes = BK * (RA + RB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int t0 = blockIdx.z * tps, t1 = min(T, t0 + tps);
  const int wmi = wave % wm, wni = wave / wm;
  const bool active = (m0 + wmi * 48 < M) && (n0 + wni * 48 < N);
  // per-thread copy slots (fixed across stages)
  const int cpa = BM / 8, cpr = (BM + BN) / 8, total = BK * cpr;
  int64_t goff[MAXC];      // element offset from the stage's first token row; -1: always zero
  int loff[MAXC], lrow[MAXC];
  bool isb[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    goff[i] = -1; loff[i] = -1; lrow[i] = 0; isb[i] = false;
    if (c < total) {
      const int row = c / cpr, ch = c - row * cpr;
      lrow[i] = row;
      if (ch < cpa) {
        const int col = m0 + ch * 8;
        loff[i] = row * RA + ch * 16;
        if (col < M) goff[i] = (int64_t)row * lda + col;
      } else {
        const int col = n0 + (ch - cpa) * 8;
        isb[i] = true;
        loff[i] = BK * RA + row * RB + (ch - cpa) * 16;
        if (col < N) goff[i] = (int64_t)row * ldb + col;
      }
    }
  }
  bf16x8 r[MAXC];
  auto gload = [&](int tk) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
      if (goff[i] >= 0 && tk + lrow[i] < t1) v = *reinterpret_cast<const bf16x8*>((isb[i] ? B + (int64_t)tk * ldb : A + (int64_t)tk * lda) + goff[i]);
      r[i] = v;
    }
  };
  auto lstore = [&](char* b) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (loff[i] >= 0) *reinterpret_cast<bf16x8*>(b + loff[i]) = r[i];
  };
  f32x4 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Round 5: the column sums of A ride along (dW = dy^T x is called with A = dy: they are the bias gradient of the linear layer - until now
  // a separate pass over dy, 240 MB per step at the end of the backward pass - and the A fragments already in registers give them as one
  // more MFMA block against a fragment of ones).  One wave per row block (workgroup column 0 only); fp32 atomics into the bias-gradient slot.
  const bool csum = colsum != nullptr && blockIdx.y == 0 && wni == 0 && active;
  f32x4 cs[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.f;
  gload(t0);
  lstore(lds);
  __syncthreads();
  int s = 0;
  for (int tk = t0; tk < t1; tk += BK, s ^= 1) {
    const bool more = tk + BK < t1;
    if (more) gload(tk + BK);
    if (active) {
      const char* bA = lds + s * stage_bytes;
      const char* bB = bA + BK * RA;
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 af[3], bfr[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          af[t] = TnFrag<bf16>::load(bA + ks * 32 * RA, RA, wmi * 48 + t * 16, lane);
          bfr[t] = TnFrag<bf16>::load(bB + ks * 32 * RB, RB, wni * 48 + t * 16, lane);
        }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
        if (csum) {      // (wave-uniform)
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) cs[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[mt], cs[mt], 0, 0, 0);
        }
      }
    }
    if (more) lstore(lds + (s ^ 1) * stage_bytes);   // last read one iteration ago: every wave is past that barrier
    __syncthreads();
  }
  if (csum && lane < 16) {      // every row of the ones block holds the same sums: row 0 (lanes 0..15, register 0) carries columns m = fi
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const int m = m0 + wmi * 48 + mt * 16 + lane;
      if (m < M) atomicAdd(colsum + m, cs[mt][0]);
    }
  }
  if (active) {
    const int fi = lane & 15, fq = lane >> 4;
    if (partial) {   // partial[split][M][N]: 16-byte stores, summed by gemm_tn_partial_reduce_kernel
      float* pp = partial + (int64_t)blockIdx.z * M * N;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
          *reinterpret_cast<f32x4*>(pp + (int64_t)(m0 + wmi * 48 + mt * 16 + fi) * N + n0 + wni * 48 + nt * 16 + fq * 4) = acc[mt][nt];
    } else {
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
          store_out4<float>(C, ldc, m0 + wmi * 48 + mt * 16 + fi, n0 + wni * 48 + nt * 16 + fq * 4, M, N, acc[mt][nt], Epi{nullptr, MISEG_ACT_NONE, nullptr, 0, nullptr, 0, 0, nullptr}, mode);
    }
  }
}

// C[m][n] (+)= sum over splits of partial[s][m][n]; one thread per 4 consecutive n and per group of TN_RG splits
// (blockIdx.y); with more than one group the groups meet in fp32 atomics on a zero-filled / accumulate-mode C.
static constexpr int TN_RG = 16;
__global__ void __launch_bounds__(256) gemm_tn_partial_reduce_kernel(const float* __restrict__ partial, float* __restrict__ C, int64_t ldc, int M, int N, int splits,
                                                                     int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // over M * N / 4
  if (i >= M * (N / 4)) return;
  const int m = i / (N / 4), n = (i - m * (N / 4)) * 4;
  const int s0 = blockIdx.y * TN_RG, s1 = min(splits, s0 + TN_RG);
  const float* p = partial + (int64_t)m * N + n;
  const int64_t stride = (int64_t)M * N;
  f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  int s = s0;
  for (; s + 3 < s1; s += 4) {
    a0 += *reinterpret_cast<const f32x4*>(p + (s + 0) * stride);
    a1 += *reinterpret_cast<const f32x4*>(p + (s + 1) * stride);
    a2 += *reinterpret_cast<const f32x4*>(p + (s + 2) * stride);
    a3 += *reinterpret_cast<const f32x4*>(p + (s + 3) * stride);
  }
  for (; s < s1; ++s) a0 += *reinterpret_cast<const f32x4*>(p + s * stride);
  const f32x4 v = (a0 + a1) + (a2 + a3);
  float* c = C + (int64_t)m * ldc + n;
  if (gridDim.y > 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(c + r, v[r]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = accumulate ? c[r] + v[r] : v[r];
  }
}

// dst[i0][i1][i2] (+)= src[i0*s0 + i1*s1 + i2*s2]   (fp32; weight-gradient re-layout)
__global__ void __launch_bounds__(256) permute3_kernel(const float* __restrict__ src, float* __restrict__ dst, int n0, int n1, int n2, int64_t s0, int64_t s1,
                                                       int64_t s2, int accumulate) {
  const int64_t total = (int64_t)n0 * n1 * n2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int i2 = (int)(i % n2);
    const int64_t t = i / n2;
    const int i1 = (int)(t % n1), i0 = (int)(t / n1);
    const float v = src[i0 * s0 + i1 * s1 + i2 * s2];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

}  // namespace miseg

using namespace miseg;

struct TnStreamPlan { int wm, wn, gx, gy, splits, tps; };

// streaming TN path: bf16 operands, fp32 C, channel counts in multiples of 48, >= 2048 tokens, library-chosen split
static bool tn_stream_plan(const miseg_gemm_params* p, TnStreamPlan* pl) {
  if (!(p->ta == 1 && p->tb == 1 && p->dtype == MISEG_BF16 && p->out_dtype == MISEG_F32 && p->split_k == 0)) return false;
  if (!(p->M % 48 == 0 && p->N % 48 == 0 && p->K >= 2048)) return false;
  if (((uintptr_t)p->A % 16) || ((uintptr_t)p->B % 16) || (p->lda % 8) || (p->ldb % 8)) return false;
  if (p->N <= 48) { pl->wm = 4; pl->wn = 1; } else if (p->M <= 48) { pl->wm = 1; pl->wn = 4; } else { pl->wm = 2; pl->wn = 2; }
  pl->gx = cdiv(p->M, pl->wm * 48);
  pl->gy = cdiv(p->N, pl->wn * 48);
  int splits = cdiv(384, pl->gx * pl->gy);      // (round 5: 512 -> 384 - fewer partial tiles for the batched sum at the end of the pass: +0.5 % on the step, 256 the same, 128 -1 %)
  const int max_splits = cdiv(p->K, 4 * 64);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  pl->tps = cdiv(cdiv(p->K, splits), 64) * 64;
  pl->splits = cdiv(p->K, pl->tps);
  return true;
}

// the small-M kernel (M <= 2048 rows of one sample) takes the same folds (gemm_nt_small_kernel)
static bool nt_small_ok(const miseg_gemm_params* p) {
  if (!p || p->ta || p->tb || p->dtype != MISEG_BF16 || p->out_dtype != MISEG_BF16) return false;
  const bool al_a = ((uintptr_t)p->A % 16 == 0) && (p->lda % 8 == 0), al_b = ((uintptr_t)p->B % 16 == 0) && (p->ldb % 8 == 0);
  const bool epi_vec_ok = (!p->res || (((uintptr_t)p->res % 8 == 0) && p->ldres % 4 == 0)) && (!p->epi_mode || (((uintptr_t)p->aux % 8 == 0) && p->ldaux % 4 == 0));
  return p->split_k <= 1 && !p->accumulate && p->M <= 2048 && p->M >= 16 && p->K % 32 == 0 && p->N % 16 == 0 && al_a && al_b && ((uintptr_t)p->C % 8 == 0) &&
         p->ldc % 4 == 0 && epi_vec_ok && !p->scat_cout;
}
extern "C" int miseg_gemm_fuses_stat(const miseg_gemm_params* p) {
  if (p && !p->ta && !p->tb && p->dtype == p->out_dtype && p->K == 1) {      // the rank-1 kernel (stem shortcut), either dtype
    const int n16 = p->dtype == MISEG_BF16 ? 8 : 4;
    return p->split_k <= 1 && !p->accumulate && p->act == MISEG_ACT_NONE && !p->res && !p->epi_mode && p->N % n16 == 0 && p->N <= 128 &&
           ((uintptr_t)p->C % 16 == 0) && p->ldc % n16 == 0 && !p->scat_cout;
  }
  if (!p || p->ta || p->tb || p->dtype != MISEG_BF16 || p->out_dtype != MISEG_BF16) return 0;
  if (nt_small_ok(p) && p->act == MISEG_ACT_NONE && !p->an.stat) return 1;      // the small-M kernel: any N (round 5)
  const size_t lds = (size_t)p->N * (p->K * 2 + 16) + (size_t)p->N * 4;
  const bool al_a = ((uintptr_t)p->A % 16 == 0) && (p->lda % 8 == 0), al_b = ((uintptr_t)p->B % 16 == 0) && (p->ldb % 8 == 0);
  const bool epi_vec_ok = (!p->res || (((uintptr_t)p->res % 8 == 0) && p->ldres % 4 == 0)) && (!p->epi_mode || (((uintptr_t)p->aux % 8 == 0) && p->ldaux % 4 == 0));
  return p->split_k <= 1 && !p->accumulate && p->act == MISEG_ACT_NONE && (p->K == 48 || p->K == 96 || p->K == 192) && p->N % 16 == 0 && p->N <= 96 &&
         p->M >= 4096 && lds <= 96 * 1024 && al_a && al_b && ((uintptr_t)p->C % 8 == 0) && p->ldc % 4 == 0 && epi_vec_ok;
}

// the streaming NT kernel with the consumer-side norm fold (ANORM): Swin qkv / fc1 at the high-resolution stages
static bool nt_stream_common_ok(const miseg_gemm_params* p) {
  if (!p || p->ta || p->tb || p->dtype != MISEG_BF16 || p->out_dtype != MISEG_BF16) return false;
  const bool al_a = ((uintptr_t)p->A % 16 == 0) && (p->lda % 8 == 0), al_b = ((uintptr_t)p->B % 16 == 0) && (p->ldb % 8 == 0);
  const bool epi_vec_ok = (!p->res || (((uintptr_t)p->res % 8 == 0) && p->ldres % 4 == 0)) && (!p->epi_mode || (((uintptr_t)p->aux % 8 == 0) && p->ldaux % 4 == 0));
  return p->split_k <= 1 && !p->accumulate && p->N % 16 == 0 && p->M >= 4096 && al_a && al_b && ((uintptr_t)p->C % 8 == 0) && p->ldc % 4 == 0 && epi_vec_ok &&
         !p->scat_cout;
}
extern "C" int miseg_gemm_fuses_anorm(const miseg_gemm_params* p) {
  if (nt_small_ok(p) && !p->stat && p->K <= 384 && p->an.num_styles >= 1 && p->an.num_styles <= MISEG_MAX_STYLES &&
      (!p->an_out || (((uintptr_t)p->an_out % 16) == 0 && p->ld_an_out % 8 == 0)))
    return 1;
  if (!nt_stream_common_ok(p) || !(p->K == 48 || p->K == 96) || p->stat) return 0;
  if (p->an.num_styles < 1 || p->an.num_styles > MISEG_MAX_STYLES) return 0;
  if (p->an_out && (((uintptr_t)p->an_out % 16) != 0 || p->ld_an_out % 8 != 0)) return 0;
  const size_t lds = (size_t)p->N * (p->K * 2 + 16) + (size_t)p->N * 4;
  return lds <= 96 * 1024;
}
// ... with the norm-backward sums of its output (STAT == 2): the data-gradient GEMMs behind qkv / fc1
extern "C" int miseg_gemm_fuses_bstat(const miseg_gemm_params* p) {
  if (nt_small_ok(p) && p->act == MISEG_ACT_NONE && !p->epi_mode && !p->an.stat && p->bs_x && ((uintptr_t)p->bs_x % 8) == 0 && p->ld_bs_x % 4 == 0 && p->N <= 384) return 1;
  if (!nt_stream_common_ok(p) || p->act != MISEG_ACT_NONE || p->epi_mode || p->an.stat) return 0;
  // (K = 384, the fc1 data gradient of the 96-channel stage: its operand fragments fill the register file - 156 bytes of scratch per lane with
  // the sums beside them - and it keeps the reduction launch)
  if (!(p->K == 144 || p->K == 288 || p->K == 48 || p->K == 96 || p->K == 192) || !(p->N == 48 || p->N == 96)) return 0;
  if (!p->bs_x || ((uintptr_t)p->bs_x % 8) != 0 || p->ld_bs_x % 4 != 0) return 0;
  const size_t lds = (size_t)p->N * (p->K * 2 + 16) + (size_t)p->N * 4;
  return lds <= 96 * 1024;
}

extern "C" int miseg_rank1_stats(const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int dtype, void* stat, miseg_stream_t stream_) {
  hipStream_t s = (hipStream_t)stream_;
  MISEG_REQUIRE(x && w && stat && M > 0 && N > 0 && N <= 128, MISEG_E_BADARG, "rank1_stats: bad args");
  return dispatch_dtype(dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    constexpr int N16 = Vec16<T>::N;
    MISEG_REQUIRE(N % N16 == 0, MISEG_E_UNSUPPORTED, "rank1_stats: N %d is not a multiple of %d", N, N16);
    int sb = cdiv(M, 256 / (N / N16));
    if (sb > 2048) sb = 2048;
    gemm_nt_k1_stat_kernel<T><<<sb, 256, 0, s>>>((const T*)x, ldx, (const T*)w, ldw, (T*)nullptr, 0, M, N, nullptr, (double*)stat);
    MISEG_LAUNCH_CHECK("rank1_stats");
    return MISEG_OK;
  });
}

extern "C" int miseg_gemm_fuses_scatter(const miseg_gemm_params* p) {
  if (!p || p->ta || p->tb || p->dtype != MISEG_BF16 || p->out_dtype != MISEG_BF16 || p->scat_cout <= 0) return 0;
  const size_t lds = (size_t)p->N * (p->K * 2 + 16) + (size_t)p->N * 4;
  const bool al_a = ((uintptr_t)p->A % 16 == 0) && (p->lda % 8 == 0), al_b = ((uintptr_t)p->B % 16 == 0) && (p->ldb % 8 == 0);
  return p->split_k <= 1 && !p->accumulate && p->act == MISEG_ACT_NONE && !p->res && !p->epi_mode && !p->stat && (p->K == 48 || p->K == 96) &&
         p->N == 8 * p->scat_cout && p->scat_cout % 4 == 0 && p->N % 16 == 0 && p->scat_d > 0 && p->scat_h > 0 && p->scat_w > 0 &&
         p->M % (p->scat_d * p->scat_h * p->scat_w) == 0 && p->M >= 4096 && lds <= 96 * 1024 && al_a && al_b && ((uintptr_t)p->C % 8 == 0) && p->ldc % 4 == 0;
}

extern "C" size_t miseg_gemm_workspace_bytes(const miseg_gemm_params* p) {
  TnStreamPlan pl;
  if (!p || !tn_stream_plan(p, &pl) || pl.splits <= 1) return 0;
  return (size_t)pl.splits * p->M * p->N * sizeof(float);
}

extern "C" int miseg_gemm_fuses_stat(const miseg_gemm_params* p);
extern "C" int miseg_gemm_fuses_scatter(const miseg_gemm_params* p);
extern "C" int miseg_gemm_fuses_anorm(const miseg_gemm_params* p);
extern "C" int miseg_gemm_fuses_bstat(const miseg_gemm_params* p);

// column groups of the streaming NT kernel: the smallest divisor d of the N / 16 column tiles that brings the grid to >= 300 workgroups
static int nt_stream_groups(int blocks, int N) {
  const int tiles = N / 16;
  for (int d = 1; d <= tiles; ++d)
    if (tiles % d == 0 && blocks * d >= 300) return d;
  return tiles;
}

template <class T, class TO>
static int launch_gemm(const miseg_gemm_params* p, hipStream_t s) {
  constexpr int N16 = Vec16<T>::N;
  const bool al_a = ((uintptr_t)p->A % 16 == 0) && (p->lda % N16 == 0);
  const bool al_b = ((uintptr_t)p->B % 16 == 0) && (p->ldb % N16 == 0);
  int split = p->split_k > 1 ? p->split_k : 1;
  const bool f32out = std::is_same<TO, float>::value;
  if (split > 1 && !f32out) return set_error(MISEG_E_BADARG, "gemm: split_k needs fp32 output");
  if (p->accumulate && !f32out) return set_error(MISEG_E_BADARG, "gemm: accumulate needs fp32 output");
  int mode = split > 1 ? 2 : (p->accumulate ? 1 : 0);
  Epi epi{p->bias, p->act, p->res, p->ldres, p->aux, p->ldaux, p->epi_mode, (double*)p->stat, p->scat_d, p->scat_h, p->scat_w, p->scat_cout};
  epi.bx = p->bs_x; epi.ldbx = p->ld_bs_x; epi.bstat_in = (const double*)p->bs_stat; epi.beps = p->bs_eps;
  epi.an_stat = (const double*)p->an.stat; epi.an_styles = p->an.styles; epi.an_eps = p->an.eps; epi.an_out = p->an_out; epi.ld_an_out = p->ld_an_out;
  for (int i = 0; i < MISEG_MAX_STYLES; ++i) { epi.an_gamma[i] = i < p->an.num_styles ? p->an.gamma[i] : nullptr; epi.an_beta[i] = i < p->an.num_styles ? p->an.beta[i] : nullptr; }
  const bool bstat = p->stat && p->stat_mode == 2;
  if (p->an.stat && !miseg_gemm_fuses_anorm(p)) return set_error(MISEG_E_UNSUPPORTED, "gemm: folded instance norm on this shape / path (ask miseg_gemm_fuses_anorm first)");
  if (bstat && !miseg_gemm_fuses_bstat(p)) return set_error(MISEG_E_UNSUPPORTED, "gemm: norm-backward sums on this shape / path (ask miseg_gemm_fuses_bstat first)");
  if (p->stat && p->stat_mode != 0 && p->stat_mode != 2) return set_error(MISEG_E_BADARG, "gemm: stat_mode %d", p->stat_mode);
  const bool epi_vec_ok = (!p->res || (((uintptr_t)p->res % 8 == 0) && p->ldres % 4 == 0)) && (!p->epi_mode || (((uintptr_t)p->aux % 8 == 0) && p->ldaux % 4 == 0));
  if (p->scat_cout && !miseg_gemm_fuses_scatter(p)) return set_error(MISEG_E_UNSUPPORTED, "gemm: scattered (transposed-conv) store on this shape / path (ask miseg_gemm_fuses_scatter first)");
  if (p->stat && !bstat && !miseg_gemm_fuses_stat(p)) return set_error(MISEG_E_UNSUPPORTED, "gemm: fused statistics on this shape / path (ask miseg_gemm_fuses_stat first)");
  if (p->ta == 0 && p->tb == 0) {
    if ((p->res || p->epi_mode) && (split > 1 || p->accumulate)) return set_error(MISEG_E_BADARG, "gemm: residual / auxiliary epilogue with split_k or accumulate");
    if (p->epi_mode && !p->aux) return set_error(MISEG_E_BADARG, "gemm: epi_mode %d needs aux", p->epi_mode);
    const int kstage = 8 * N16;
    int kps = cdiv(cdiv(p->K, split), kstage) * kstage;
    split = cdiv(p->K, kps);
    if (split > 1 && !p->accumulate) {
      // atomics need a zeroed destination
      if (p->ldc == p->N) { if (fill_words_async(p->C, 0, (size_t)p->M * p->N, s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset"); }
      else if (fill_words_2d_async(p->C, (size_t)p->ldc, 0, (size_t)p->N, (size_t)p->M, s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset2d");
    }
    if (split > 1 && p->act != MISEG_ACT_NONE) return set_error(MISEG_E_BADARG, "gemm: activation with split_k");
    if (split > 1 && p->act != MISEG_ACT_NONE) return set_error(MISEG_E_BADARG, "gemm: activation with split_k");
    if constexpr (std::is_same<T, TO>::value) {
      if (p->K == 1 && split == 1 && !p->accumulate && p->act == MISEG_ACT_NONE && !p->res && !p->epi_mode && p->N % N16 == 0 && ((uintptr_t)p->C % 16 == 0) &&
          p->ldc % N16 == 0) {
        if (p->stat) {
          int sb = cdiv(p->M, 256 / (p->N / N16));
          if (sb > 2048) sb = 2048;
          gemm_nt_k1_stat_kernel<T><<<sb, 256, 0, s>>>((const T*)p->A, p->lda, (const T*)p->B, p->ldb, (T*)p->C, p->ldc, p->M, p->N, p->bias, (double*)p->stat);
          MISEG_LAUNCH_CHECK("gemm_nt_k1(stat)");
          return MISEG_OK;
        }
        int64_t blocks = ((int64_t)p->M * (p->N / N16) + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        gemm_nt_k1_kernel<T><<<(int)blocks, 256, 0, s>>>((const T*)p->A, p->lda, (const T*)p->B, p->ldb, (T*)p->C, p->ldc, p->M, p->N, p->bias);
        MISEG_LAUNCH_CHECK("gemm_nt_k1");
        return MISEG_OK;
      }
    }
    if constexpr (std::is_same<T, bf16>::value && std::is_same<TO, bf16>::value) {
      // deep-stage linears (see gemm_nt_small_kernel)
      if (split == 1 && !p->accumulate && p->M <= 2048 && p->K % 32 == 0 && p->N % 16 == 0 && al_a && al_b && ((uintptr_t)p->C % 8 == 0) && p->ldc % 4 == 0 && epi_vec_ok) {
        // the largest tile that still leaves >= 140 workgroups (sweep over the deep-stage / ViT shapes in the kernel's comment: with fewer
        // the launch is one workgroup's K loop long; beyond, the smaller tile only adds operand traffic)
        // round 4: a grid of at most 256 workgroups - one round on the 256 CUs - beats a finer tile that needs a second, partly empty round
        // (216 x 768 -> 3072: 32 x 64 tiles = 336 workgroups 11.6 us, 64 x 48 tiles = 256 workgroups 9.0 us; 216 x 768 -> 2304: 32 x 64 = 252
        // workgroups 7.6 us, 64 x 48 = 192: 8.9): the candidate with the most workgroups <= 256 wins when it has >= 140, else the rule above
        int mtw = 1, ntw = 1;
        {
          static const int cand[6][2] = {{4, 3}, {2, 4}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
          int best = -1;
          int64_t best_wg = 0;
          for (int c = 0; c < 6; ++c) {
            const int a_ = cand[c][0], b_ = cand[c][1];
            if (p->N % (16 * b_) != 0 || (a_ >= 2 && p->M <= 16 * (a_ / 2))) continue;
            const int64_t wg = (int64_t)cdiv(p->M, 16 * a_) * (p->N / (16 * b_));
            if (wg >= 140 && wg <= 256 && wg > best_wg) { best = c; best_wg = wg; }
          }
          if (best >= 0) { mtw = cand[best][0]; ntw = cand[best][1]; }
          else {
            for (int c = 1; c < 6; ++c) {
              const int a_ = cand[c][0], b_ = cand[c][1];
              if (p->N % (16 * b_) != 0 || (a_ == 2 && p->M <= 16)) continue;
              mtw = a_; ntw = b_;
              if ((int64_t)cdiv(p->M, 16 * a_) * (p->N / (16 * b_)) >= 140) break;
            }
          }
        }
        const int gm = cdiv(p->M, 16 * mtw);
        dim3 grid(gm * (p->N / (16 * ntw)));
        const bool ge = p->act == MISEG_ACT_GELU;
        const int smode = p->an.stat ? 3 : !p->stat ? 0 : bstat ? 2 : 1;      // 3: folded norm on the A side (no statistics with it)
#define SM_ARGS (const bf16*)p->A, p->lda, (const bf16*)p->B, p->ldb, (bf16*)p->C, p->ldc, p->M, p->N, p->K, epi, gm
#define SM_LAUNCH(m_, n_)                                                                                                                       \
  do {                                                                                                                                          \
    if (smode == 3) { if (ge) gemm_nt_small_kernel<m_, n_, true, 0, true><<<grid, 256, 0, s>>>(SM_ARGS); else gemm_nt_small_kernel<m_, n_, false, 0, true><<<grid, 256, 0, s>>>(SM_ARGS); } \
    else if (smode == 2) gemm_nt_small_kernel<m_, n_, false, 2><<<grid, 256, 0, s>>>(SM_ARGS);                                                  \
    else if (smode == 1) gemm_nt_small_kernel<m_, n_, false, 1><<<grid, 256, 0, s>>>(SM_ARGS);                                                  \
    else if (ge) gemm_nt_small_kernel<m_, n_, true><<<grid, 256, 0, s>>>(SM_ARGS);                                                              \
    else gemm_nt_small_kernel<m_, n_, false><<<grid, 256, 0, s>>>(SM_ARGS);                                                                     \
  } while (0)
        if (mtw == 4) SM_LAUNCH(4, 3);
        else if (mtw == 2) { if (ntw == 4) SM_LAUNCH(2, 4); else if (ntw == 2) SM_LAUNCH(2, 2); else SM_LAUNCH(2, 1); }
        else { if (ntw == 4) SM_LAUNCH(1, 4); else if (ntw == 2) SM_LAUNCH(1, 2); else SM_LAUNCH(1, 1); }
#undef SM_LAUNCH
#undef SM_ARGS
        MISEG_LAUNCH_CHECK("gemm_nt_small");
        return MISEG_OK;
      }
      // tall-skinny streaming path (see gemm_nt_stream_kernel)
      const size_t lds = (size_t)p->N * (p->K * 2 + 16) + (size_t)p->N * 4;
      const bool st_ok = split == 1 && !p->accumulate && (p->K == 48 || p->K == 96 || p->K == 192 || p->K == 144 || p->K == 288 || p->K == 384) && p->N % 16 == 0 && p->M >= 4096 && lds <= 96 * 1024 &&
                         al_a && al_b && ((uintptr_t)p->C % 8 == 0) && p->ldc % 4 == 0 && epi_vec_ok;
      const bool stat_ok = st_ok && p->act == MISEG_ACT_NONE && p->N <= 96 && (p->K == 48 || p->K == 96 || p->K == 192);
      if (p->stat && !bstat && !stat_ok) return set_error(MISEG_E_UNSUPPORTED, "gemm: fused statistics on this shape / path (ask miseg_gemm_fuses_stat first)");
      if (p->an.stat) {          // miseg_gemm_fuses_anorm held: K in {48, 96}; the norm's apply pass rides in the operand load
        const int mtiles = cdiv(p->M, 32);
        int blocks = cdiv(mtiles, 4);
        const int cap = lds > 80 * 1024 ? 256 : 512;
        if (blocks > cap) blocks = cap;
        const int ns = nt_stream_groups(blocks, p->N), nper = p->N / ns;
        const size_t ldsa = (size_t)nper * (p->K * 2 + 16) + (size_t)nper * 4 + (size_t)2 * p->K * 4;
#define AN_CASE(k16, g)                                                                                                                       \
  (void)hipFuncSetAttribute((const void*)gemm_nt_stream_kernel<k16, g, 12, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsa); \
  gemm_nt_stream_kernel<k16, g, 12, 0, false, true><<<dim3(blocks, ns), 256, ldsa, s>>>((const bf16*)p->A, p->lda, (const bf16*)p->B, p->ldb, (bf16*)p->C, p->ldc, p->M, p->N, epi, nper)
        const bool ge = p->act == MISEG_ACT_GELU;
        if (p->K == 48) { if (ge) { AN_CASE(3, true); } else { AN_CASE(3, false); } }
        else { if (ge) { AN_CASE(6, true); } else { AN_CASE(6, false); } }
#undef AN_CASE
        MISEG_LAUNCH_CHECK("gemm_nt_stream(anorm)");
        return MISEG_OK;
      }
      if (bstat) {               // miseg_gemm_fuses_bstat held: N in {48, 96}; the norm-backward reduction rides in the epilogue
        const int mtiles = cdiv(p->M, 32);
        int blocks = cdiv(mtiles, 4);
        if (blocks > 512) blocks = 512;
        const int nch = p->K >= 288 ? 3 : 6;      // (the long rows' A fragments fill the registers: fewer accumulator tiles beside the sums)
        int ns = nt_stream_groups(blocks, p->N);
        while (p->N / ns > 16 * nch) ++ns;          // one accumulator chunk per column group (the sums are indexed by the chunk's tile)
        while ((p->N / 16) % ns) ++ns;
        const int nper = p->N / ns;
        size_t ldsb = (size_t)nper * (p->K * 2 + 16) + (size_t)nper * 4 * 3;
        if (ldsb < (size_t)2 * nper * 65 * sizeof(float)) ldsb = (size_t)2 * nper * 65 * sizeof(float);
#define BS_CASE(k16, nch_)                                                                                                                      \
  (void)hipFuncSetAttribute((const void*)gemm_nt_stream_kernel<k16, false, nch_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
  gemm_nt_stream_kernel<k16, false, nch_, 2><<<dim3(blocks, ns), 256, ldsb, s>>>((const bf16*)p->A, p->lda, (const bf16*)p->B, p->ldb, (bf16*)p->C, p->ldc, p->M, p->N, epi, nper)
        switch (p->K) {
          case 48: BS_CASE(3, 6); break;
          case 96: BS_CASE(6, 6); break;
          case 192: BS_CASE(12, 6); break;
          case 144: BS_CASE(9, 6); break;
          default: BS_CASE(18, 3); break;
        }
#undef BS_CASE
        MISEG_LAUNCH_CHECK("gemm_nt_stream(bstat)");
        return MISEG_OK;
      }
      if (p->scat_cout) {        // miseg_gemm_fuses_scatter held: st_ok, K in {48, 96}
        const int mtiles = cdiv(p->M, 32);
        int blocks = cdiv(mtiles, 4);
        const int cap = lds > 80 * 1024 ? 256 : 512;
        if (blocks > cap) blocks = cap;
#define SC_CASE(k16)                                                                                                                         \
  (void)hipFuncSetAttribute((const void*)gemm_nt_stream_kernel<k16, false, 12, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  gemm_nt_stream_kernel<k16, false, 12, false, true><<<blocks, 256, lds, s>>>((const bf16*)p->A, p->lda, (const bf16*)p->B, p->ldb, (bf16*)p->C, p->ldc, p->M, p->N, epi, p->N)
        if (p->K == 48) { SC_CASE(3); } else { SC_CASE(6); }
#undef SC_CASE
        MISEG_LAUNCH_CHECK("gemm_nt_stream(scatter)");
        return MISEG_OK;
      }
      if (st_ok && p->stat) {
        const int mtiles = cdiv(p->M, 32);
        int blocks = cdiv(mtiles, 4);
        if (blocks > 512) blocks = 512;
        const int ns = nt_stream_groups(blocks, p->N), nper = p->N / ns;
        size_t lds2 = (size_t)2 * nper * 65 * sizeof(float);
        const size_t ldsg = (size_t)nper * (p->K * 2 + 16) + (size_t)nper * 4;
        if (lds2 < ldsg) lds2 = ldsg;
#define STS_CASE(k16)                                                                                                                        \
  (void)hipFuncSetAttribute((const void*)gemm_nt_stream_kernel<k16, false, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
  gemm_nt_stream_kernel<k16, false, 6, true><<<dim3(blocks, ns), 256, lds2, s>>>((const bf16*)p->A, p->lda, (const bf16*)p->B, p->ldb, (bf16*)p->C, p->ldc, p->M, p->N, epi, nper)
        if (p->K == 48) { STS_CASE(3); } else if (p->K == 96) { STS_CASE(6); } else { STS_CASE(12); }
#undef STS_CASE
        MISEG_LAUNCH_CHECK("gemm_nt_stream(stat)");
        return MISEG_OK;
      }
      if (st_ok) {
        const int mtiles = cdiv(p->M, 32);
        int blocks = cdiv(mtiles, 4);
        const int cap = lds > 80 * 1024 ? 256 : 512;
        if (blocks > cap) blocks = cap;
        const int ns = nt_stream_groups(blocks, p->N), nper = p->N / ns;
        const size_t ldsg = (size_t)nper * (p->K * 2 + 16) + (size_t)nper * 4;
#define ST_CASE(k16, g, nch)                                                                                                                  \
  (void)hipFuncSetAttribute((const void*)gemm_nt_stream_kernel<k16, g, nch>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg);         \
  gemm_nt_stream_kernel<k16, g, nch><<<dim3(blocks, ns), 256, ldsg, s>>>((const bf16*)p->A, p->lda, (const bf16*)p->B, p->ldb, (bf16*)p->C, p->ldc, p->M, p->N, epi, nper)
        const bool ge = p->act == MISEG_ACT_GELU;
        if (p->K == 48) { if (ge) { ST_CASE(3, true, 12); } else { ST_CASE(3, false, 12); } }
        else if (p->K == 96) { if (ge) { ST_CASE(6, true, 12); } else { ST_CASE(6, false, 12); } }
        else if (p->K == 192) { if (ge) { ST_CASE(12, true, 12); } else { ST_CASE(12, false, 12); } }
        else if (p->K == 144) { if (ge) { ST_CASE(9, true, 12); } else { ST_CASE(9, false, 12); } }
        else if (p->K == 288) { if (ge) { ST_CASE(18, true, 6); } else { ST_CASE(18, false, 6); } }
        else { if (ge) { ST_CASE(24, true, 3); } else { ST_CASE(24, false, 3); } }
#undef ST_CASE
        MISEG_LAUNCH_CHECK("gemm_nt_stream");
        return MISEG_OK;
      }
    }
    if (p->stat) return set_error(MISEG_E_UNSUPPORTED, "gemm: fused statistics on this shape / path (ask miseg_gemm_fuses_stat first)");
    const int nt = (p->N % 64 == 0) ? 4 : (p->N % 48 == 0) ? 3 : (p->N <= 16) ? 1 : (p->N <= 32) ? 2 : (p->N <= 48) ? 3 : 4;
    dim3 grid(cdiv(p->M, 128), cdiv(p->N, 16 * nt), split);
#define NT_CASE(n)                                                                                                                          \
  case n:                                                                                                                                   \
    gemm_nt_kernel<T, TO, n><<<grid, 256, 0, s>>>((const T*)p->A, p->lda, (const T*)p->B, p->ldb, (TO*)p->C, p->ldc, p->M, p->N, p->K, epi, \
                                                  mode, al_a, al_b, kps);                                                          \
    break;
    switch (nt) { NT_CASE(1) NT_CASE(2) NT_CASE(3) NT_CASE(4) }
#undef NT_CASE
  } else if (p->ta == 1 && p->tb == 1) {
    if (p->bias || p->act != MISEG_ACT_NONE) return set_error(MISEG_E_UNSUPPORTED, "gemm TN: no bias/act epilogue");
    if constexpr (std::is_same<T, bf16>::value && std::is_same<TO, float>::value) {
      // streaming path for tall token streams (see gemm_tn_stream_kernel); K is the token count here
      TnStreamPlan pl;
      if (tn_stream_plan(p, &pl)) {
        float* partial = nullptr;
        if (pl.splits > 1) {
          MISEG_REQUIRE(p->workspace, MISEG_E_BADARG, "gemm: workspace required (miseg_gemm_workspace_bytes)");
          partial = (float*)p->workspace;
        }
        const int BM = pl.wm * 48, BN = pl.wn * 48;
        const size_t lds = (size_t)2 * 64 * (BM * 2 + BN * 2 + 32);
        hipFuncSetAttribute((const void*)gemm_tn_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        gemm_tn_stream_kernel<<<dim3(pl.gx, pl.gy, pl.splits), 256, lds, s>>>((const bf16*)p->A, p->lda, (const bf16*)p->B, p->ldb, (float*)p->C, p->ldc, p->M, p->N,
                                                                             p->K, pl.wm, pl.wn, pl.tps, p->accumulate ? 1 : 0, partial, p->tn_colsum);
        if (partial && !p->defer_reduce) {
          const int groups = cdiv(pl.splits, TN_RG);
          if (groups > 1 && !p->accumulate) {   // the groups meet in atomics
            if (p->ldc == p->N) { if (fill_words_async(p->C, 0, (size_t)p->M * p->N, s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset"); }
            else if (fill_words_2d_async(p->C, (size_t)p->ldc, 0, (size_t)p->N, (size_t)p->M, s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset2d");
          }
          gemm_tn_partial_reduce_kernel<<<dim3(cdiv(p->M * (p->N / 4), 256), groups), 256, 0, s>>>(partial, (float*)p->C, p->ldc, p->M, p->N, pl.splits, p->accumulate);
        }
        MISEG_LAUNCH_CHECK("gemm_tn_stream");
        return MISEG_OK;
      }
    }
    if (p->tn_colsum) return set_error(MISEG_E_UNSUPPORTED, "gemm TN: column sums ride on the streaming path only (ask miseg_gemm_tn_fuses_colsum first)");
    constexpr int TN_BK = std::is_same<T, bf16>::value ? 128 : 64;
    if (p->split_k == 0) {   // auto: enough workgroups for the chip, >= 512 reduction rows each
      const int tiles = cdiv(p->M, 64) * cdiv(p->N, 64);
      split = cdiv(p->K, 512);
      if (split > 1024 / tiles) split = 1024 / tiles;
      if (split < 1) split = 1;
    }
    int kps = cdiv(cdiv(p->K, split), TN_BK) * TN_BK;
    split = cdiv(p->K, kps);
    mode = split > 1 ? 2 : (p->accumulate ? 1 : 0);
    if (split > 1 && !p->accumulate) {
      if (p->ldc == p->N) { if (fill_words_async(p->C, 0, (size_t)p->M * p->N, s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset"); }
      else if (fill_words_2d_async(p->C, (size_t)p->ldc, 0, (size_t)p->N, (size_t)p->M, s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset2d");
    }
    dim3 grid(cdiv(p->M, 64), cdiv(p->N, 64), split);
    gemm_tn_kernel<T, TO><<<grid, 256, 0, s>>>((const T*)p->A, p->lda, (const T*)p->B, p->ldb, (TO*)p->C, p->ldc, p->M, p->N, p->K, mode, al_a, al_b, kps);
  } else {
    return set_error(MISEG_E_UNSUPPORTED, "gemm: only NT (ta=tb=0) and TN (ta=tb=1) are implemented");
  }
  MISEG_LAUNCH_CHECK("gemm");
  return MISEG_OK;
}

extern "C" int miseg_gemm(const miseg_gemm_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->A && p->B && p->C, MISEG_E_BADARG, "gemm: null pointer");
  MISEG_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, MISEG_E_BADARG, "gemm: bad shape M=%d N=%d K=%d", p->M, p->N, p->K);
  MISEG_REQUIRE(p->act == MISEG_ACT_NONE || p->act == MISEG_ACT_GELU, MISEG_E_UNSUPPORTED, "gemm: act %d", p->act);
  if (p->dtype == MISEG_F32 && p->out_dtype == MISEG_F32) return launch_gemm<float, float>(p, s);
  if (p->dtype == MISEG_BF16 && p->out_dtype == MISEG_BF16) return launch_gemm<bf16, bf16>(p, s);
  if (p->dtype == MISEG_BF16 && p->out_dtype == MISEG_F32) return launch_gemm<bf16, float>(p, s);
  return set_error(MISEG_E_BADARG, "gemm: dtype %d -> %d", p->dtype, p->out_dtype);
}

extern "C" int miseg_gemm_tn_fuses_colsum(const miseg_gemm_params* p) {
  TnStreamPlan pl;
  return (p && p->ta == 1 && p->tb == 1 && p->dtype == MISEG_BF16 && p->out_dtype == MISEG_F32 && tn_stream_plan(p, &pl)) ? 1 : 0;
}

extern "C" int miseg_gemm_tn_splits(const miseg_gemm_params* p) {
  TnStreamPlan pl;
  return (p && tn_stream_plan(p, &pl)) ? pl.splits : 0;
}

namespace miseg {
struct TnReduceBatch { miseg_tn_reduce_desc d[MISEG_TN_REDUCE_BATCH]; int n; };
// one thread per 4 consecutive n of one problem AND per group of TNB_SG splits (8 loads in flight): a thread that walked all ~400
// splits of a 48x48 problem alone was a 100-deep dependent chain (78 us for the launch); the groups meet in fp32 atomics (<= 16
// arrivals per element), a single group adds with one 16-byte read-modify-write
static constexpr int TNB_SG = 32;
__global__ void __launch_bounds__(256) gemm_tn_reduce_batch_kernel(TnReduceBatch b) {
  int k = 0;
  while (k + 1 < b.n && b.d[k + 1].block0 <= (int)blockIdx.x) ++k;
  const miseg_tn_reduce_desc d = b.d[k];
  const int per = cdiv(d.M * (d.N / 4), 256);                 // blocks per split group
  const int local = blockIdx.x - d.block0, grp = local / per;
  const int i = (local - grp * per) * 256 + threadIdx.x;
  if (i >= d.M * (d.N / 4)) return;
  const int m = i / (d.N / 4), n = (i - m * (d.N / 4)) * 4;
  const float* p = d.partial + (int64_t)m * d.N + n;
  const int64_t stride = (int64_t)d.M * d.N;
  const int s0 = grp * TNB_SG, s1 = min(d.splits, s0 + TNB_SG);
  f32x4 a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  int s = s0;
  for (; s + 7 < s1; s += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += *reinterpret_cast<const f32x4*>(p + (s + u) * stride);
  }
  for (; s < s1; ++s) a[0] += *reinterpret_cast<const f32x4*>(p + s * stride);
  const f32x4 v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  if (d.regroup > 0) {      // see gemm_tn_body: column n = j * regroup + c goes to column c * (N / regroup) + j
    const int per = d.N / d.regroup;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* c = d.C + (int64_t)m * d.ldc + ((n + r) % d.regroup) * per + (n + r) / d.regroup;
      if (d.splits <= TNB_SG) *c += v[r]; else atomicAdd(c, v[r]);
    }
    return;
  }
  float* c = d.C + (int64_t)m * d.ldc + n;
  if (d.splits <= TNB_SG) {
    if ((d.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(d.C) & 15) == 0) {
      f32x4* c4 = reinterpret_cast<f32x4*>(c);
      *c4 = *c4 + v;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] += v[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(c + r, v[r]);
  }
}
}  // namespace miseg

extern "C" int miseg_gemm_tn_reduce_batch(const miseg_tn_reduce_desc* descs, int n, miseg_stream_t s_) {
  MISEG_REQUIRE(descs && n > 0 && n <= MISEG_TN_REDUCE_BATCH, MISEG_E_BADARG, "gemm_tn_reduce_batch: 1..%d reductions", MISEG_TN_REDUCE_BATCH);
  miseg::TnReduceBatch b;
  b.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    MISEG_REQUIRE(descs[i].partial && descs[i].C && descs[i].M > 0 && descs[i].N > 0 && descs[i].N % 4 == 0 && descs[i].splits > 0, MISEG_E_BADARG,
                  "gemm_tn_reduce_batch: descriptor %d", i);
    MISEG_REQUIRE(descs[i].regroup >= 0 && (descs[i].regroup == 0 || descs[i].N % descs[i].regroup == 0), MISEG_E_BADARG, "gemm_tn_reduce_batch: descriptor %d: regroup", i);
    b.d[i] = descs[i];
    b.d[i].block0 = blocks;
    blocks += cdiv(descs[i].M * (descs[i].N / 4), 256) * cdiv(descs[i].splits, miseg::TNB_SG);
  }
  miseg::gemm_tn_reduce_batch_kernel<<<blocks, 256, 0, (hipStream_t)s_>>>(b);
  MISEG_LAUNCH_CHECK("gemm_tn_reduce_batch");
  return MISEG_OK;
}

extern "C" int miseg_gemm_tn_group(const miseg_gemm_tn_desc* descs, int n, int dtype, miseg_stream_t s_) {
  MISEG_REQUIRE(descs && n > 0 && n <= MISEG_GEMM_GROUP, MISEG_E_BADARG, "gemm_tn_group: 1..%d problems", MISEG_GEMM_GROUP);
  MISEG_REQUIRE(dtype == MISEG_BF16 || dtype == MISEG_F32, MISEG_E_BADARG, "gemm_tn_group: dtype %d", dtype);
  TnGroup g;
  g.n = n;
  int blocks = 0;
  const int n16 = dtype == MISEG_F32 ? 4 : 8, bk = dtype == MISEG_F32 ? 64 : miseg::TN_GROUP_BK;
  for (int i = 0; i < n; ++i) {
    const miseg_gemm_tn_desc& d = descs[i];
    MISEG_REQUIRE(d.A && d.B && d.C && d.M > 0 && d.N > 0 && d.K > 0, MISEG_E_BADARG, "gemm_tn_group: problem %d", i);
    TnGroup::P& q = g.p[i];
    q.A = d.A; q.lda = d.lda; q.B = d.B; q.ldb = d.ldb; q.C = d.C; q.ldc = d.ldc; q.M = d.M; q.N = d.N; q.K = d.K;
    MISEG_REQUIRE(d.regroup >= 0 && (d.regroup == 0 || d.N % d.regroup == 0), MISEG_E_BADARG, "gemm_tn_group: problem %d: regroup %d does not divide N %d", i, d.regroup, d.N);
    q.regroup = d.regroup;
    q.gx = cdiv(d.M, 64); q.gy = cdiv(d.N, 64);
    // split the reduction only when it is long: split partials meet in fp32 atomics, and ~4 M of them (the 12^3-stage problems,
    // K = 1728, split 4 ways) cost 3x what the whole family takes unsplit (75 -> 25 us); the group as a whole fills the chip
    int split = cdiv(d.K, 2048);
    if (split > 1024 / (q.gx * q.gy)) split = 1024 / (q.gx * q.gy);
    if (split < 1) split = 1;
    q.kps = cdiv(cdiv(d.K, split), bk) * bk;
    split = cdiv(d.K, q.kps);
    q.mode = split > 1 ? 2 : (d.zeroed ? 0 : 1);     // accumulate into C (atomics when the reduction is split) unless C is known to be zero
    q.va = ((uintptr_t)d.A % 16 == 0) && (d.lda % n16 == 0);
    q.vb = ((uintptr_t)d.B % 16 == 0) && (d.ldb % n16 == 0);
    q.block0 = blocks;
    blocks += q.gx * q.gy * split;
  }
  if (dtype == MISEG_BF16) gemm_tn_group_kernel<bf16><<<blocks, 256, 0, (hipStream_t)s_>>>(g);
  else gemm_tn_group_kernel<float><<<blocks, 256, 0, (hipStream_t)s_>>>(g);
  MISEG_LAUNCH_CHECK("gemm_tn_group");
  return MISEG_OK;
}

extern "C" int miseg_permute3(const float* src, float* dst, int n0, int n1, int n2, int64_t s0, int64_t s1, int64_t s2, int accumulate, miseg_stream_t s_) {
  MISEG_REQUIRE(src && dst && n0 > 0 && n1 > 0 && n2 > 0, MISEG_E_BADARG, "permute3: bad args");
  const int64_t total = (int64_t)n0 * n1 * n2;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  permute3_kernel<<<g, 256, 0, (hipStream_t)s_>>>(src, dst, n0, n1, n2, s0, s1, s2, accumulate);
  MISEG_LAUNCH_CHECK("permute3");
  return MISEG_OK;
}
