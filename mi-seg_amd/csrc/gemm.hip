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

__device__ __forceinline__ float gelu_erf(float t) { return 0.5f * t * (1.f + erff(t * 0.70710678118654752f)); }

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
__device__ __forceinline__ void store_out4(TO* C, int64_t ldc, int m, int n, int M, int N, f32x4 v, const float* bias, int act, int mode /*0 store,1 add,2 atomic*/) {
  if (m >= M) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (n + r < N) {
      float x = v[r];
      if (bias) x += bias[n + r];
      if (act == MISEG_ACT_GELU) x = gelu_erf(x);
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
                                                      int M, int N, int K, const float* __restrict__ bias, int act, int mode, bool vec_a, bool vec_b,
                                                      int k_per_split) {
  typedef typename Vec16<T>::type VT;
  constexpr int KPC = Mma<T>::KPC;
  constexpr int BM = 128, BN = 16 * NT, CH = 8 /*16B chunks per row per stage*/, ROWB = CH * 16 + 16 /*bytes, padded*/;
  __shared__ __attribute__((aligned(16))) char lds[(BM + BN) * ROWB];
  char* lA = lds;
  char* lB = lds + BM * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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
  }
  const float* b_eff = (blockIdx.z == 0) ? bias : nullptr;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      store_out4<TO>(C, ldc, m0 + wave * 32 + mt * 16 + fi, n0 + nt * 16 + fq * 4, M, N, acc[mt][nt], b_eff, act, mode);
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

template <class T, class TO>
__global__ void __launch_bounds__(256) gemm_tn_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, TO* __restrict__ C, int64_t ldc,
                                                      int M, int N, int K, int mode, bool vec_a, bool vec_b, int k_per_split) {
  typedef typename Vec16<T>::type VT;
  constexpr int KPC = Mma<T>::KPC;     // elements per 16-byte chunk (here along m / n)
  constexpr int BM = 64, BN = 64, BK = std::is_same<T, bf16>::value ? 128 : 64;   // k rows per stage
  constexpr int ROWB = BM * (int)sizeof(T) + 16;   // bytes per k-row of a tile (padded)
  __shared__ __attribute__((aligned(16))) char lds[2 * BK * ROWB];
  char* lA = lds;
  char* lB = lds + BK * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      store_out4<TO>(C, ldc, m0 + wm * 32 + mt * 16 + fi, n0 + wn * 32 + nt * 16 + fq * 4, M, N, acc[mt][nt], nullptr, MISEG_ACT_NONE, mode);
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
  if (p->ta == 0 && p->tb == 0) {
    const int kstage = 8 * N16;
    int kps = cdiv(cdiv(p->K, split), kstage) * kstage;
    split = cdiv(p->K, kps);
    if (split > 1 && !p->accumulate) {
      // atomics need a zeroed destination
      if (p->ldc == p->N) { if (hipMemsetAsync(p->C, 0, (size_t)p->M * p->N * sizeof(float), s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset"); }
      else if (hipMemset2DAsync(p->C, p->ldc * sizeof(float), 0, (size_t)p->N * sizeof(float), p->M, s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset2d");
    }
    if (split > 1 && p->act != MISEG_ACT_NONE) return set_error(MISEG_E_BADARG, "gemm: activation with split_k");
    const int nt = (p->N % 64 == 0) ? 4 : (p->N % 48 == 0) ? 3 : (p->N <= 16) ? 1 : (p->N <= 32) ? 2 : (p->N <= 48) ? 3 : 4;
    dim3 grid(cdiv(p->M, 128), cdiv(p->N, 16 * nt), split);
#define NT_CASE(n)                                                                                                                          \
  case n:                                                                                                                                   \
    gemm_nt_kernel<T, TO, n><<<grid, 256, 0, s>>>((const T*)p->A, p->lda, (const T*)p->B, p->ldb, (TO*)p->C, p->ldc, p->M, p->N, p->K, p->bias, \
                                                  p->act, mode, al_a, al_b, kps);                                                          \
    break;
    switch (nt) { NT_CASE(1) NT_CASE(2) NT_CASE(3) NT_CASE(4) }
#undef NT_CASE
  } else if (p->ta == 1 && p->tb == 1) {
    if (p->bias || p->act != MISEG_ACT_NONE) return set_error(MISEG_E_UNSUPPORTED, "gemm TN: no bias/act epilogue");
    constexpr int TN_BK = std::is_same<T, bf16>::value ? 128 : 64;
    int kps = cdiv(cdiv(p->K, split), TN_BK) * TN_BK;
    split = cdiv(p->K, kps);
    mode = split > 1 ? 2 : (p->accumulate ? 1 : 0);
    if (split > 1 && !p->accumulate) {
      if (p->ldc == p->N) { if (hipMemsetAsync(p->C, 0, (size_t)p->M * p->N * sizeof(float), s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset"); }
      else if (hipMemset2DAsync(p->C, p->ldc * sizeof(float), 0, (size_t)p->N * sizeof(float), p->M, s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "gemm: memset2d");
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

extern "C" int miseg_permute3(const float* src, float* dst, int n0, int n1, int n2, int64_t s0, int64_t s1, int64_t s2, int accumulate, miseg_stream_t s_) {
  MISEG_REQUIRE(src && dst && n0 > 0 && n1 > 0 && n2 > 0, MISEG_E_BADARG, "permute3: bad args");
  const int64_t total = (int64_t)n0 * n1 * n2;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  permute3_kernel<<<g, 256, 0, (hipStream_t)s_>>>(src, dst, n0, n1, n2, s0, s1, s2, accumulate);
  MISEG_LAUNCH_CHECK("permute3");
  return MISEG_OK;
}
