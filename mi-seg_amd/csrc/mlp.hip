// Fused MLP of the high-resolution Swin blocks: y = W2 gelu(W1 x + b1) + b2 (+ res), tokens x 48 -> 192 -> 48
// (MONAI MLPBlock as wired at networks/blocks/swin_transformer_block.py:97,176-205; exact-erf GELU).
//
// Unfused, a stage-0 block moves 411 MB through HBM for its MLP, 300 MB of it the hidden pre-activation z and activation h
// (110,592 tokens x 192): fc1 writes both, fc2 reads h, the backward reads z and h again.  Here the hidden tile never leaves the wave:
// with v_mfma_f32_16x16x16_bf16 the accumulator layout of  z^T[hid][tok] = W1 x^T  (lane = token column fi, rows = 4 hidden channels
// 4kg..4kg+3 of tile j) is exactly the B-operand layout of  y^T[c][tok] = W2 h^T  (lane = token column fi, k = 4 hidden channels), so
// gelu(z_j) rounded to bf16 goes straight back into the matrix pipe, one 16-channel slice of the hidden layer at a time (no 192-wide
// accumulator either).  The backward kernel recomputes z_j the same way (same instructions, same order: the same h as the forward),
// forms dz_j = (W2^T dy^T)_j * gelu'(z_j), feeds it into dx^T += W1^T dz_j, and writes dz and h once for the two weight-gradient products,
// which stay miseg_gemm TN launches (their 27 x 9 accumulator tiles per wave do not fit beside this).
// A operands (weights) come from LDS images with 16-byte padded rows (conflict-free 8-byte fragment reads), B operands (tokens)
// straight from global memory: a lane's operand is 8 contiguous bytes of its token's row.
#include "common.h"

namespace miseg {

typedef __attribute__((ext_vector_type(4))) short s16x4_m;

__device__ __forceinline__ f32x4 mma16(const s16x4_m& a, const s16x4_m& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// cdf(t) = 0.5 (1 + erf(t / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7), e = exp(-t^2 / 2): the expressions of gemm.hip
__device__ __forceinline__ float mlp_gelu_cdf(float t, float& e) {
  const float a = fabsf(t) * 0.70710678118654752f;
  const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));      // v_rcp_f32 (1 ulp); __frcp_rn is the ten-instruction IEEE division
  float p = fmaf(1.061405429f, k, -1.453152027f);
  p = fmaf(p, k, 1.421413741f);
  p = fmaf(p, k, -0.284496736f);
  p = fmaf(p, k, 0.254829592f);
  e = __expf(-a * a);
  const float half_tail = 0.5f * p * k * e;
  return t >= 0.f ? 1.f - half_tail : half_tail;
}

static constexpr int MLP_C = 48, MLP_H = 192, MLP_CS = MLP_C / 16, MLP_HS = MLP_H / 16;
static constexpr int MLP_ROW_C = MLP_C * 2 + 16;      // byte stride of an LDS row of C channels  ([HID][C] images)
static constexpr int MLP_ROW_H = MLP_H * 2 + 16;      // ... of HID channels ([C][HID] images)

// image of a [R][K] bf16 matrix with padded rows; K * 2 bytes per row are copied as 16-byte pieces
// (round 5: eight loads in flight per thread before the first LDS write - one piece per trip waited a memory round trip per 8 KB of the
// workgroup's image: 8 - 12 trips for the 48-channel images, 19 for the 96-channel ones, most of the launch at 1 - 2 tiles per wave)
__device__ __forceinline__ void mlp_stage(char* img, int rowb, const bf16* __restrict__ w, int R, int K) {
  const int per = K / 8, total = R * per, nthr = blockDim.x;
  constexpr int NB = 8;      // (16: the 96-channel launches 23 -> 22 us, the 48-channel ones 28.5 -> 29.5)
  for (int base = 0; base < total; base += nthr * NB) {
    bf16x8 v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int c = base + u * nthr + (int)threadIdx.x, cc = c < total ? c : 0;      // (clamped: unconditional loads, exact vmcnt)
      const int r = cc / per, ch = cc - r * per;
      v[u] = *reinterpret_cast<const bf16x8*>(w + (int64_t)r * K + ch * 8);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int c = base + u * nthr + (int)threadIdx.x;
      const int r = c / per, ch = c - r * per;
      if (c < total) *reinterpret_cast<bf16x8*>(img + r * rowb + ch * 16) = v[u];
    }
  }
}
__device__ __forceinline__ s16x4_m mlp_frag(const char* img, int rowb, int row, int k) { return *reinterpret_cast<const s16x4_m*>(img + row * rowb + k * 2); }

// round 5: the (conditional) instance norm in front of the MLP (norm2 of the Swin block, swin_transformer_block.py:176-205) folded into
// the token load (ANORM: x is the norm's RAW input, one sample; fma(x, sc, sh) rounded to bf16 exactly as instnorm_apply_kernel stores it;
// `out` receives norm(x) for the backward pass), and the norm's backward sums (sum dx, sum dx * xhat) in the epilogue of mlp_bwd_kernel (BSTAT)
struct MlpNorm {
  const double* stat; const int32_t* styles; const float* gamma[MISEG_MAX_STYLES]; const float* beta[MISEG_MAX_STYLES]; float eps;
  bf16* out; int64_t ldo;
};
struct MlpBstat { const bf16* x; int64_t ldx; const double* stat_in; float eps; double* dstat; };

// (norm.hip::mean_rstd / gather_stat for one channel of a one-sample statistics buffer [16][1][C][2])
__device__ __forceinline__ void mlp_mean_rstd(const double* __restrict__ stat, int C, int ch, double invS, float eps, float& m, float& rs) {
  double sum = 0.0, sq = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) { sum += stat[((int64_t)r * C + ch) * 2]; sq += stat[((int64_t)r * C + ch) * 2 + 1]; }
  const double mu = sum * invS;
  double var = fma(sq, invS, -mu * mu);
  if (var < 0.0) var = 0.0;
  m = (float)mu;
  rs = 1.0f / sqrtf((float)var + eps);
}

// STAT: instance-norm statistics of the rounded output (one sample), layout / reduction of gemm_nt_stream_kernel
static constexpr int MLP_FWD_WAVES = 8;
// CC (round 5): channels of the block - 48 (stage 1: 63 KB of weight images, two workgroups per CU) or 96 (stage 2: 155 KB, one workgroup per CU;
// until then two GEMM launches that wrote the hidden pre-activation and activation, 21 MB, for the backward pass to read back)
template <bool STAT, bool ANORM = false, int CC = 48>
__global__ void __launch_bounds__(MLP_FWD_WAVES * 64, CC == 48 ? 2 : 1) mlp_fwd_kernel(const bf16* __restrict__ x, int64_t ldx, const bf16* __restrict__ w1, const float* __restrict__ b1,
                                                         const bf16* __restrict__ w2, const float* __restrict__ b2, const bf16* __restrict__ res, int64_t ldres,
                                                         bf16* __restrict__ y, int64_t ldy, int M, double* __restrict__ stat, MlpNorm an) {
  constexpr int MLP_C = CC, MLP_H = 4 * CC, MLP_CS = CC / 16, MLP_HS = MLP_H / 16, MLP_ROW_C = CC * 2 + 16, MLP_ROW_H = MLP_H * 2 + 16;      // (shadow the stage-1 constants)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* w1i = lds;                                  // [HID][C]
  char* w2i = w1i + MLP_H * MLP_ROW_C;              // [C][HID]
  float* lb1 = reinterpret_cast<float*>(w2i + MLP_C * MLP_ROW_H);
  float* lb2 = lb1 + MLP_H;
  float* lsc = lb2 + MLP_C;                         // ANORM: scale / shift per channel
  float* lsh = lsc + MLP_C;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, kg = lane >> 4;
  const int mtiles = (M + 15) / 16, nwaves = gridDim.x * MLP_FWD_WAVES;
  s16x4_m xc[MLP_CS], xn[MLP_CS];
  auto loadx = [&](int tile, s16x4_m (&f)[MLP_CS]) {
    const bf16* p = x + (int64_t)min(tile * 16 + fi, M - 1) * ldx + 4 * kg;
#pragma unroll
    for (int s = 0; s < MLP_CS; ++s) f[s] = *reinterpret_cast<const s16x4_m*>(p + 16 * s);
  };
  int tile = blockIdx.x * MLP_FWD_WAVES + wave;
  if (tile < mtiles) loadx(tile, xc);
  mlp_stage(w1i, MLP_ROW_C, w1, MLP_H, MLP_C);
  mlp_stage(w2i, MLP_ROW_H, w2, MLP_C, MLP_H);
  for (int i = tid; i < MLP_H; i += MLP_FWD_WAVES * 64) lb1[i] = b1 ? b1[i] : 0.f;
  for (int i = tid; i < MLP_C; i += MLP_FWD_WAVES * 64) lb2[i] = b2 ? b2[i] : 0.f;
  if constexpr (ANORM) {
    const int st = an.styles ? an.styles[0] : 0;      // (select chain: no run-time index into the by-value argument struct)
    const float* g = st == 0 ? an.gamma[0] : st == 1 ? an.gamma[1] : st == 2 ? an.gamma[2] : an.gamma[3];
    const float* be = st == 0 ? an.beta[0] : st == 1 ? an.beta[1] : st == 2 ? an.beta[2] : an.beta[3];
    for (int k = tid; k < MLP_C; k += MLP_FWD_WAVES * 64) {
      float m_, r_;
      mlp_mean_rstd(an.stat, MLP_C, k, 1.0 / M, an.eps, m_, r_);
      const float sc = r_ * (g ? g[k] : 1.f);
      lsc[k] = sc;
      lsh[k] = (be ? be[k] : 0.f) - m_ * sc;
    }
  }
  __syncthreads();
  float asc[MLP_CS][4], ash[MLP_CS][4];
  if constexpr (ANORM) {
#pragma unroll
    for (int s = 0; s < MLP_CS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) { asc[s][e] = lsc[16 * s + 4 * kg + e]; ash[s][e] = lsh[16 * s + 4 * kg + e]; }
  }
  auto normx = [&](int tile_, s16x4_m (&f)[MLP_CS]) {
    if constexpr (ANORM) {
      const int row = tile_ * 16 + fi;
      bf16* orow = (an.out && row < M) ? an.out + (int64_t)row * an.ldo + 4 * kg : nullptr;
#pragma unroll
      for (int s = 0; s < MLP_CS; ++s) {
        bf16x4 v = __builtin_bit_cast(bf16x4, f[s]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16)fmaf((float)v[e], asc[s][e], ash[s][e]);
        f[s] = __builtin_bit_cast(s16x4_m, v);
        if (orow) *reinterpret_cast<bf16x4*>(orow + 16 * s) = v;
      }
    }
  };
  if (blockIdx.x * MLP_FWD_WAVES + wave < mtiles) normx(blockIdx.x * MLP_FWD_WAVES + wave, xc);
  float ssum[MLP_CS][4], ssq[MLP_CS][4];
#pragma unroll
  for (int i = 0; i < MLP_CS; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
  for (; tile < mtiles; tile += nwaves) {
    if (tile + nwaves < mtiles) loadx(tile + nwaves, xn);
    const int row = tile * 16 + fi;
    f32x4 yacc[MLP_CS];
#pragma unroll
    for (int i = 0; i < MLP_CS; ++i) yacc[i] = *reinterpret_cast<const f32x4*>(lb2 + 16 * i + 4 * kg);
    // (not unrolled further: fully unrolled the compiler hoists all 72 weight fragments of a tile into registers - 251 VGPRs, two waves
    // per SIMD - and the GELU's vector work of one slice has nothing to overlap with)
#pragma unroll 2
    for (int j = 0; j < MLP_HS; ++j) {
      f32x4 z = *reinterpret_cast<const f32x4*>(lb1 + 16 * j + 4 * kg);
#pragma unroll
      for (int s = 0; s < MLP_CS; ++s) z = mma16(mlp_frag(w1i, MLP_ROW_C, 16 * j + fi, 16 * s + 4 * kg), xc[s], z);
      float e;
      const bf16x4 h = bf16x4{(bf16)(z[0] * mlp_gelu_cdf(z[0], e)), (bf16)(z[1] * mlp_gelu_cdf(z[1], e)), (bf16)(z[2] * mlp_gelu_cdf(z[2], e)),
                              (bf16)(z[3] * mlp_gelu_cdf(z[3], e))};
      const s16x4_m hb = __builtin_bit_cast(s16x4_m, h);
#pragma unroll
      for (int i = 0; i < MLP_CS; ++i) yacc[i] = mma16(mlp_frag(w2i, MLP_ROW_H, 16 * i + fi, 16 * j + 4 * kg), hb, yacc[i]);
    }
    if (row < M) {
#pragma unroll
      for (int i = 0; i < MLP_CS; ++i) {
        f32x4 v = yacc[i];
        if (res) {
          const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(res + (int64_t)row * ldres + 16 * i + 4 * kg);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)r4[r];
        }
        const bf16x4 o4 = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(y + (int64_t)row * ldy + 16 * i + 4 * kg) = o4;
        if constexpr (STAT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float q = (float)o4[r]; ssum[i][r] += q; ssq[i][r] = fmaf(q, q, ssq[i][r]); }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < MLP_CS; ++s) xc[s] = xn[s];
    if (tile + nwaves < mtiles) normx(tile + nwaves, xc);
  }
  if constexpr (STAT) {
    __syncthreads();      // the weight images are dead once every wave is past its last tile
    float* red = reinterpret_cast<float*>(lds);
    constexpr int RS = MLP_FWD_WAVES * 16 + 1, N = MLP_C;
#pragma unroll
    for (int i = 0; i < MLP_CS; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 16 * i + 4 * kg + r;
        red[(0 * N + col) * RS + wave * 16 + fi] = ssum[i][r];
        red[(1 * N + col) * RS + wave * 16 + fi] = ssq[i][r];
      }
    __syncthreads();
    for (int o = tid; o < 2 * N; o += MLP_FWD_WAVES * 64) {
      const int k = o / N, col = o - k * N;
      const float* rp = red + (k * N + col) * RS;
      float tot = 0.f;
#pragma unroll 16
      for (int i = 0; i < MLP_FWD_WAVES * 16; ++i) tot += rp[i];
      atomicAdd(stat + ((int64_t)(blockIdx.x & 15) * N + col) * 2 + k, (double)tot);      // [16 replicas][B = 1][N][2]
    }
  }
}

// 8 waves per workgroup: the three weight images (63 KB) allow two workgroups per CU, and 16 resident waves hide the GELU's vector work
static constexpr int MLP_BWD_WAVES = 8;
// CC = 96 (round 5): W1 and W2^T alone fill the LDS (2 x 80 KB); the A operand of dx^T += W1^T dz_j is read TRANSPOSED from the W1 image
// (ds_read_b64_tr_b16) instead of from a third, transposed image
template <bool BSTAT, int CC = 48>
__global__ void __launch_bounds__(MLP_BWD_WAVES * 64, CC == 48 ? 2 : 1) mlp_bwd_kernel(const bf16* __restrict__ x, int64_t ldx, const bf16* __restrict__ dy, int64_t lddy,
                                                         const bf16* __restrict__ w1, const float* __restrict__ b1, const bf16* __restrict__ w2t,
                                                         const bf16* __restrict__ w1t, bf16* __restrict__ dz, int64_t lddz, bf16* __restrict__ h, int64_t ldh,
                                                         bf16* __restrict__ dx, int64_t lddx, int M, MlpBstat bs) {
  constexpr int MLP_C = CC, MLP_H = 4 * CC, MLP_CS = CC / 16, MLP_HS = MLP_H / 16, MLP_ROW_C = CC * 2 + 16, MLP_ROW_H = MLP_H * 2 + 16;      // (shadow the stage-1 constants)
  constexpr bool W1T_IMAGE = CC == 48;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* w1i = lds;                                  // [HID][C]   z = W1 x
  char* w2ti = w1i + MLP_H * MLP_ROW_C;             // [HID][C]   dh = W2^T dy
  char* w1ti = w2ti + MLP_H * MLP_ROW_C;            // [C][HID]   dx = W1^T dz   (CC == 48 only)
  float* lb1 = reinterpret_cast<float*>(w1ti + (W1T_IMAGE ? MLP_C * MLP_ROW_H : 0));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, kg = lane >> 4;
  const int mtiles = (M + 15) / 16, nwaves = gridDim.x * MLP_BWD_WAVES;
  s16x4_m xc[MLP_CS], gc[MLP_CS], xn[MLP_CS], gn[MLP_CS];
  auto load2 = [&](int tile, s16x4_m (&fx)[MLP_CS], s16x4_m (&fg)[MLP_CS]) {
    const int r = min(tile * 16 + fi, M - 1);
    const bf16* px = x + (int64_t)r * ldx + 4 * kg;
    const bf16* pg = dy + (int64_t)r * lddy + 4 * kg;
#pragma unroll
    for (int s = 0; s < MLP_CS; ++s) { fx[s] = *reinterpret_cast<const s16x4_m*>(px + 16 * s); fg[s] = *reinterpret_cast<const s16x4_m*>(pg + 16 * s); }
  };
  int tile = blockIdx.x * MLP_BWD_WAVES + wave;
  if (tile < mtiles) load2(tile, xc, gc);
  mlp_stage(w1i, MLP_ROW_C, w1, MLP_H, MLP_C);
  mlp_stage(w2ti, MLP_ROW_C, w2t, MLP_H, MLP_C);
  if constexpr (W1T_IMAGE) mlp_stage(w1ti, MLP_ROW_H, w1t, MLP_C, MLP_H);
  for (int i = tid; i < MLP_H; i += MLP_BWD_WAVES * 64) lb1[i] = b1 ? b1[i] : 0.f;
  float* lmean = lb1 + MLP_H;        // BSTAT: mean / rstd of the norm in front of the MLP
  float* lrstd = lmean + MLP_C;
  if constexpr (BSTAT) {
    for (int k = tid; k < MLP_C; k += MLP_BWD_WAVES * 64) mlp_mean_rstd(bs.stat_in, MLP_C, k, 1.0 / M, bs.eps, lmean[k], lrstd[k]);
  }
  __syncthreads();
  float bsum[MLP_CS][4], bsq[MLP_CS][4];
#pragma unroll
  for (int i = 0; i < MLP_CS; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { bsum[i][r] = 0.f; bsq[i][r] = 0.f; }
  for (; tile < mtiles; tile += nwaves) {
    if (tile + nwaves < mtiles) load2(tile + nwaves, xn, gn);
    const int row = tile * 16 + fi;
    const bool live = row < M;
    f32x4 dxacc[MLP_CS];
#pragma unroll
    for (int i = 0; i < MLP_CS; ++i) dxacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int j = 0; j < MLP_HS; ++j) {
      f32x4 z = *reinterpret_cast<const f32x4*>(lb1 + 16 * j + 4 * kg);
      f32x4 dh = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < MLP_CS; ++s) {
        z = mma16(mlp_frag(w1i, MLP_ROW_C, 16 * j + fi, 16 * s + 4 * kg), xc[s], z);
        dh = mma16(mlp_frag(w2ti, MLP_ROW_C, 16 * j + fi, 16 * s + 4 * kg), gc[s], dh);
      }
      float hv[4], dzv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e;
        const float c = mlp_gelu_cdf(z[r], e);
        hv[r] = z[r] * c;
        dzv[r] = dh[r] * fmaf(z[r] * 0.39894228040143268f, e, c);
      }
      const bf16x4 h4 = bf16x4{(bf16)hv[0], (bf16)hv[1], (bf16)hv[2], (bf16)hv[3]};
      const bf16x4 d4 = bf16x4{(bf16)dzv[0], (bf16)dzv[1], (bf16)dzv[2], (bf16)dzv[3]};
      if (live) {
        *reinterpret_cast<bf16x4*>(h + (int64_t)row * ldh + 16 * j + 4 * kg) = h4;
        *reinterpret_cast<bf16x4*>(dz + (int64_t)row * lddz + 16 * j + 4 * kg) = d4;
      }
      const s16x4_m db = __builtin_bit_cast(s16x4_m, d4);
#pragma unroll
      for (int i = 0; i < MLP_CS; ++i) {
        s16x4_m wf;
        if constexpr (W1T_IMAGE) wf = mlp_frag(w1ti, MLP_ROW_H, 16 * i + fi, 16 * j + 4 * kg);
        else      // lane (row c = 16 i + fi, k = hidden 16 j + 4 kg ..): element [hid][c] of the W1 image, four rows down one column
          wf = __builtin_bit_cast(s16x4_m, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                   (__attribute__((address_space(3))) bf16x4*)(w1i + (16 * j + 4 * kg + (fi >> 2)) * MLP_ROW_C + (16 * i + 4 * (fi & 3)) * 2)));
        dxacc[i] = mma16(wf, db, dxacc[i]);
      }
    }
    if (live && dx) {
#pragma unroll
      for (int i = 0; i < MLP_CS; ++i) {
        const bf16x4 o4 = bf16x4{(bf16)dxacc[i][0], (bf16)dxacc[i][1], (bf16)dxacc[i][2], (bf16)dxacc[i][3]};
        *reinterpret_cast<bf16x4*>(dx + (int64_t)row * lddx + 16 * i + 4 * kg) = o4;
        if constexpr (BSTAT) {      // the terms of instnorm_bwd_reduce_kernel on the rounded gradient: s += g, q = fma(g, (x - m) * rs, q)
          const bf16x4 x4 = *reinterpret_cast<const bf16x4*>(bs.x + (int64_t)row * bs.ldx + 16 * i + 4 * kg);
          const f32x4 m4 = *reinterpret_cast<const f32x4*>(lmean + 16 * i + 4 * kg), r4 = *reinterpret_cast<const f32x4*>(lrstd + 16 * i + 4 * kg);
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float q = (float)o4[r]; bsum[i][r] += q; bsq[i][r] = fmaf(q, ((float)x4[r] - m4[r]) * r4[r], bsq[i][r]); }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < MLP_CS; ++s) { xc[s] = xn[s]; gc[s] = gn[s]; }
  }
  if constexpr (BSTAT) {
    __syncthreads();      // the weight images are dead once every wave is past its last tile
    float* red = reinterpret_cast<float*>(lds);
    constexpr int RS = MLP_BWD_WAVES * 16 + 1, N = MLP_C;
#pragma unroll
    for (int i = 0; i < MLP_CS; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 16 * i + 4 * kg + r;
        red[(0 * N + col) * RS + wave * 16 + fi] = bsum[i][r];
        red[(1 * N + col) * RS + wave * 16 + fi] = bsq[i][r];
      }
    __syncthreads();
    for (int o = tid; o < 2 * N; o += MLP_BWD_WAVES * 64) {
      const int k = o / N, col = o - k * N;
      const float* rp = red + (k * N + col) * RS;
      float tot = 0.f;
#pragma unroll 16
      for (int i = 0; i < MLP_BWD_WAVES * 16; ++i) tot += rp[i];
      atomicAdd(bs.dstat + ((int64_t)(blockIdx.x & 15) * N + col) * 2 + k, (double)tot);      // [16 replicas][B = 1][N][2]
    }
  }
}

}  // namespace miseg

using namespace miseg;

extern "C" int miseg_mlp_fused(int M, int C, int HID, int dtype) { return dtype == MISEG_BF16 && (C == 48 || C == 96) && HID == 4 * C && M >= 4096; }

static int mlp_check(const miseg_mlp_params* p, const char* what) {
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_mlp_params), MISEG_E_BADARG, "%s: struct_size %u != %zu (header / binding drift)", what, p ? p->struct_size : 0u,
                sizeof(miseg_mlp_params));
  MISEG_REQUIRE(miseg_mlp_fused(p->M, p->C, p->HID, p->dtype), MISEG_E_UNSUPPORTED, "%s: M %d C %d HID %d dtype %d (ask miseg_mlp_fused)", what, p->M, p->C, p->HID,
                p->dtype);
  MISEG_REQUIRE(p->x && p->w1 && p->ldx >= p->C && p->ldx % 4 == 0 && ((uintptr_t)p->x % 8) == 0 && ((uintptr_t)p->w1 % 16) == 0, MISEG_E_BADARG,
                "%s: x / w1 (8-byte aligned rows, 16-byte aligned weights)", what);
  return MISEG_OK;
}

extern "C" int miseg_mlp_fwd(const miseg_mlp_params* p, miseg_stream_t s_) {
  if (int rc = mlp_check(p, "mlp_fwd")) return rc;
  MISEG_REQUIRE(p->w2 && p->y && ((uintptr_t)p->w2 % 16) == 0 && ((uintptr_t)p->y % 8) == 0 && p->ldy % 4 == 0 && p->ldy >= p->C, MISEG_E_BADARG, "mlp_fwd: w2 / y");
  MISEG_REQUIRE(!p->res || (((uintptr_t)p->res % 8) == 0 && p->ldres % 4 == 0), MISEG_E_BADARG, "mlp_fwd: res alignment");
  const int Cc = p->C, Hh = 4 * Cc, ROW_C = Cc * 2 + 16, ROW_H = Hh * 2 + 16;
  size_t lds = (size_t)Hh * ROW_C + (size_t)Cc * ROW_H + (size_t)(Hh + Cc + 2 * Cc) * 4;
  const size_t red = (size_t)2 * Cc * (MLP_FWD_WAVES * 16 + 1) * sizeof(float);      // the statistics reduction re-uses the images
  if (p->stat && lds < red) lds = red;
  int blocks = cdiv(cdiv(p->M, 16), MLP_FWD_WAVES);
  if (blocks > 512) blocks = 512;
  hipStream_t s = (hipStream_t)s_;
  MlpNorm an{};
  if (p->an.stat) {
    MISEG_REQUIRE(p->an.num_styles >= 1 && p->an.num_styles <= MISEG_MAX_STYLES, MISEG_E_BADARG, "mlp_fwd: an.num_styles %d", p->an.num_styles);
    MISEG_REQUIRE(!p->an_out || (((uintptr_t)p->an_out % 8) == 0 && p->ld_an_out % 4 == 0 && p->ld_an_out >= p->C), MISEG_E_BADARG, "mlp_fwd: an_out alignment");
    an.stat = (const double*)p->an.stat; an.styles = p->an.styles; an.eps = p->an.eps; an.out = (bf16*)p->an_out; an.ldo = p->ld_an_out;
    for (int i = 0; i < p->an.num_styles; ++i) { an.gamma[i] = p->an.gamma[i]; an.beta[i] = p->an.beta[i]; }
  }
#define MLP_FWD_LAUNCH_C(ST, AN, CCV)                                                                                                                        \
  do {                                                                                                                                                      \
    MISEG_SET_SMEM((mlp_fwd_kernel<ST, AN, CCV>), lds);                                                                                                     \
    mlp_fwd_kernel<ST, AN, CCV><<<blocks, MLP_FWD_WAVES * 64, lds, s>>>((const bf16*)p->x, p->ldx, (const bf16*)p->w1, p->b1, (const bf16*)p->w2, p->b2,   \
                                                                        (const bf16*)p->res, p->ldres, (bf16*)p->y, p->ldy, p->M, (double*)p->stat, an);     \
  } while (0)
#define MLP_FWD_LAUNCH(ST, AN) do { if (Cc == 48) MLP_FWD_LAUNCH_C(ST, AN, 48); else MLP_FWD_LAUNCH_C(ST, AN, 96); } while (0)
  if (Cc == 96 && blocks > 256) blocks = 256;      // (one 155 KB workgroup per CU)
  if (p->stat) { if (p->an.stat) MLP_FWD_LAUNCH(true, true); else MLP_FWD_LAUNCH(true, false); }
  else { if (p->an.stat) MLP_FWD_LAUNCH(false, true); else MLP_FWD_LAUNCH(false, false); }
#undef MLP_FWD_LAUNCH
#undef MLP_FWD_LAUNCH_C
  MISEG_LAUNCH_CHECK("mlp_fwd");
  return MISEG_OK;
}

extern "C" int miseg_mlp_bwd(const miseg_mlp_params* p, miseg_stream_t s_) {
  if (int rc = mlp_check(p, "mlp_bwd")) return rc;
  MISEG_REQUIRE(p->dy && p->w2t && p->w1t && p->dz && p->h, MISEG_E_BADARG, "mlp_bwd: null pointer");
  MISEG_REQUIRE(((uintptr_t)p->dy % 8) == 0 && p->lddy % 4 == 0 && ((uintptr_t)p->dz % 8) == 0 && p->lddz % 4 == 0 && ((uintptr_t)p->h % 8) == 0 && p->ldh % 4 == 0 &&
                    (!p->dx || (((uintptr_t)p->dx % 8) == 0 && p->lddx % 4 == 0)) && ((uintptr_t)p->w2t % 16) == 0 && ((uintptr_t)p->w1t % 16) == 0,
                MISEG_E_BADARG, "mlp_bwd: alignment");
  const int Cc = p->C, Hh = 4 * Cc, ROW_C = Cc * 2 + 16, ROW_H = Hh * 2 + 16;
  const size_t lds = (size_t)2 * Hh * ROW_C + (Cc == 48 ? (size_t)Cc * ROW_H : 0) + (size_t)(Hh + 2 * Cc) * 4;
  int blocks = cdiv(cdiv(p->M, 16), MLP_BWD_WAVES);
  if (blocks > (Cc == 48 ? 512 : 256)) blocks = Cc == 48 ? 512 : 256;
  MlpBstat bs{};
  if (p->bs_dstat) {
    MISEG_REQUIRE(p->dx && p->bs_x && p->bs_stat && ((uintptr_t)p->bs_x % 8) == 0 && p->ld_bs_x % 4 == 0, MISEG_E_BADARG, "mlp_bwd: norm-backward sums need dx, bs_x (8-byte aligned rows), bs_stat");
    bs.x = (const bf16*)p->bs_x; bs.ldx = p->ld_bs_x; bs.stat_in = (const double*)p->bs_stat; bs.eps = p->bs_eps; bs.dstat = (double*)p->bs_dstat;
#define MLP_BWD_LAUNCH(BS, CCV)                                                                                                                             \
  do {                                                                                                                                                      \
    MISEG_SET_SMEM((mlp_bwd_kernel<BS, CCV>), lds);                                                                                                         \
    mlp_bwd_kernel<BS, CCV><<<blocks, MLP_BWD_WAVES * 64, lds, (hipStream_t)s_>>>((const bf16*)p->x, p->ldx, (const bf16*)p->dy, p->lddy, (const bf16*)p->w1, \
                                                                                p->b1, (const bf16*)p->w2t, (const bf16*)p->w1t, (bf16*)p->dz, p->lddz,      \
                                                                                (bf16*)p->h, p->ldh, (bf16*)p->dx, p->lddx, p->M, bs);                      \
  } while (0)
    if (Cc == 48) MLP_BWD_LAUNCH(true, 48); else MLP_BWD_LAUNCH(true, 96);
  } else {
    if (Cc == 48) MLP_BWD_LAUNCH(false, 48); else MLP_BWD_LAUNCH(false, 96);
  }
#undef MLP_BWD_LAUNCH
  MISEG_LAUNCH_CHECK("mlp_bwd");
  return MISEG_OK;
}
