// Global multi-head self-attention core on the matrix cores (bf16, head_dim 64, <= 256 tokens per sample, no bias / mask / padding):
// the ViT encoder of C-UNETR (BASELINE configs[2]; reference networks/blocks/transformer_block.py:76-110 -> MONAI SABlock:
// softmax((q k^T) * scale) v on "b h (qkv l d)" channels).  Reached through miseg_winattn_fwd / _bwd when the one window covers the whole
// token grid (csrc/attention.hip dispatches here); other shapes keep the one-lane-per-query VALU kernels.
//
// Everything is v_mfma_f32_16x16x32_bf16 with the transposed score tile S^T = K Q^T (keys on the accumulator rows, the query on the lane's
// column), so the softmax statistics of a query are lane-local + two cross-lane steps, and the exponentiated tile is directly the B operand
// of O^T += V^T P^T - the k index of that product is a fixed permutation of the keys that the V^T fragment read mirrors.  Sizes are tiny
// (12 heads x 216 tokens: 143 MFLOP per layer), so the design goal is latency: forward = one workgroup per (sample, head, 64 queries);
// backward = one launch whose workgroups take one of two roles, "dQ" (64 queries) or "dK,dV" (64 keys), each recomputing the scores it
// needs from the saved log-sum-exp - no atomics, no cross-workgroup reduction, bit-reproducible.
#include "common.h"

namespace miseg {

static constexpr int GA_HD = 64;
static constexpr int GA_RS = GA_HD + 8;              // row stride (elements) of the row-major LDS images: 144 B, conflict-free 16-byte reads
static constexpr int GA_MAXN = 256;
static constexpr float GA_LOG2E = 1.4426950408889634f;

struct GaGeom {
  int B, n, NP, heads, C;
  int64_t ldq, ldo, lddo, lddq;
  float scale;
};

__device__ __forceinline__ f32x4 ga_mma(const bf16x8& a, const bf16x8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ bf16x8 ga_zero8() {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
  return v;
}

// rows [NP][GA_RS] <- src[(row0 + r) * ld + col0 .. +63] for r < n, zero rows beyond; 256 threads
__device__ __forceinline__ void ga_stage_rows(bf16* img, const bf16* __restrict__ src, int64_t ld, int n, int NP, int tid) {
  for (int idx = tid; idx < NP * 8; idx += 256) {
    const int r = idx >> 3, g8 = idx & 7;
    bf16x8 v = ga_zero8();
    if (r < n) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)r * ld + 8 * g8);
    *reinterpret_cast<bf16x8*>(img + r * GA_RS + 8 * g8) = v;
  }
}

// dim-major copy [64][TS] (TS = NP + 8) of the same rows
__device__ __forceinline__ void ga_stage_transposed(bf16* img, int TS, const bf16* __restrict__ src, int64_t ld, int n, int NP, int tid) {
  for (int idx = tid; idx < NP * 8; idx += 256) {
    const int r = idx >> 3, g8 = idx & 7;
    bf16x8 v = ga_zero8();
    if (r < n) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)r * ld + 8 * g8);
#pragma unroll
    for (int e = 0; e < 8; ++e) img[(8 * g8 + e) * TS + r] = v[e];
  }
}

// B operand of an accumulator-as-operand product over the token pair (tile 2u, tile 2u + 1): slot j < 4 = row 4 kg + j of tile 2u, j >= 4 = of tile 2u + 1
__device__ __forceinline__ bf16x8 ga_pack(const f32x4& lo, const f32x4& hi) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = (bf16)lo[j]; v[4 + j] = (bf16)hi[j]; }
  return v;
}

// matching A operand from a dim-major image: row = dim, tokens 32 u + 4 kg .. +3 and 32 u + 16 + 4 kg .. +3
__device__ __forceinline__ bf16x8 ga_tfrag(const bf16* timg, int TS, int dim, int u, int kg) {
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(timg + dim * TS + 32 * u + 4 * kg);
  const bf16x4 b = *reinterpret_cast<const bf16x4*>(timg + dim * TS + 32 * u + 16 + 4 * kg);
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
  return v;
}

__device__ __forceinline__ float ga_colreduce_max(float v) {   // over the four lanes (kg = 0..3) that share a column
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float ga_colreduce_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------------------------------ forward
// grid (ceil(n / 64), heads, B), 256 threads.  LDS: K row-major [NP][72], V^T dim-major [64][NP + 8].
// DROP (round 4): dropout on the attention probabilities (MONAI SABlock.drop_weights, reference transformer_block.py:59) - miseg_dropout's mask
// over [samples * heads * n][n], re-created from its key: an accumulator lane holds the 4 consecutive keys 16 t + 4 kg .. + 3 of its query,
// i.e. exactly one 64-bit hash per tile; the softmax denominator is taken BEFORE the mask, O^T += V^T (P o M s)^T.
template <int NT, bool DROP>     // NT = NP / 16 (even): the score tiles stay in registers, so the count is a compile-time constant
__global__ void __launch_bounds__(256) gattn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, float* __restrict__ lse, GaGeom g, AttnDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char ga_lds[];
  const int TS = g.NP + 8;
  bf16* Kimg = reinterpret_cast<bf16*>(ga_lds);
  bf16* Vt = Kimg + g.NP * GA_RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const bf16* base = qkv + (int64_t)b * g.n * g.ldq + h * GA_HD;
  ga_stage_rows(Kimg, base + g.C, g.ldq, g.n, g.NP, tid);
  ga_stage_transposed(Vt, TS, base + 2 * g.C, g.ldq, g.n, g.NP, tid);
  const int fi = lane & 15, kg = lane >> 4;
  const int q = blockIdx.x * 64 + wave * 16 + fi;
  bf16x8 qb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qb[ks] = q < g.n ? *reinterpret_cast<const bf16x8*>(base + (int64_t)q * g.ldq + 32 * ks + 8 * kg) : ga_zero8();
  __syncthreads();
  if (blockIdx.x * 64 + wave * 16 >= g.n) return;         // a whole wave of padding queries (after the only barrier)
  const float c = g.scale * GA_LOG2E;
  f32x4 st[NT];
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc = ga_mma(*reinterpret_cast<const bf16x8*>(Kimg + (16 * t + fi) * GA_RS + 32 * ks + 8 * kg), qb[ks], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[r] = (16 * t + 4 * kg + r < g.n) ? acc[r] * c : -INFINITY;      // padded keys take no part
      m = fmaxf(m, acc[r]);
    }
    st[t] = acc;
  }
  m = ga_colreduce_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      st[t][r] = exp2f(st[t][r] - m);
      l += st[t][r];
    }
  l = ga_colreduce_sum(l);
  if (DROP) {
    const uint64_t k0 = dropout_step_key(dr.key, dr.step_dev);
    const int64_t row = ((int64_t)b * g.heads + h) * g.n + q;
    const int dcg = (g.n + 3) / 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint64_t hh = dropout_group_hash(k0, row, dcg, 16 * t + 4 * kg);
#pragma unroll
      for (int r = 0; r < 4; ++r) st[t][r] = dropout_keeps(hh, r, dr.thresh) ? st[t][r] * dr.scale : 0.f;
    }
  }
  f32x4 o[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) o[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    const bf16x8 pb = ga_pack(st[2 * u], st[2 * u + 1]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) o[mt] = ga_mma(ga_tfrag(Vt, TS, 16 * mt + fi, u, kg), pb, o[mt]);
  }
  if (q < g.n) {
    const float inv = 1.f / l;
    bf16* orow = out + ((int64_t)b * g.n + q) * g.ldo + h * GA_HD;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      bf16x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (bf16)(o[mt][r] * inv);
      *reinterpret_cast<bf16x4*>(orow + 16 * mt + 4 * kg) = v;
    }
    if (kg == 0) lse[((int64_t)b * g.heads + h) * g.n + q] = (m + log2f(l)) * 0.6931471805599453f;   // natural-log units, like every forward kernel: any backward kernel may consume it
  }
}

// ------------------------------------------------------------------------------------------------------------------ backward
// grid (2 * ceil(n / 64), heads, B), 256 threads.  blockIdx.x < nqt: role dQ for queries 64 blockIdx.x ..; else role dK,dV for keys.
//   dQ  : LDS K, V row-major + K^T dim-major.                      dQ = scale * dS K
//   dKV : LDS Q, dO row-major + Q^T, dO^T dim-major + lse, delta.    dV = P^T dO, dK = scale * dS^T Q
// with P = exp2(S c - lse), dS = P o (dO V^T - delta), delta_q = sum_d dO[q][d] O[q][d].
// DROP: dS = P o (M s o dP - delta) with delta from the dropped output, dV = (P o M s)^T dO.  Role dQ holds 4 keys of one query (one hash per
// tile), role dK,dV one key of 4 queries (four hashes per tile).
template <int NT, bool DROP>
__global__ void __launch_bounds__(256) gattn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out, const bf16* __restrict__ dout,
                                                        const float* __restrict__ lse, bf16* __restrict__ dqkv, GaGeom g, int nqt, AttnDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char ga_lds[];
  const int TS = g.NP + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int fi = lane & 15, kg = lane >> 4;
  const float c = g.scale * GA_LOG2E;
  const bf16* base = qkv + (int64_t)b * g.n * g.ldq + h * GA_HD;
  const bf16* obase = out + (int64_t)b * g.n * g.ldo + h * GA_HD;
  const bf16* dobase = dout + (int64_t)b * g.n * g.lddo + h * GA_HD;
  bf16* dbase = dqkv + (int64_t)b * g.n * g.lddq + h * GA_HD;
  const float* lrow = lse + ((int64_t)b * g.heads + h) * g.n;
  const uint64_t dk0 = DROP ? dropout_step_key(dr.key, dr.step_dev) : 0ull;
  const int64_t drow0 = ((int64_t)b * g.heads + h) * g.n;      // first row of this (sample, head) in the [samples * heads * n][n] mask
  const int dcg = (g.n + 3) / 4;
  if ((int)blockIdx.x < nqt) {
    // ------------------------------------------------------------------ role dQ
    bf16* Kimg = reinterpret_cast<bf16*>(ga_lds);
    bf16* Vimg = Kimg + g.NP * GA_RS;
    bf16* Kt = Vimg + g.NP * GA_RS;
    ga_stage_rows(Kimg, base + g.C, g.ldq, g.n, g.NP, tid);
    ga_stage_rows(Vimg, base + 2 * g.C, g.ldq, g.n, g.NP, tid);
    ga_stage_transposed(Kt, TS, base + g.C, g.ldq, g.n, g.NP, tid);
    const int q = blockIdx.x * 64 + wave * 16 + fi;
    const bool qv = q < g.n;
    bf16x8 qb[2], dob[2];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qb[ks] = qv ? *reinterpret_cast<const bf16x8*>(base + (int64_t)q * g.ldq + 32 * ks + 8 * kg) : ga_zero8();
      dob[ks] = qv ? *reinterpret_cast<const bf16x8*>(dobase + (int64_t)q * g.lddo + 32 * ks + 8 * kg) : ga_zero8();
      const bf16x8 ov = qv ? *reinterpret_cast<const bf16x8*>(obase + (int64_t)q * g.ldo + 32 * ks + 8 * kg) : ga_zero8();
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += (float)dob[ks][e] * (float)ov[e];
    }
    delta = ga_colreduce_sum(delta);
    const float lq = qv ? lrow[q] * GA_LOG2E : INFINITY;
    __syncthreads();
    if (blockIdx.x * 64 + wave * 16 >= g.n) return;
    f32x4 dq[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) dq[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
      f32x4 ds[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * u + half;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = ga_mma(*reinterpret_cast<const bf16x8*>(Kimg + (16 * t + fi) * GA_RS + 32 * ks + 8 * kg), qb[ks], s);
          dp = ga_mma(*reinterpret_cast<const bf16x8*>(Vimg + (16 * t + fi) * GA_RS + 32 * ks + 8 * kg), dob[ks], dp);
        }
        const uint64_t hh = DROP ? dropout_group_hash(dk0, drow0 + q, dcg, 16 * t + 4 * kg) : 0ull;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = (16 * t + 4 * kg + r < g.n) ? exp2f(s[r] * c - lq) : 0.f;
          if (DROP) {
            const float pm = dropout_keeps(hh, r, dr.thresh) ? p * dr.scale : 0.f;
            ds[half][r] = fmaf(pm, dp[r], -p * delta);
          } else {
            ds[half][r] = p * (dp[r] - delta);
          }
        }
      }
      const bf16x8 dsb = ga_pack(ds[0], ds[1]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) dq[mt] = ga_mma(ga_tfrag(Kt, TS, 16 * mt + fi, u, kg), dsb, dq[mt]);
    }
    if (qv) {
      bf16* drow = dbase + (int64_t)q * g.lddq;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (bf16)(dq[mt][r] * g.scale);
        *reinterpret_cast<bf16x4*>(drow + 16 * mt + 4 * kg) = v;
      }
    }
    return;
  }
  // -------------------------------------------------------------------- role dK, dV
  bf16* Qimg = reinterpret_cast<bf16*>(ga_lds);
  bf16* dOimg = Qimg + g.NP * GA_RS;
  bf16* Qt = dOimg + g.NP * GA_RS;
  bf16* dOt = Qt + GA_HD * TS;
  float* lsh = reinterpret_cast<float*>(dOt + GA_HD * TS);
  float* dsh = lsh + g.NP;
  ga_stage_rows(Qimg, base, g.ldq, g.n, g.NP, tid);
  ga_stage_transposed(Qt, TS, base, g.ldq, g.n, g.NP, tid);
  ga_stage_transposed(dOt, TS, dobase, g.lddo, g.n, g.NP, tid);
  // dO rows + delta: 8 consecutive lanes share a row
  for (int idx = tid; idx < g.NP * 8; idx += 256) {
    const int r = idx >> 3, g8 = idx & 7;
    bf16x8 v = ga_zero8(), ov = ga_zero8();
    if (r < g.n) {
      v = *reinterpret_cast<const bf16x8*>(dobase + (int64_t)r * g.lddo + 8 * g8);
      ov = *reinterpret_cast<const bf16x8*>(obase + (int64_t)r * g.ldo + 8 * g8);
    }
    *reinterpret_cast<bf16x8*>(dOimg + r * GA_RS + 8 * g8) = v;
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += (float)v[e] * (float)ov[e];
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if (g8 == 0) {
      dsh[r] = d;
      lsh[r] = r < g.n ? lrow[r] * GA_LOG2E : INFINITY;      // a padded query row: p = exp2(-inf) = 0
    }
  }
  const int k0 = ((int)blockIdx.x - nqt) * 64 + wave * 16;
  const int key = k0 + fi;
  const bool kv = key < g.n;
  bf16x8 kb[2], vb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    kb[ks] = kv ? *reinterpret_cast<const bf16x8*>(base + g.C + (int64_t)key * g.ldq + 32 * ks + 8 * kg) : ga_zero8();
    vb[ks] = kv ? *reinterpret_cast<const bf16x8*>(base + 2 * g.C + (int64_t)key * g.ldq + 32 * ks + 8 * kg) : ga_zero8();
  }
  __syncthreads();
  if (k0 >= g.n) return;
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) { dk[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    f32x4 pp[2], ds[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int t = 2 * u + half;
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        s = ga_mma(*reinterpret_cast<const bf16x8*>(Qimg + (16 * t + fi) * GA_RS + 32 * ks + 8 * kg), kb[ks], s);
        dp = ga_mma(*reinterpret_cast<const bf16x8*>(dOimg + (16 * t + fi) * GA_RS + 32 * ks + 8 * kg), vb[ks], dp);
      }
      const f32x4 lq = *reinterpret_cast<const f32x4*>(lsh + 16 * t + 4 * kg);
      const f32x4 dl = *reinterpret_cast<const f32x4*>(dsh + 16 * t + 4 * kg);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(s[r] * c - lq[r]);
        if (DROP) {
          const uint64_t hh = dropout_group_hash(dk0, drow0 + 16 * t + 4 * kg + r, dcg, key);
          const float pm = dropout_keeps(hh, key, dr.thresh) ? p * dr.scale : 0.f;
          pp[half][r] = pm;
          ds[half][r] = fmaf(pm, dp[r], -p * dl[r]);
        } else {
          pp[half][r] = p;
          ds[half][r] = p * (dp[r] - dl[r]);
        }
      }
    }
    const bf16x8 pb = ga_pack(pp[0], pp[1]), dsb = ga_pack(ds[0], ds[1]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      dv[mt] = ga_mma(ga_tfrag(dOt, TS, 16 * mt + fi, u, kg), pb, dv[mt]);
      dk[mt] = ga_mma(ga_tfrag(Qt, TS, 16 * mt + fi, u, kg), dsb, dk[mt]);
    }
  }
  if (kv) {
    bf16* drow = dbase + (int64_t)key * g.lddq;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      bf16x4 a, v;
#pragma unroll
      for (int r = 0; r < 4; ++r) { a[r] = (bf16)(dk[mt][r] * g.scale); v[r] = (bf16)dv[mt][r]; }
      *reinterpret_cast<bf16x4*>(drow + g.C + 16 * mt + 4 * kg) = a;
      *reinterpret_cast<bf16x4*>(drow + 2 * g.C + 16 * mt + 4 * kg) = v;
    }
  }
}

static bool ga_geom(const miseg_winattn_params* p, GaGeom* g) {
  const int n = p->D * p->H * p->W;
  if (p->dtype != MISEG_BF16 || p->bias_table || p->heads <= 0 || p->C != p->heads * GA_HD) return false;
  if (p->wd != p->D || p->wh != p->H || p->ww != p->W || (p->sd | p->sh | p->sw) != 0 || n > GA_MAXN || n < 1) return false;
  if (p->ldq % 8 || p->ldo % 4 || ((uintptr_t)p->qkv & 15) || ((uintptr_t)p->out & 7)) return false;
  g->B = p->B; g->n = n; g->NP = (n + 31) / 32 * 32; g->heads = p->heads; g->C = p->C; g->ldq = p->ldq; g->ldo = p->ldo; g->lddo = 0; g->lddq = 0; g->scale = p->scale;
  return true;
}

bool global_attn_takes(const miseg_winattn_params* p) {
  GaGeom g;
  return ga_geom(p, &g) && g.NP / 16 >= 2 && g.NP / 16 <= 16;
}

// 1: handled (rc in *rc), 0: not this kernel's shape
int global_attn_fwd(const miseg_winattn_params* p, const AttnDrop& dr, hipStream_t s, int* rc) {
  GaGeom g;
  if (!ga_geom(p, &g)) return 0;
  const size_t sh = (size_t)(g.NP * GA_RS + GA_HD * (g.NP + 8)) * 2;
  const dim3 grid(cdiv(g.n, 64), g.heads, g.B);
#define GA_FWD_D(NT, D) do { MISEG_SET_SMEM((gattn_fwd_kernel<NT, D>), sh); \
    gattn_fwd_kernel<NT, D><<<grid, 256, sh, s>>>((const bf16*)p->qkv, (bf16*)p->out, p->lse, g, dr); } while (0)
#define GA_FWD(NT) do { if (dr.thresh) GA_FWD_D(NT, true); else GA_FWD_D(NT, false); } while (0)
  switch (g.NP / 16) {
    case 2: GA_FWD(2); break; case 4: GA_FWD(4); break; case 6: GA_FWD(6); break; case 8: GA_FWD(8); break;
    case 10: GA_FWD(10); break; case 12: GA_FWD(12); break; case 14: GA_FWD(14); break; case 16: GA_FWD(16); break;
    default: return 0;
  }
#undef GA_FWD
#undef GA_FWD_D
  hipError_t e = hipGetLastError();
  *rc = e == hipSuccess ? MISEG_OK : set_error(MISEG_E_LAUNCH, "gattn_fwd: %s", hipGetErrorString(e));
  return 1;
}

int global_attn_bwd(const miseg_winattn_bwd_params* p, const AttnDrop& dr, hipStream_t s, int* rc) {
  GaGeom g;
  if (!ga_geom(&p->f, &g)) return 0;
  if (p->lddo % 8 || p->lddq % 4 || p->f.ldo % 8 || ((uintptr_t)p->dout & 15) || ((uintptr_t)p->f.out & 15) || ((uintptr_t)p->dqkv & 7)) return 0;
  g.lddo = p->lddo; g.lddq = p->lddq;
  const int nqt = cdiv(g.n, 64);
  const size_t sh_q = (size_t)(2 * g.NP * GA_RS + GA_HD * (g.NP + 8)) * 2;
  const size_t sh_k = (size_t)(2 * g.NP * GA_RS + 2 * GA_HD * (g.NP + 8)) * 2 + (size_t)2 * g.NP * 4;
  const size_t sh = sh_q > sh_k ? sh_q : sh_k;
  const dim3 grid(2 * nqt, g.heads, g.B);
#define GA_BWD_D(NT, D) do { MISEG_SET_SMEM((gattn_bwd_kernel<NT, D>), sh); \
    gattn_bwd_kernel<NT, D><<<grid, 256, sh, s>>>((const bf16*)p->f.qkv, (const bf16*)p->f.out, (const bf16*)p->dout, p->f.lse, (bf16*)p->dqkv, g, nqt, dr); } while (0)
#define GA_BWD(NT) do { if (dr.thresh) GA_BWD_D(NT, true); else GA_BWD_D(NT, false); } while (0)
  switch (g.NP / 16) {
    case 2: GA_BWD(2); break; case 4: GA_BWD(4); break; case 6: GA_BWD(6); break; case 8: GA_BWD(8); break;
    case 10: GA_BWD(10); break; case 12: GA_BWD(12); break; case 14: GA_BWD(14); break; case 16: GA_BWD(16); break;
    default: return 0;
  }
#undef GA_BWD
#undef GA_BWD_D
  hipError_t e = hipGetLastError();
  *rc = e == hipSuccess ? MISEG_OK : set_error(MISEG_E_LAUNCH, "gattn_bwd: %s", hipGetErrorString(e));
  return 1;
}

}  // namespace miseg
