// (Conditional) InstanceNorm and LayerNorm over channels-last rows -- HBM-bound kernels.
// Roofline: bytes only.  stats = 1 read of x; apply = 1 read (+1 residual) + 1 write;
// backward = reduce (dy, y, x reads) + apply (dy, y, x reads, dx [+dres] writes).
// Thread layout: tx owns one 16-byte channel vector for the whole block (its affine/statistics live in
// registers), ty walks rows, so every wave-instruction reads contiguous row segments.
#include <cstdlib>
#include "common.h"

namespace miseg {

#ifdef MISEG_NORM_STAMPS      // debug build (scripts/debug/norm_stamps.py): where a small norm launch spends its time, seen by thread 0 of workgroup 0
__device__ unsigned long long g_norm_stamps[16];
#define NSTAMP(i)                                                                                   \
  do {                                                                                              \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {                \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                   \
      g_norm_stamps[i] = __builtin_readcyclecounter();                                              \
    }                                                                                               \
  } while (0)
#else
#define NSTAMP(i)
#endif

static constexpr int NORM_THREADS = 256;
static constexpr int NORM_TX_MAX = 32;   // channel vectors per block (wider rows are tiled over blockIdx.z)

struct NormGeom {
  int vec;      // elements per lane access (Vec16<T>::N or 1)
  int cv;       // channel vectors per row
  int tx, ty;   // thread grid inside a block
  int rpb;      // rows per block
  int chunks;   // row chunks per sample
  int ctiles;   // channel tiles
};

static NormGeom norm_geom(int S, int C, bool vec_ok, int vecN, int target_blocks = 1024) {
  NormGeom g;
  g.vec = (vec_ok && C % vecN == 0) ? vecN : 1;
  g.cv = C / g.vec;
  g.tx = g.cv < NORM_TX_MAX ? g.cv : NORM_TX_MAX;
  g.ty = NORM_THREADS / g.tx;
  g.ctiles = cdiv(g.cv, g.tx);
  // enough workgroups to fill 256 CUs even on the small grids of the deep stages, >= 4 rows per lane
  // ~4 workgroups per CU for the streaming kernels (one per CU ran the 48^3 tensors at half the HBM rate); the statistics kernel
  // asks for 256 (fewer fp64 atomics)
  int rpb = cdiv(S, target_blocks);
  if (rpb < 4 * g.ty) rpb = 4 * g.ty;
  if (rpb > 1024) rpb = 1024;
  g.rpb = rpb;
  g.chunks = cdiv(S, rpb);
  return g;
}

template <class T, int VEC> struct RowVec {
  float v[VEC];
  __device__ __forceinline__ void load(const T* p) {
    if constexpr (VEC == 1) {
      v[0] = to_f32(p[0]);
    } else {
      typename Vec16<T>::type t = *reinterpret_cast<const typename Vec16<T>::type*>(p);
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = to_f32(t[i]);
    }
  }
  __device__ __forceinline__ void store(T* p) const {
    if constexpr (VEC == 1) {
      p[0] = from_f32<T>(v[0]);
    } else {
      typename Vec16<T>::type t;
#pragma unroll
      for (int i = 0; i < VEC; ++i) t[i] = from_f32<T>(v[i]);
      *reinterpret_cast<typename Vec16<T>::type*>(p) = t;
    }
  }
};

// Statistics buffers hold NORM_R replicas of the fp64 (sum, sum of squares) pairs: double [NORM_R][B][C][2].  A block adds
// its partial sums to replica (chunk mod NORM_R): with one row, the ~900 blocks of a 96^3 tensor queue on 96 addresses and
// the reduction costs 2x the streaming read (measured 49 us vs 15 us); consumers add the replicas up in their prologue.
static constexpr int NORM_R = 16;

// block-level reduction over ty of two per-thread vectors, then one fp64 atomic per (channel, which) -- no finalize
// kernel, no workspace.
template <int VEC>
__device__ __forceinline__ void block_reduce_to_stat(float* red, const float* s, const float* q, int tx, int ty, int tx_n, int ty_n, int c0, int C,
                                                     double* stat_b) {
  if (ty < ty_n) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      red[(ty * 2 + 0) * tx_n * VEC + tx * VEC + i] = s[i];
      red[(ty * 2 + 1) * tx_n * VEC + tx * VEC + i] = q[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * tx_n * VEC; e += NORM_THREADS) {
    const int which = e / (tx_n * VEC), col = e % (tx_n * VEC);
    float acc = 0.f;
    for (int y = 0; y < ty_n; ++y) acc += red[(y * 2 + which) * tx_n * VEC + col];
    const int ch = c0 * VEC + col;
    if (ch < C) atomicAdd(stat_b + 2 * ch + which, (double)acc);
  }
}

// workgroup prologue of the consumers: sums[(col) * 2 + which] = sum over replicas of stat[r][b][c0*VEC + col][which]
__device__ __forceinline__ void gather_stat(double* sums, const double* __restrict__ stat, int64_t rstride, int b, int C, int ch0, int ncols) {
  for (int e = threadIdx.x; e < 2 * ncols; e += NORM_THREADS) {
    const int ch = ch0 + (e >> 1);
    double acc = 0.0;
    if (ch < C) {
      const double* p = stat + ((int64_t)b * C + ch) * 2 + (e & 1);
#pragma unroll
      for (int r = 0; r < NORM_R; ++r) acc += p[r * rstride];
    }
    sums[e] = acc;
  }
  __syncthreads();
}

// mean and 1/sqrt(var + eps) of one channel; the variance is formed in fp64 (cancellation), the root in fp32
__device__ __forceinline__ void mean_rstd(const double* sums2, double invS, float eps, float& m, float& rs) {
  const double mu = sums2[0] * invS;
  double var = fma(sums2[1], invS, -mu * mu);
  if (var < 0.0) var = 0.0;
  m = (float)mu;
  rs = 1.0f / sqrtf((float)var + eps);
}

// ---------------------------------------------------------------------------------------------------
// stats: stat[b][c] += (sum x, sum x^2) over this block's rows      grid (chunks, B, ctiles)
// ---------------------------------------------------------------------------------------------------
template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_stats_kernel(const T* __restrict__ x, int64_t ldx, int S, int C, int cv, int tx_n, int ty_n, int rpb,
                                                                      double* __restrict__ stat) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int b = blockIdx.y, chunk = blockIdx.x, c0 = blockIdx.z * tx_n;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int r0 = chunk * rpb, r1 = min(S, r0 + rpb);
  const T* xb = x + (int64_t)b * S * ldx;
  const int c = c0 + tx;
  float s[VEC], q[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = q[i] = 0.f;
  if (ty < ty_n && c < cv) {
    int r = r0 + ty;
    for (; r + 3 * ty_n < r1; r += 4 * ty_n) {   // 4 independent row loads in flight per lane
      RowVec<T, VEC> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u].load(xb + (int64_t)(r + u * ty_n) * ldx + c * VEC);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s[i] += v[u].v[i]; q[i] = fmaf(v[u].v[i], v[u].v[i], q[i]); }
    }
    for (; r < r1; r += ty_n) {
      RowVec<T, VEC> v;
      v.load(xb + (int64_t)r * ldx + c * VEC);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { s[i] += v.v[i]; q[i] = fmaf(v.v[i], v.v[i], q[i]); }
    }
  }
  block_reduce_to_stat<VEC>(red, s, q, tx, ty, tx_n, ty_n, c0, C, stat + ((int64_t)(chunk % NORM_R) * gridDim.y + b) * C * 2);
}

// The second launch of a convolution split over its input chunks (conv3d.hip: the small grids of the deep stages): y = round(sum of the
// fp32 slabs [+ res]) and, in the same pass, the instance-norm statistics of y - the split path used to leave them to a separate
// miseg_instnorm_stats launch over the tensor it had just written (one launch-and-drain latency per convolution of stages >= 3).
// Same thread geometry and replicated fp64 reduction as instnorm_stats_kernel; vector lanes read VEC floats of every slab.
template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) slabs_to_out_stats_kernel(const float* __restrict__ slabs, int nslabs, int64_t slab_stride, T* __restrict__ y,
                                                                          int64_t ldy, const T* __restrict__ res, int64_t ldres, int S, int C, int cv, int tx_n,
                                                                          int ty_n, int rpb, double* __restrict__ stat) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int b = blockIdx.y, chunk = blockIdx.x, c0 = blockIdx.z * tx_n;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int r0 = chunk * rpb, r1 = min(S, r0 + rpb);
  const int c = c0 + tx;
  float s[VEC], q[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = q[i] = 0.f;
  if (ty < ty_n && c < cv) {
    for (int r = r0 + ty; r < r1; r += ty_n) {
      const int64_t row = (int64_t)b * S + r;
      float a[VEC];
      const float* p = slabs + row * C + c * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) a[i] = 0.f;
      if constexpr (VEC % 4 == 0) {
        for (int z0 = 0; z0 < nslabs; z0 += 4) {      // four slabs' loads in flight together (clamped: a repeated slab is not added)
          f32x4 v[4][VEC / 4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int z = min(z0 + u, nslabs - 1);
#pragma unroll
            for (int k = 0; k < VEC / 4; ++k) v[u][k] = *reinterpret_cast<const f32x4*>(p + z * slab_stride + 4 * k);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (z0 + u < nslabs) {
#pragma unroll
              for (int k = 0; k < VEC / 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[4 * k + e] += v[u][k][e];
            }
          }
        }
      } else {
        for (int z = 0; z < nslabs; ++z)
#pragma unroll
          for (int i = 0; i < VEC; ++i) a[i] += p[z * slab_stride + i];
      }
      RowVec<T, VEC> o;
      if (res) {
        o.load(res + row * ldres + c * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) a[i] += o.v[i];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        o.v[i] = to_f32(from_f32<T>(a[i]));      // statistics of the stored (rounded) values, as miseg_instnorm_stats would read them
        s[i] += o.v[i];
        q[i] = fmaf(o.v[i], o.v[i], q[i]);
      }
      o.store(y + row * ldy + c * VEC);
    }
  }
  if (stat) block_reduce_to_stat<VEC>(red, s, q, tx, ty, tx_n, ty_n, c0, C, stat + ((int64_t)(chunk % NORM_R) * gridDim.y + b) * C * 2);
}

struct StylePtrs {
  const float* gamma[MISEG_MAX_STYLES];
  const float* beta[MISEG_MAX_STYLES];
};

template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_apply_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ res, int64_t ldres,
                                                                      T* __restrict__ y, int64_t ldy, int S, int C, int cv, int tx_n, int ty_n, int rpb,
                                                                      const double* __restrict__ stat, float eps, const int32_t* __restrict__ styles,
                                                                      StylePtrs sp, int act, float slope, const double* __restrict__ rstat, StylePtrs rsp,
                                                                      const T* __restrict__ r1x, int64_t ldr1x, const T* __restrict__ r1w) {
  // rstat != nullptr: `res` is the RAW input of a second (shortcut) instance norm with its own statistics / affine rows and is
  // normalised on the fly: y = act(norm(x) + norm_r(res)) in one pass (UnetResBlock with a 1x1x1 shortcut conv, dynunet_block.py:118-124)
  // r1x != nullptr (with rstat, res == nullptr): that raw input is the 1x1x1 convolution of a ONE-channel image (the stem block) and is
  // not stored at all: res[row][c] = round(r1x[row] * r1w[c]), formed here exactly as the rank-1 GEMM would have rounded it
  extern __shared__ __attribute__((aligned(16))) double sums[];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int c = blockIdx.z * tx_n + tx;
  double* rsums = sums + 2 * tx_n * VEC;
  gather_stat(sums, stat, (int64_t)gridDim.y * C * 2, b, C, blockIdx.z * tx_n * VEC, tx_n * VEC);
  if (rstat) gather_stat(rsums, rstat, (int64_t)gridDim.y * C * 2, b, C, blockIdx.z * tx_n * VEC, tx_n * VEC);
  if (ty >= ty_n || c >= cv) return;
  const int r0 = chunk * rpb, r1 = min(S, r0 + rpb);
  const int st = styles ? styles[b] : 0;
  const float* g = sp.gamma[st];
  const float* be = sp.beta[st];
  const int64_t boff = (int64_t)b * S;
  const double invS = 1.0 / S;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = c * VEC + i;
    float m, rs;
    mean_rstd(sums + (tx * VEC + i) * 2, invS, eps, m, rs);
    const float gg = g ? g[ch] : 1.f, bb = be ? be[ch] : 0.f;
    sc[i] = rs * gg;
    sh[i] = bb - m * sc[i];
  }
  float rsc[VEC], rsh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { rsc[i] = 1.f; rsh[i] = 0.f; }
  if (rstat) {
    const float* rg = rsp.gamma[st];
    const float* rb = rsp.beta[st];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int ch = c * VEC + i;
      float m, rs;
      mean_rstd(rsums + (tx * VEC + i) * 2, invS, eps, m, rs);
      rsc[i] = rs * (rg ? rg[ch] : 1.f);
      rsh[i] = (rb ? rb[ch] : 0.f) - m * rsc[i];
    }
  }
  float r1wv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) r1wv[i] = r1x ? to_f32(r1w[c * VEC + i]) : 0.f;
#pragma unroll 4
  for (int r = r0 + ty; r < r1; r += ty_n) {
    RowVec<T, VEC> v, o;
    v.load(x + (boff + r) * ldx + c * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.v[i] = fmaf(v.v[i], sc[i], sh[i]);
    if (res || r1x) {
      RowVec<T, VEC> rr;
      if (res) {
        rr.load(res + (boff + r) * ldres + c * VEC);
      } else {
        const float xs = to_f32(r1x[(boff + r) * ldr1x]);
#pragma unroll
        for (int i = 0; i < VEC; ++i) rr.v[i] = to_f32(from_f32<T>(xs * r1wv[i]));
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.v[i] += fmaf(rr.v[i], rsc[i], rsh[i]);
    }
    if (act == MISEG_ACT_LEAKY) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.v[i] = o.v[i] > 0.f ? o.v[i] : o.v[i] * slope;
    }
    o.store(y + (boff + r) * ldy + c * VEC);
  }
}

// ---------------------------------------------------------------------------------------------------
// backward: reduce (sum dz, sum dz*xhat) -> dstat (fp64 atomics) ; apply (its chunk-0 workgroups also add the affine gradients)
// ---------------------------------------------------------------------------------------------------
template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_bwd_reduce_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ yact, int64_t ldy,
                                                                           const T* __restrict__ x, int64_t ldx, int S, int C, int cv, int tx_n, int ty_n, int rpb,
                                                                           const double* __restrict__ stat, float eps, int act, float slope,
                                                                           double* __restrict__ dstat, const int32_t* __restrict__ styles, StylePtrs sp) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  double* sums = reinterpret_cast<double*>(red);   // prologue only; the reduction reuses the space after a barrier
  const int b = blockIdx.y, chunk = blockIdx.x, c0 = blockIdx.z * tx_n;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int r0 = chunk * rpb, r1 = min(S, r0 + rpb);
  const int64_t boff = (int64_t)b * S;
  const int c = c0 + tx;
  gather_stat(sums, stat, (int64_t)gridDim.y * C * 2, b, C, c0 * VEC, tx_n * VEC);
  float s[VEC], q[VEC], m[VEC], rs[VEC];
  const double invS = 1.0 / S;
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s[i] = q[i] = 0.f; mean_rstd(sums + (tx * VEC + i) * 2, invS, eps, m[i], rs[i]); }
  float zsc[VEC], zsh[VEC];     // the forward's scale / shift (instnorm_apply_kernel): only to recompute the activation's sign
  {
    const int st = styles ? styles[b] : 0;
    const float* gz = sp.gamma[st];
    const float* bz = sp.beta[st];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int ch = c * VEC + i;
      zsc[i] = rs[i] * ((gz && c < cv) ? gz[ch] : 1.f);
      zsh[i] = ((bz && c < cv) ? bz[ch] : 0.f) - m[i] * zsc[i];
    }
  }
  __syncthreads();
  if (ty < ty_n && c < cv) {
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += ty_n) {
      RowVec<T, VEC> g, xv;
      g.load(dy + (boff + r) * lddy + c * VEC);
      xv.load(x + (boff + r) * ldx + c * VEC);
      if (act == MISEG_ACT_LEAKY) {
        if (yact) {
          RowVec<T, VEC> yv;
          yv.load(yact + (boff + r) * ldy + c * VEC);
#pragma unroll
          for (int i = 0; i < VEC; ++i) g.v[i] = yv.v[i] > 0.f ? g.v[i] : g.v[i] * slope;
        } else {   // no residual went into the activation: its sign is the sign of the forward's fma(x, sc, sh), recomputed from x
#pragma unroll
          for (int i = 0; i < VEC; ++i) g.v[i] = fmaf(xv.v[i], zsc[i], zsh[i]) > 0.f ? g.v[i] : g.v[i] * slope;
        }
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        s[i] += g.v[i];
        q[i] = fmaf(g.v[i], (xv.v[i] - m[i]) * rs[i], q[i]);
      }
    }
  }
  block_reduce_to_stat<VEC>(red, s, q, tx, ty, tx_n, ty_n, c0, C, dstat + ((int64_t)(chunk % NORM_R) * gridDim.y + b) * C * 2);
}

struct StyleGradPtrs {
  float* dgamma[MISEG_MAX_STYLES];
  float* dbeta[MISEG_MAX_STYLES];
};

template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_bwd_apply_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ yact, int64_t ldy,
                                                                          const T* __restrict__ x, int64_t ldx, T* __restrict__ dx, int64_t lddx,
                                                                          T* __restrict__ dres, int64_t lddres, int S, int C, int cv, int tx_n, int ty_n, int rpb,
                                                                          const double* __restrict__ stat, float eps, const int32_t* __restrict__ styles,
                                                                          StylePtrs sp, int act, float slope, const double* __restrict__ dstat, StyleGradPtrs gp,
                                                                          const T* __restrict__ gadd, int64_t ldgadd) {
  extern __shared__ __attribute__((aligned(16))) double sums[];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int c = blockIdx.z * tx_n + tx;
  double* dsums = sums + 2 * tx_n * VEC;
  gather_stat(sums, stat, (int64_t)gridDim.y * C * 2, b, C, blockIdx.z * tx_n * VEC, tx_n * VEC);
  gather_stat(dsums, dstat, (int64_t)gridDim.y * C * 2, b, C, blockIdx.z * tx_n * VEC, tx_n * VEC);
  if (chunk == 0) {   // the affine gradients of this (sample, channel tile): dgamma = sum dz*xhat, dbeta = sum dz
    const int st0 = styles ? styles[b] : 0;
    for (int e = threadIdx.x; e < tx_n * VEC; e += NORM_THREADS) {
      const int ch = blockIdx.z * tx_n * VEC + e;
      if (ch < C) {
        if (gp.dgamma[st0]) atomicAdd(gp.dgamma[st0] + ch, (float)dsums[2 * e + 1]);
        if (gp.dbeta[st0]) atomicAdd(gp.dbeta[st0] + ch, (float)dsums[2 * e]);
      }
    }
  }
  if (ty >= ty_n || c >= cv) return;
  const int r0 = chunk * rpb, r1 = min(S, r0 + rpb);
  const int st = styles ? styles[b] : 0;
  const float* g = sp.gamma[st];
  const int64_t boff = (int64_t)b * S;
  const double invS = 1.0 / S;
  float m[VEC], rs[VEC], sc[VEC], a[VEC], bq[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = c * VEC + i;
    mean_rstd(sums + (tx * VEC + i) * 2, invS, eps, m[i], rs[i]);
    sc[i] = rs[i] * (g ? g[ch] : 1.f);
    a[i] = (float)(dsums[(tx * VEC + i) * 2] * invS);
    bq[i] = (float)(dsums[(tx * VEC + i) * 2 + 1] * invS);
  }
  float zsc[VEC], zsh[VEC];
  {
    const float* bz = sp.beta[st];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { zsc[i] = sc[i]; zsh[i] = (bz ? bz[c * VEC + i] : 0.f) - m[i] * sc[i]; }
  }
#pragma unroll 4
  for (int r = r0 + ty; r < r1; r += ty_n) {
    RowVec<T, VEC> gv, xv, o;
    gv.load(dy + (boff + r) * lddy + c * VEC);
    xv.load(x + (boff + r) * ldx + c * VEC);
    if (act == MISEG_ACT_LEAKY) {
      if (yact) {
        RowVec<T, VEC> yv;
        yv.load(yact + (boff + r) * ldy + c * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) gv.v[i] = yv.v[i] > 0.f ? gv.v[i] : gv.v[i] * slope;
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) gv.v[i] = fmaf(xv.v[i], zsc[i], zsh[i]) > 0.f ? gv.v[i] : gv.v[i] * slope;
      }
    }
    if (dres) gv.store(dres + (boff + r) * lddres + c * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float xh = (xv.v[i] - m[i]) * rs[i];
      o.v[i] = sc[i] * (gv.v[i] - a[i] - xh * bq[i]);
    }
    if (gadd) {
      RowVec<T, VEC> ga;
      ga.load(gadd + (boff + r) * ldgadd + c * VEC);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.v[i] += ga.v[i];
    }
    o.store(dx + (boff + r) * lddx + c * VEC);
  }
}

// ---------------------------------------------------------------------------------------------------
// Small tensors (the deep stages: <= 512 rows per sample): statistics and normalisation in ONE launch, one workgroup per
// (sample, tile of tx_n channel vectors) that walks all rows twice (the second pass hits L1/L2).  A launch costs ~4 us
// on this chip whatever it does, so the 2 + 2 launches of forward + backward become 1 + 1.
// ---------------------------------------------------------------------------------------------------
// per-column totals of two per-thread vectors over the ty rows of the workgroup -> tot[(tx * VEC + i) * 2 + which] (double)
template <int VEC>
__device__ __forceinline__ void team_totals(float* red, double* tot, const float* s, const float* q, int tx, int ty, int tx_n, int ty_n) {
  if (ty < ty_n) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      red[(ty * 2 + 0) * tx_n * VEC + tx * VEC + i] = s[i];
      red[(ty * 2 + 1) * tx_n * VEC + tx * VEC + i] = q[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * tx_n * VEC; e += NORM_THREADS) {
    const int which = e / (tx_n * VEC), col = e % (tx_n * VEC);
    double acc = 0.0;
    for (int y = 0; y < ty_n; ++y) acc += (double)red[(y * 2 + which) * tx_n * VEC + col];
    tot[col * 2 + which] = acc;
  }
  __syncthreads();
}

// ---- backward of y = LeakyReLU(norm_a(xa) + norm_b(xb)) (instnorm_apply_kernel with rstat): both norms see the same activation-
// masked gradient g, so ONE reduction pass yields sum g, sum g*xhat_a, sum g*xhat_b and ONE apply pass writes both input gradients
// (10 passes over the tensor instead of the 13 of two chained norm backwards, no `dres` tensor, two launches instead of four).
// the pre-activation of the residual norm pair exactly as instnorm_apply_kernel forms it (same operations in the same order: the
// recomputed sign must be the sign the forward pass saw)
__device__ __forceinline__ float pair_preact(float xa, float sca, float sha, float xb, float scb, float shb) {
  float o = fmaf(xa, sca, sha);
  o += fmaf(xb, scb, shb);
  return o;
}
__device__ __forceinline__ void pair_preact_coeffs(const StylePtrs& spa, const StylePtrs& spb, int st, int ch, float ma, float rsa, float mb, float rsb,
                                                   float& sca, float& sha, float& scb, float& shb) {
  const float* ga = spa.gamma[st];
  const float* ba = spa.beta[st];
  const float* gb = spb.gamma[st];
  const float* bb = spb.beta[st];
  sca = rsa * (ga ? ga[ch] : 1.f);
  sha = (ba ? ba[ch] : 0.f) - ma * sca;
  scb = rsb * (gb ? gb[ch] : 1.f);
  shb = (bb ? bb[ch] : 0.f) - mb * scb;
}

template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_pair_bwd_reduce_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ yact, int64_t ldy,
                                                                                const T* __restrict__ xa, int64_t ldxa, const T* __restrict__ xb, int64_t ldxb,
                                                                                int S, int C, int cv, int tx_n, int ty_n, int rpb,
                                                                                const double* __restrict__ stat_a, const double* __restrict__ stat_b, float eps,
                                                                                float slope, double* __restrict__ dstat_a, double* __restrict__ dstat_b,
                                                                                const int32_t* __restrict__ styles, StylePtrs spa, StylePtrs spb,
                                                                                const T* __restrict__ r1x, int64_t ldr1x, const T* __restrict__ r1w) {
  // r1x != nullptr: xb is not stored - xb[row][c] = round(r1x[row] * r1w[c]) (see instnorm_apply_kernel)
  // yact == nullptr: the LeakyReLU's sign is recomputed from the two inputs with the forward pass's own expression (instnorm_apply_kernel:
  // fmaf(xa, sc_a, sh_a) + fmaf(xb, sc_b, sh_b), identical operands) - one tensor less to read in both passes
  extern __shared__ __attribute__((aligned(16))) float red[];
  double* sums_a = reinterpret_cast<double*>(red);   // prologue only; the reductions reuse the space after a barrier
  double* sums_b = sums_a + 2 * tx_n * VEC;
  const int b = blockIdx.y, chunk = blockIdx.x, c0 = blockIdx.z * tx_n;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int r0 = chunk * rpb, r1 = min(S, r0 + rpb);
  const int64_t boff = (int64_t)b * S;
  const int c = c0 + tx;
  gather_stat(sums_a, stat_a, (int64_t)gridDim.y * C * 2, b, C, c0 * VEC, tx_n * VEC);
  gather_stat(sums_b, stat_b, (int64_t)gridDim.y * C * 2, b, C, c0 * VEC, tx_n * VEC);
  float s[VEC], qa[VEC], qb[VEC], ma[VEC], rsa[VEC], mb[VEC], rsb[VEC], zsa[VEC], zha[VEC], zsb[VEC], zhb[VEC];
  const double invS = 1.0 / S;
  const int st = styles ? styles[b] : 0;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    s[i] = qa[i] = qb[i] = 0.f;
    mean_rstd(sums_a + (tx * VEC + i) * 2, invS, eps, ma[i], rsa[i]);
    mean_rstd(sums_b + (tx * VEC + i) * 2, invS, eps, mb[i], rsb[i]);
    const int ch = min(c * VEC + i, C - 1);
    pair_preact_coeffs(spa, spb, st, ch, ma[i], rsa[i], mb[i], rsb[i], zsa[i], zha[i], zsb[i], zhb[i]);
  }
  float r1wv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) r1wv[i] = r1x ? to_f32(r1w[min(c * VEC + i, C - 1)]) : 0.f;
  __syncthreads();
  if (ty < ty_n && c < cv) {
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += ty_n) {
      RowVec<T, VEC> g, yv, va, vb;
      g.load(dy + (boff + r) * lddy + c * VEC);
      if (yact) yv.load(yact + (boff + r) * ldy + c * VEC);
      va.load(xa + (boff + r) * ldxa + c * VEC);
      if (r1x) {
        const float xs = to_f32(r1x[(boff + r) * ldr1x]);
#pragma unroll
        for (int i = 0; i < VEC; ++i) vb.v[i] = to_f32(from_f32<T>(xs * r1wv[i]));
      } else {
        vb.load(xb + (boff + r) * ldxb + c * VEC);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float pre = yact ? yv.v[i] : pair_preact(va.v[i], zsa[i], zha[i], vb.v[i], zsb[i], zhb[i]);
        const float gi = pre > 0.f ? g.v[i] : g.v[i] * slope;
        s[i] += gi;
        qa[i] = fmaf(gi, (va.v[i] - ma[i]) * rsa[i], qa[i]);
        qb[i] = fmaf(gi, (vb.v[i] - mb[i]) * rsb[i], qb[i]);
      }
    }
  }
  block_reduce_to_stat<VEC>(red, s, qa, tx, ty, tx_n, ty_n, c0, C, dstat_a + ((int64_t)(chunk % NORM_R) * gridDim.y + b) * C * 2);
  __syncthreads();
  block_reduce_to_stat<VEC>(red, s, qb, tx, ty, tx_n, ty_n, c0, C, dstat_b + ((int64_t)(chunk % NORM_R) * gridDim.y + b) * C * 2);
}

template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_pair_bwd_apply_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ yact, int64_t ldy,
                                                                               const T* __restrict__ xa, int64_t ldxa, const T* __restrict__ xb, int64_t ldxb,
                                                                               T* __restrict__ dxa, int64_t lddxa, T* __restrict__ dxb, int64_t lddxb, int S, int C,
                                                                               int cv, int tx_n, int ty_n, int rpb, const double* __restrict__ stat_a,
                                                                               const double* __restrict__ stat_b, float eps, const int32_t* __restrict__ styles,
                                                                               StylePtrs spa, StylePtrs spb, float slope, const double* __restrict__ dstat_a,
                                                                               const double* __restrict__ dstat_b, StyleGradPtrs gpa, StyleGradPtrs gpb,
                                                                               const T* __restrict__ r1x, int64_t ldr1x, const T* __restrict__ r1w,
                                                                               float* __restrict__ r1dw) {
  // r1x != nullptr: xb = round(r1x[row] * r1w[c]) is not stored (see instnorm_apply_kernel) and neither is its gradient: the only consumer
  // is the weight gradient of that 1x1x1 convolution, dW[c] += sum over rows of round(dxb[row][c]) * r1x[row], reduced here
  extern __shared__ __attribute__((aligned(16))) double sums[];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int c = blockIdx.z * tx_n + tx;
  const int nc = 2 * tx_n * VEC;
  double* sa = sums;
  double* sb = sums + nc;
  double* da = sums + 2 * nc;
  double* db = sums + 3 * nc;
  gather_stat(sa, stat_a, (int64_t)gridDim.y * C * 2, b, C, blockIdx.z * tx_n * VEC, tx_n * VEC);
  gather_stat(sb, stat_b, (int64_t)gridDim.y * C * 2, b, C, blockIdx.z * tx_n * VEC, tx_n * VEC);
  gather_stat(da, dstat_a, (int64_t)gridDim.y * C * 2, b, C, blockIdx.z * tx_n * VEC, tx_n * VEC);
  gather_stat(db, dstat_b, (int64_t)gridDim.y * C * 2, b, C, blockIdx.z * tx_n * VEC, tx_n * VEC);
  const int st = styles ? styles[b] : 0;
  if (chunk == 0) {   // the affine gradients of this (sample, channel tile), both norms
    for (int e = threadIdx.x; e < tx_n * VEC; e += NORM_THREADS) {
      const int ch = blockIdx.z * tx_n * VEC + e;
      if (ch < C) {
        if (gpa.dgamma[st]) atomicAdd(gpa.dgamma[st] + ch, (float)da[2 * e + 1]);
        if (gpa.dbeta[st]) atomicAdd(gpa.dbeta[st] + ch, (float)da[2 * e]);
        if (gpb.dgamma[st]) atomicAdd(gpb.dgamma[st] + ch, (float)db[2 * e + 1]);
        if (gpb.dbeta[st]) atomicAdd(gpb.dbeta[st] + ch, (float)db[2 * e]);
      }
    }
  }
  const bool live = ty < ty_n && c < cv;
  if (!live && !r1x) return;
  float dwacc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) dwacc[i] = 0.f;
  if (live) {
  const int r0 = chunk * rpb, r1 = min(S, r0 + rpb);
  const float* ga = spa.gamma[st];
  const float* gb = spb.gamma[st];
  const int64_t boff = (int64_t)b * S;
  const double invS = 1.0 / S;
  float ma[VEC], rsa[VEC], sca[VEC], aa[VEC], bqa[VEC], mb[VEC], rsb[VEC], scb[VEC], bqb[VEC], zsa[VEC], zha[VEC], zsb[VEC], zhb[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = c * VEC + i;
    mean_rstd(sa + (tx * VEC + i) * 2, invS, eps, ma[i], rsa[i]);
    mean_rstd(sb + (tx * VEC + i) * 2, invS, eps, mb[i], rsb[i]);
    sca[i] = rsa[i] * (ga ? ga[ch] : 1.f);
    scb[i] = rsb[i] * (gb ? gb[ch] : 1.f);
    pair_preact_coeffs(spa, spb, st, ch, ma[i], rsa[i], mb[i], rsb[i], zsa[i], zha[i], zsb[i], zhb[i]);
    aa[i] = (float)(da[(tx * VEC + i) * 2] * invS);          // mean of g: the same for both norms
    bqa[i] = (float)(da[(tx * VEC + i) * 2 + 1] * invS);
    bqb[i] = (float)(db[(tx * VEC + i) * 2 + 1] * invS);
  }
  float r1wv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) r1wv[i] = r1x ? to_f32(r1w[c * VEC + i]) : 0.f;
#pragma unroll 4
  for (int r = r0 + ty; r < r1; r += ty_n) {
    RowVec<T, VEC> g, yv, va, vb, oa, ob;
    g.load(dy + (boff + r) * lddy + c * VEC);
    if (yact) yv.load(yact + (boff + r) * ldy + c * VEC);
    va.load(xa + (boff + r) * ldxa + c * VEC);
    float xs = 0.f;
    if (r1x) {
      xs = to_f32(r1x[(boff + r) * ldr1x]);
#pragma unroll
      for (int i = 0; i < VEC; ++i) vb.v[i] = to_f32(from_f32<T>(xs * r1wv[i]));
    } else {
      vb.load(xb + (boff + r) * ldxb + c * VEC);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float pre = yact ? yv.v[i] : pair_preact(va.v[i], zsa[i], zha[i], vb.v[i], zsb[i], zhb[i]);
      const float gi = pre > 0.f ? g.v[i] : g.v[i] * slope;
      oa.v[i] = sca[i] * (gi - aa[i] - (va.v[i] - ma[i]) * rsa[i] * bqa[i]);
      ob.v[i] = scb[i] * (gi - aa[i] - (vb.v[i] - mb[i]) * rsb[i] * bqb[i]);
    }
    oa.store(dxa + (boff + r) * lddxa + c * VEC);
    if (r1x) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) dwacc[i] = fmaf(to_f32(from_f32<T>(ob.v[i])), xs, dwacc[i]);
    } else {
      ob.store(dxb + (boff + r) * lddxb + c * VEC);
    }
  }
  }
  if (r1x) {     // every thread of the workgroup is here: column sums of dwacc over ty, one fp32 atomic per channel and workgroup
    float* redf = reinterpret_cast<float*>(sums);
    __syncthreads();       // the statistics in `sums` have been read by everyone
    if (ty < ty_n) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) redf[(ty * tx_n + tx) * VEC + i] = dwacc[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < tx_n * VEC; e += NORM_THREADS) {
      const int ch = blockIdx.z * tx_n * VEC + e;
      float tot = 0.f;
      for (int t = 0; t < ty_n; ++t) tot += redf[t * tx_n * VEC + e];
      if (ch < C) atomicAdd(r1dw + ch, tot);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Register-resident fused kernels (round 3): tensors of up to NORM_FUSED_MAX_ROWS = 2048 rows per sample (the 12^3, 6^3 and 3^3 stages).
// One workgroup per (sample, tile of tx_n channel vectors); thread (tx, ty) owns rows ty, ty + ty_n, ... - at most FUSED_MAXR of them - of
// its channel vector and keeps them IN REGISTERS (packed, 4 registers per 16-byte row): every tensor is read exactly once, all its loads are
// in flight together and are issued BEFORE the statistics are gathered, the reduction over the rows is a cross-lane step + one LDS round
// across the four waves, and the second pass (normalise / the input gradient) runs from the registers.  The round-2 kernels walked the rows
// twice with four loads in flight behind the statistics gather (6.5 - 11 us for 166 KB); the chunked pairs they replace at 12^3 were
// two launches with an fp64-atomic hand-off between them (8 + 8 us).
// ---------------------------------------------------------------------------------------------------
static constexpr int FUSED_MAXR = 8;
static constexpr int NORM_FUSED_MAX_ROWS_HW = FUSED_MAXR * NORM_THREADS;
#define NORM_FUSED_MAX_ROWS NORM_FUSED_MAX_ROWS_HW

template <class T, int VEC> struct PRow {       // one row's VEC channels of a lane as loaded (packed: VEC * sizeof(T) bytes, one load)
  typedef T raw_t __attribute__((ext_vector_type(VEC)));
  raw_t raw;
  __device__ __forceinline__ void load(const T* p) { raw = *reinterpret_cast<const raw_t*>(p); }
  __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<raw_t*>(p) = raw; }
  __device__ __forceinline__ float get(int i) const { return to_f32(raw[i]); }
  __device__ __forceinline__ void set(int i, float v) { raw[i] = from_f32<T>(v); }
};
template <class T> struct PRow<T, 1> {
  T raw;
  __device__ __forceinline__ void load(const T* p) { raw = p[0]; }
  __device__ __forceinline__ void store(T* p) const { p[0] = raw; }
  __device__ __forceinline__ float get(int) const { return to_f32(raw); }
  __device__ __forceinline__ void set(int, float v) { raw = from_f32<T>(v); }
};

// totals over the ty rows of the workgroup of NVV per-thread floats.  thread = ty * tx_n + tx with tx_n a power of two <= 16, so the lanes of
// a 16-lane DPP row that share tx are those congruent modulo tx_n: rotations of the row by 8, 4, .. tx_n (plain VALU instructions with a DPP
// operand - the xor shuffles of the first version went through the LDS crossbar, 1.3 - 2.1 us of a 10 us launch) leave every lane with
// its class's row total; the 16 rows of the workgroup then meet in LDS.  tot[k * tx_n + tx] (double) is valid for every thread afterwards.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int NVV>
__device__ __forceinline__ void fused_totals(float (&v)[NVV], float* red, double* tot, int tx_n) {
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 4;
  if (tx_n <= 8) {
#pragma unroll
    for (int k = 0; k < NVV; ++k) v[k] = dpp_add<0x128>(v[k]);      // row_ror:8
  }
  if (tx_n <= 4) {
#pragma unroll
    for (int k = 0; k < NVV; ++k) v[k] = dpp_add<0x124>(v[k]);      // row_ror:4
  }
  if (tx_n <= 2) {
#pragma unroll
    for (int k = 0; k < NVV; ++k) v[k] = dpp_add<0x122>(v[k]);      // row_ror:2
  }
  if (tx_n <= 1) {
#pragma unroll
    for (int k = 0; k < NVV; ++k) v[k] = dpp_add<0x121>(v[k]);      // row_ror:1
  }
  if ((lane & 15) < tx_n) {
#pragma unroll
    for (int k = 0; k < NVV; ++k) red[(row * NVV + k) * tx_n + (lane & 15)] = v[k];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NVV * tx_n; e += NORM_THREADS) {
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < NORM_THREADS / 16; ++w) acc += (double)red[w * NVV * tx_n + e];
    tot[e] = acc;
  }
  __syncthreads();
}

template <int VEC> __device__ __forceinline__ void add_floats(const float* __restrict__ p, float (&a)[VEC]) {
  if constexpr (VEC % 4 == 0) {
#pragma unroll
    for (int j = 0; j < VEC / 4; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * j);
      a[4 * j] += v[0]; a[4 * j + 1] += v[1]; a[4 * j + 2] += v[2]; a[4 * j + 3] += v[3];
    }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) a[j] += p[j];
  }
}

// SLABS: the input is still the `nslabs` fp32 partial slabs [B * S][C] of a split convolution (miseg_conv3_params.defer_slabs): a lane sums
// its elements, rounds them to T - that IS the convolution's output, written to x for the backward pass - and goes on as below.  One launch
// where the split convolution's reduce launch and the norm's apply launch were two (the <= 2048-row layers of the deep stages).
template <class T, int VEC, bool SLABS = false>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_fused_fwd_kernel(T* __restrict__ x, int64_t ldx, const T* __restrict__ res, int64_t ldres,
                                                                          T* __restrict__ y, int64_t ldy, int S, int C, int cv, int tx_n, int ty_n,
                                                                          double* __restrict__ stat, float eps, const int32_t* __restrict__ styles, StylePtrs sp,
                                                                          int act, float slope, const float* __restrict__ slabs = nullptr, int nslabs = 0,
                                                                          int64_t slab_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  double* tot = reinterpret_cast<double*>(red + (NORM_THREADS / 16) * 2 * VEC * tx_n);
  const int b = blockIdx.y;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int c = blockIdx.x * tx_n + tx;
  const bool live = c < cv;
  const int64_t boff = (int64_t)b * S;
  // (style id and affine rows are requested first: they are in flight with the rows)
  const int st = styles ? styles[b] : 0;
  float gam[VEC], bet[VEC];
  {
    const float* g = sp.gamma[st];
    const float* be = sp.beta[st];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int ch = min(c * VEC + i, C - 1);
      gam[i] = g ? g[ch] : 1.f;
      bet[i] = be ? be[ch] : 0.f;
    }
  }
  PRow<T, VEC> xr[FUSED_MAXR], rr[FUSED_MAXR];
  if constexpr (SLABS) {
#pragma unroll
    for (int u = 0; u < FUSED_MAXR; ++u) {
      const int r = ty + u * ty_n;
      if (live && r < S) {
        float a[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) a[i] = 0.f;
        const float* sp0 = slabs + (boff + r) * (int64_t)C + c * VEC;
        for (int k = 0; k < nslabs; ++k) add_floats<VEC>(sp0 + k * slab_stride, a);
#pragma unroll
        for (int i = 0; i < VEC; ++i) xr[u].set(i, a[i]);
        xr[u].store(x + (boff + r) * ldx + c * VEC);
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < FUSED_MAXR; ++u) {
      const int r = ty + u * ty_n;
      if (live && r < S) xr[u].load(x + (boff + r) * ldx + c * VEC);
    }
  }
  if (res) {
#pragma unroll
    for (int u = 0; u < FUSED_MAXR; ++u) {
      const int r = ty + u * ty_n;
      if (live && r < S) rr[u].load(res + (boff + r) * ldres + c * VEC);
    }
  }
  float sq[2 * VEC];
#pragma unroll
  for (int i = 0; i < 2 * VEC; ++i) sq[i] = 0.f;
#pragma unroll
  for (int u = 0; u < FUSED_MAXR; ++u) {
    if (live && ty + u * ty_n < S) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float v = xr[u].get(i); sq[i] += v; sq[VEC + i] = fmaf(v, v, sq[VEC + i]); }
    }
  }
  fused_totals<2 * VEC>(sq, red, tot, tx_n);      // tot[(which * VEC + i) * tx_n + tx]
  // replica 0 of the statistics buffer (the others stay zero): the backward pass reads it like any other
  for (int e = threadIdx.x; e < 2 * VEC * tx_n; e += NORM_THREADS) {
    const int k = e / tx_n, t = e - k * tx_n, which = k / VEC, i = k - which * VEC;
    const int ch = (blockIdx.x * tx_n + t) * VEC + i;
    if (ch < C) stat[((int64_t)b * C + ch) * 2 + which] = tot[e];
  }
  if (!live) return;
  const double invS = 1.0 / S;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const double two[2] = {tot[i * tx_n + tx], tot[(VEC + i) * tx_n + tx]};
    float m, rs;
    mean_rstd(two, invS, eps, m, rs);
    sc[i] = rs * gam[i];
    sh[i] = bet[i] - m * sc[i];
  }
#pragma unroll
  for (int u = 0; u < FUSED_MAXR; ++u) {
    const int r = ty + u * ty_n;
    if (r < S) {
      float v[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = fmaf(xr[u].get(i), sc[i], sh[i]);
      if (res) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] += rr[u].get(i);
      }
      if (act == MISEG_ACT_LEAKY) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
      }
      PRow<T, VEC> o;
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.set(i, v[i]);
      o.store(y + (boff + r) * ldy + c * VEC);
    }
  }
}

// SLABS: dy is still the `nslabs` fp32 partial slabs of the split data-gradient convolution in front of this norm (see the forward kernel):
// summed and rounded to T in registers - the incoming gradient is never written to memory at all.
template <class T, int VEC, bool SLABS = false>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_fused_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ yact, int64_t ldy,
                                                                          const T* __restrict__ x, int64_t ldx, T* __restrict__ dx, int64_t lddx,
                                                                          T* __restrict__ dres, int64_t lddres, int S, int C, int cv, int tx_n, int ty_n,
                                                                          const double* __restrict__ stat, float eps, const int32_t* __restrict__ styles,
                                                                          StylePtrs sp, StyleGradPtrs gp, int act, float slope, const T* __restrict__ gadd,
                                                                          int64_t ldgadd, const float* __restrict__ slabs = nullptr, int nslabs = 0,
                                                                          int64_t slab_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  double* tot = reinterpret_cast<double*>(red + (NORM_THREADS / 16) * 2 * VEC * tx_n);   // totals of (g, g * xhat)
  double* sums = tot + 2 * VEC * tx_n;                                                  // forward statistics [col][2]
  const int b = blockIdx.y;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int c = blockIdx.x * tx_n + tx;
  const bool live = c < cv;
  const int64_t boff = (int64_t)b * S;
  const bool masked = act == MISEG_ACT_LEAKY;
  NSTAMP(0);
  // style id, affine rows and all row loads first: nothing below depends on them until the statistics have arrived as well
  const int st = styles ? styles[b] : 0;
  float gam[VEC], bet[VEC];
  {
    const float* gz = sp.gamma[st];
    const float* bz = sp.beta[st];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int ch = min(c * VEC + i, C - 1);
      gam[i] = gz ? gz[ch] : 1.f;
      bet[i] = (bz && masked && !yact) ? bz[ch] : 0.f;
    }
  }
  PRow<T, VEC> gr[FUSED_MAXR], xr[FUSED_MAXR], yr[FUSED_MAXR];
  const bool from_y = masked && yact;      // a residual entered the activation: its sign comes from the stored output
#pragma unroll
  for (int u = 0; u < FUSED_MAXR; ++u) {
    const int r = ty + u * ty_n;
    if (live && r < S) {
      if constexpr (SLABS) {
        float a[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) a[i] = 0.f;
        const float* sp0 = slabs + (boff + r) * (int64_t)C + c * VEC;
        for (int k = 0; k < nslabs; ++k) add_floats<VEC>(sp0 + k * slab_stride, a);
#pragma unroll
        for (int i = 0; i < VEC; ++i) gr[u].set(i, a[i]);
      } else {
        gr[u].load(dy + (boff + r) * lddy + c * VEC);
      }
      xr[u].load(x + (boff + r) * ldx + c * VEC);
      if (from_y) yr[u].load(yact + (boff + r) * ldy + c * VEC);
    }
  }
  NSTAMP(1);
  gather_stat(sums, stat, (int64_t)gridDim.y * C * 2, b, C, blockIdx.x * tx_n * VEC, tx_n * VEC);
  NSTAMP(2);
  const double invS = 1.0 / S;
  float m[VEC], rs[VEC], sc[VEC], zsh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    mean_rstd(sums + (tx * VEC + i) * 2, invS, eps, m[i], rs[i]);
    sc[i] = rs[i] * gam[i];
    zsh[i] = bet[i] - m[i] * sc[i];      // the forward's shift: without a residual the activation's sign is the sign of fma(x, sc, zsh)
  }
  // the activation-masked gradient of row u; evaluated in both passes (three instructions per element) rather than stored.  The mode
  // tests are per ROW: inside the element loop they were a scalar branch per element - 40 instructions per element, 3.8 us for 7 rows
  auto gmask = [&](int u, float (&gm)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) gm[i] = gr[u].get(i);
    if (from_y) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) gm[i] = yr[u].get(i) > 0.f ? gm[i] : gm[i] * slope;
    } else if (masked) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) gm[i] = fmaf(xr[u].get(i), sc[i], zsh[i]) > 0.f ? gm[i] : gm[i] * slope;
    }
  };
  float sq[2 * VEC];
#pragma unroll
  for (int i = 0; i < 2 * VEC; ++i) sq[i] = 0.f;
#pragma unroll
  for (int u = 0; u < FUSED_MAXR; ++u) {
    if (live && ty + u * ty_n < S) {
      float gm[VEC];
      gmask(u, gm);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        sq[i] += gm[i];
        sq[VEC + i] = fmaf(gm[i], (xr[u].get(i) - m[i]) * rs[i], sq[VEC + i]);
      }
    }
  }
  NSTAMP(3);
  fused_totals<2 * VEC>(sq, red, tot, tx_n);
  NSTAMP(4);
  for (int e = threadIdx.x; e < VEC * tx_n; e += NORM_THREADS) {      // the affine gradients of this (sample, channel tile)
    const int i = e / tx_n, t = e - i * tx_n;
    const int ch = (blockIdx.x * tx_n + t) * VEC + i;
    if (ch < C) {
      if (gp.dgamma[st]) atomicAdd(gp.dgamma[st] + ch, (float)tot[(VEC + i) * tx_n + t]);
      if (gp.dbeta[st]) atomicAdd(gp.dbeta[st] + ch, (float)tot[i * tx_n + t]);
    }
  }
  NSTAMP(5);
  if (!live) return;
  float a[VEC], bq[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    a[i] = (float)(tot[i * tx_n + tx] * invS);
    bq[i] = (float)(tot[(VEC + i) * tx_n + tx] * invS);
  }
#pragma unroll
  for (int u = 0; u < FUSED_MAXR; ++u) {
    const int r = ty + u * ty_n;
    if (r < S) {
      float gm[VEC], v[VEC];
      gmask(u, gm);
      if (dres) {
        PRow<T, VEC> dr;
#pragma unroll
        for (int i = 0; i < VEC; ++i) dr.set(i, gm[i]);
        dr.store(dres + (boff + r) * lddres + c * VEC);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = sc[i] * (gm[i] - a[i] - (xr[u].get(i) - m[i]) * rs[i] * bq[i]);
      if (gadd) {
        PRow<T, VEC> ga;
        ga.load(gadd + (boff + r) * ldgadd + c * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] += ga.get(i);
      }
      PRow<T, VEC> o;
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.set(i, v[i]);
      o.store(dx + (boff + r) * lddx + c * VEC);
    }
  }
  NSTAMP(6);
}

// backward of y = LeakyReLU(norm_a(xa) + norm_b(xb)) in ONE launch (see instnorm_pair_bwd_{reduce,apply}_kernel for the arithmetic):
// dy, xa, xb stay in registers, the three sums (g, g * xhat_a, g * xhat_b) meet inside the workgroup.
template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_pair_fused_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ yact, int64_t ldy,
                                                                               const T* __restrict__ xa, int64_t ldxa, const T* __restrict__ xb, int64_t ldxb,
                                                                               T* __restrict__ dxa, int64_t lddxa, T* __restrict__ dxb, int64_t lddxb, int S, int C,
                                                                               int cv, int tx_n, int ty_n, const double* __restrict__ stat_a,
                                                                               const double* __restrict__ stat_b, float eps, const int32_t* __restrict__ styles,
                                                                               StylePtrs spa, StylePtrs spb, float slope, StyleGradPtrs gpa, StyleGradPtrs gpb) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  double* tot = reinterpret_cast<double*>(red + (NORM_THREADS / 16) * 3 * VEC * tx_n);   // totals of (g, g * xhat_a, g * xhat_b)
  double* sums_a = tot + 3 * VEC * tx_n;
  double* sums_b = sums_a + 2 * VEC * tx_n;
  const int b = blockIdx.y;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int c = blockIdx.x * tx_n + tx;
  const bool live = c < cv;
  const int64_t boff = (int64_t)b * S;
  // style id + affine rows first (in flight with the rows)
  const int st = styles ? styles[b] : 0;
  float gma[VEC], bta[VEC], gmb[VEC], btb[VEC];
  {
    const float* ga = spa.gamma[st];
    const float* ba = spa.beta[st];
    const float* gb = spb.gamma[st];
    const float* bb = spb.beta[st];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int ch = min(c * VEC + i, C - 1);
      gma[i] = ga ? ga[ch] : 1.f;
      bta[i] = ba ? ba[ch] : 0.f;
      gmb[i] = gb ? gb[ch] : 1.f;
      btb[i] = bb ? bb[ch] : 0.f;
    }
  }
  PRow<T, VEC> gr[FUSED_MAXR], ar[FUSED_MAXR], br[FUSED_MAXR], yr[FUSED_MAXR];
#pragma unroll
  for (int u = 0; u < FUSED_MAXR; ++u) {
    const int r = ty + u * ty_n;
    if (live && r < S) {
      gr[u].load(dy + (boff + r) * lddy + c * VEC);
      ar[u].load(xa + (boff + r) * ldxa + c * VEC);
      br[u].load(xb + (boff + r) * ldxb + c * VEC);
      if (yact) yr[u].load(yact + (boff + r) * ldy + c * VEC);
    }
  }
  gather_stat(sums_a, stat_a, (int64_t)gridDim.y * C * 2, b, C, blockIdx.x * tx_n * VEC, tx_n * VEC);
  gather_stat(sums_b, stat_b, (int64_t)gridDim.y * C * 2, b, C, blockIdx.x * tx_n * VEC, tx_n * VEC);
  const double invS = 1.0 / S;
  float ma[VEC], rsa[VEC], mb[VEC], rsb[VEC], sca[VEC], scb[VEC], zha[VEC], zhb[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    mean_rstd(sums_a + (tx * VEC + i) * 2, invS, eps, ma[i], rsa[i]);
    mean_rstd(sums_b + (tx * VEC + i) * 2, invS, eps, mb[i], rsb[i]);
    sca[i] = rsa[i] * gma[i];      // the expressions of pair_preact_coeffs (instnorm_apply_kernel's scale / shift): the recomputed sign
    zha[i] = bta[i] - ma[i] * sca[i];      // must be the sign the forward pass saw
    scb[i] = rsb[i] * gmb[i];
    zhb[i] = btb[i] - mb[i] * scb[i];
  }
  auto gmask = [&](int u, float (&gm)[VEC]) {      // (mode test per row, not per element: see instnorm_fused_bwd_kernel)
    if (yact) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float g = gr[u].get(i); gm[i] = yr[u].get(i) > 0.f ? g : g * slope; }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float g = gr[u].get(i);
        gm[i] = pair_preact(ar[u].get(i), sca[i], zha[i], br[u].get(i), scb[i], zhb[i]) > 0.f ? g : g * slope;
      }
    }
  };
  float sq[3 * VEC];
#pragma unroll
  for (int i = 0; i < 3 * VEC; ++i) sq[i] = 0.f;
#pragma unroll
  for (int u = 0; u < FUSED_MAXR; ++u) {
    if (live && ty + u * ty_n < S) {
      float gm[VEC];
      gmask(u, gm);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        sq[i] += gm[i];
        sq[VEC + i] = fmaf(gm[i], (ar[u].get(i) - ma[i]) * rsa[i], sq[VEC + i]);
        sq[2 * VEC + i] = fmaf(gm[i], (br[u].get(i) - mb[i]) * rsb[i], sq[2 * VEC + i]);
      }
    }
  }
  fused_totals<3 * VEC>(sq, red, tot, tx_n);
  for (int e = threadIdx.x; e < VEC * tx_n; e += NORM_THREADS) {
    const int i = e / tx_n, t = e - i * tx_n;
    const int ch = (blockIdx.x * tx_n + t) * VEC + i;
    if (ch < C) {
      if (gpa.dgamma[st]) atomicAdd(gpa.dgamma[st] + ch, (float)tot[(VEC + i) * tx_n + t]);
      if (gpa.dbeta[st]) atomicAdd(gpa.dbeta[st] + ch, (float)tot[i * tx_n + t]);
      if (gpb.dgamma[st]) atomicAdd(gpb.dgamma[st] + ch, (float)tot[(2 * VEC + i) * tx_n + t]);
      if (gpb.dbeta[st]) atomicAdd(gpb.dbeta[st] + ch, (float)tot[i * tx_n + t]);
    }
  }
  if (!live) return;
  float aa[VEC], bqa[VEC], bqb[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    aa[i] = (float)(tot[i * tx_n + tx] * invS);
    bqa[i] = (float)(tot[(VEC + i) * tx_n + tx] * invS);
    bqb[i] = (float)(tot[(2 * VEC + i) * tx_n + tx] * invS);
  }
#pragma unroll
  for (int u = 0; u < FUSED_MAXR; ++u) {
    const int r = ty + u * ty_n;
    if (r < S) {
      PRow<T, VEC> oa, ob;
      float gm[VEC];
      gmask(u, gm);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        oa.set(i, sca[i] * (gm[i] - aa[i] - (ar[u].get(i) - ma[i]) * rsa[i] * bqa[i]));
        ob.set(i, scb[i] * (gm[i] - aa[i] - (br[u].get(i) - mb[i]) * rsb[i] * bqb[i]));
      }
      oa.store(dxa + (boff + r) * lddxa + c * VEC);
      ob.store(dxb + (boff + r) * lddxb + c * VEC);
    }
  }
}

// thread geometry of the register-resident kernels: ty_n = the power of two >= S, between 16 and 256 - as FEW rows per lane as the tensor
// allows, i.e. as many workgroups as it has channel vectors: these launches are bound by the instructions one wave issues (in-kernel stamps:
// 8 rows x 8 channels per lane cost 3.2 us of arithmetic alone), not by bytes - and tx_n = 256 / ty_n <= 16 channel vectors
struct FusedGeom { int vec, cv, tx, ty; };
static FusedGeom fused_geom(int S, int C, bool vec_ok, int vecN) {
  FusedGeom g;
  int ty = 16;
  while (ty < S && ty < NORM_THREADS) ty <<= 1;
  g.ty = ty;
  g.tx = NORM_THREADS / ty;
  const int rows_per_lane = cdiv(S, ty);
  // channels per lane: 16 bytes where a lane has few rows; narrower (8- / 4-byte loads) where it has many, so that a lane handles <= 16
  // elements and the tensor spreads over 2 - 4 x the workgroups - at 1728 rows x 192 channels the 16-byte form was 24 workgroups of 56
  // elements per lane: 17 us of issue-bound arithmetic (one wave per SIMD) for 0.66 MB
  int vec = 1;
  if (vec_ok) {
    vec = vecN;
    while (vec > 1 && C % vec != 0) vec >>= 1;
    while (vec > 2 && rows_per_lane * vec > 16) vec >>= 1;
  }
  g.vec = vec;
  g.cv = C / g.vec;
  return g;
}
// launch `KERNEL<T, vec>` for the lane width the geometry chose (bf16: 8 / 4 / 2 / 1 channels, fp32: 4 / 2 / 1)
#define FUSED_DISPATCH(GEOM, LAUNCH)                          \
  do {                                                        \
    if ((GEOM).vec == 8) { if constexpr (V >= 8) { LAUNCH(8); } } \
    else if ((GEOM).vec == 4) { if constexpr (V >= 4) { LAUNCH(4); } } \
    else if ((GEOM).vec == 2) { LAUNCH(2); }                  \
    else { LAUNCH(1); }                                       \
  } while (0)
static size_t fused_smem(const FusedGeom& g, int nsum, int nstat) {      // red + tot + gathered statistics
  return (size_t)(NORM_THREADS / 16) * nsum * g.vec * g.tx * sizeof(float) + (size_t)nsum * g.vec * g.tx * sizeof(double) +
         (size_t)nstat * 2 * g.vec * g.tx * sizeof(double);
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over C per row: one wave per row (C <= 8192)
// ---------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int64_t rows, int C,
                                                            float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ mean, float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += to_f32(xr[c]);
  s = wave_sum(s);
  const float m = s / C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = to_f32(xr[c]) - m; q = fmaf(d, d, q); }
  q = wave_sum(q);
  const float rs = 1.0f / sqrtf(q / C + eps);      // correctly rounded (v_rsq_f32 is 1 ulp: the fp32 parity mode wants torch's value)
  if (lane == 0) { mean[row] = m; rstd[row] = rs; }
  T* yr = y + row * ldy;
  for (int c = lane; c < C; c += 64) {
    float v = (to_f32(xr[c]) - m) * rs;
    if (gamma) v = v * gamma[c];
    if (beta) v += beta[c];
    yr[c] = from_f32<T>(v);
  }
}

template <class T>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx, T* __restrict__ dx,
                                                            int64_t lddx, int64_t rows, int C, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  const T* gr = dy + row * lddy;
  const float m = mean[row], rs = rstd[row];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float g = to_f32(gr[c]) * (gamma ? gamma[c] : 1.f);
    const float xh = (to_f32(xr[c]) - m) * rs;
    s1 += g;
    s2 = fmaf(g, xh, s2);
  }
  s1 = wave_sum(s1) / C;
  s2 = wave_sum(s2) / C;
  T* dr = dx + row * lddx;
  for (int c = lane; c < C; c += 64) {
    const float g = to_f32(gr[c]) * (gamma ? gamma[c] : 1.f);
    const float xh = (to_f32(xr[c]) - m) * rs;
    dr[c] = from_f32<T>(rs * (g - s1 - xh * s2));
  }
}

// dgamma[c] += sum_r dy*xhat ; dbeta[c] += sum_r dy   (block handles a row slab, atomics at the end)
template <class T>
__global__ void __launch_bounds__(256) layernorm_bwd_param_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx,
                                                                  int64_t rows, int C, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, int rows_per_block) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sg = 0.f, sb = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float g = to_f32(dy[r * lddy + c]);
      sg = fmaf(g, (to_f32(x[r * ldx + c]) - mean[r]) * rstd[r], sg);
      sb += g;
    }
    if (dgamma) atomicAdd(dgamma + c, sg);
    if (dbeta) atomicAdd(dbeta + c, sb);
  }
}

}  // namespace miseg

using namespace miseg;

static bool aligned16(const void* p) { return ((uintptr_t)p % 16) == 0; }

namespace miseg {
// internal (conv3d.hip): enqueue the kernel above; stat may be null (plain sum + convert)
int slabs_to_out_stats(const float* slabs, int nslabs, void* y, int64_t ldy, const void* res, int64_t ldres, int B, int S, int C, int dtype, double* stat,
                       hipStream_t stream) {
  return dispatch_dtype(dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    constexpr int V = Vec16<T>::N;
    const bool al = aligned16(y) && ldy % V == 0 && (!res || (aligned16(res) && ldres % V == 0)) && C % 4 == 0 && aligned16(slabs);
    // the tensors of a split convolution are tiny (27 .. 1728 rows): 8 vector columns x 32 rows per workgroup pass, so that even the 3^3
    // volume is a dozen workgroups (norm_geom's >= 4 rows per lane made it three, 15 us for 16 slabs of 83 KB)
    NormGeom g;
    g.vec = (al && C % V == 0) ? V : 1;
    g.cv = C / g.vec;
    g.tx = g.cv < 8 ? g.cv : 8;
    g.ty = NORM_THREADS / g.tx;
    g.ctiles = cdiv(g.cv, g.tx);
    int per = cdiv(S, g.ty * 64);
    g.rpb = g.ty * (per < 1 ? 1 : per > 8 ? 8 : per);
    g.chunks = cdiv(S, g.rpb);
    dim3 grid(g.chunks, B, g.ctiles);
    const size_t sh = (size_t)g.ty * 2 * g.tx * g.vec * sizeof(float);
    const int64_t stride = (int64_t)B * S * C;
    if (g.vec == 1)
      slabs_to_out_stats_kernel<T, 1><<<grid, NORM_THREADS, sh, stream>>>(slabs, nslabs, stride, (T*)y, ldy, (const T*)res, ldres, S, C, g.cv, g.tx, g.ty, g.rpb, stat);
    else
      slabs_to_out_stats_kernel<T, V><<<grid, NORM_THREADS, sh, stream>>>(slabs, nslabs, stride, (T*)y, ldy, (const T*)res, ldres, S, C, g.cv, g.tx, g.ty, g.rpb, stat);
    MISEG_LAUNCH_CHECK("slabs_to_out_stats");
    return MISEG_OK;
  });
}
}  // namespace miseg

#ifdef MISEG_NORM_STAMPS
extern "C" int miseg_debug_norm_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(miseg::g_norm_stamps), sizeof(miseg::g_norm_stamps)); }
#endif

extern "C" size_t miseg_instnorm_stat_bytes(int B, int C) { return (size_t)NORM_R * B * C * 2 * sizeof(double); }

extern "C" int miseg_instnorm_stats(const miseg_instnorm_stats_params* p, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->x && p->stat, MISEG_E_BADARG, "instnorm_stats: null pointer");
  MISEG_REQUIRE(p->B > 0 && p->S > 0 && p->C > 0 && p->ldx >= p->C, MISEG_E_BADARG, "instnorm_stats: bad shape B=%d S=%d C=%d ld=%ld", p->B, p->S, p->C,
                (long)p->ldx);
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    constexpr int V = Vec16<T>::N;
    NormGeom g = norm_geom(p->S, p->C, aligned16(p->x) && p->ldx % V == 0, V, 256);
    dim3 grid(g.chunks, p->B, g.ctiles);
    size_t sh = (size_t)g.ty * 2 * g.tx * g.vec * sizeof(float);
    if (g.vec == 1) instnorm_stats_kernel<T, 1><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->x, p->ldx, p->S, p->C, g.cv, g.tx, g.ty, g.rpb, (double*)p->stat);
    else instnorm_stats_kernel<T, V><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->x, p->ldx, p->S, p->C, g.cv, g.tx, g.ty, g.rpb, (double*)p->stat);
    MISEG_LAUNCH_CHECK("instnorm_stats");
    return MISEG_OK;
  });
}

extern "C" int miseg_instnorm_apply(const miseg_instnorm_apply_params* p, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->x && p->y && p->stat, MISEG_E_BADARG, "instnorm_apply: null pointer");
  MISEG_REQUIRE(p->num_styles >= 1 && p->num_styles <= MISEG_MAX_STYLES, MISEG_E_BADARG, "instnorm_apply: num_styles %d", p->num_styles);
  MISEG_REQUIRE(p->act == MISEG_ACT_NONE || p->act == MISEG_ACT_LEAKY, MISEG_E_UNSUPPORTED, "instnorm_apply: act %d", p->act);
  MISEG_REQUIRE(!p->res_stat || p->res || p->r1x, MISEG_E_BADARG, "instnorm_apply: res_stat without res");
  MISEG_REQUIRE(!p->r1x || (p->res_stat && !p->res && p->r1w), MISEG_E_BADARG, "instnorm_apply: r1x needs res_stat and r1w, and excludes res");
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    constexpr int V = Vec16<T>::N;
    const int64_t ldor = p->ldx | p->ldy | (p->res ? p->ldres : 0);
    const bool al = aligned16(p->x) && aligned16(p->y) && (!p->res || aligned16(p->res)) && ldor % V == 0;
    NormGeom g = norm_geom(p->S, p->C, al, V);
    StylePtrs sp;
    for (int s = 0; s < MISEG_MAX_STYLES; ++s) { sp.gamma[s] = s < p->num_styles ? p->gamma[s] : nullptr; sp.beta[s] = s < p->num_styles ? p->beta[s] : nullptr; }
    StylePtrs rsp;
    for (int s = 0; s < MISEG_MAX_STYLES; ++s) {
      rsp.gamma[s] = (p->res_stat && s < p->num_styles) ? p->res_gamma[s] : nullptr;
      rsp.beta[s] = (p->res_stat && s < p->num_styles) ? p->res_beta[s] : nullptr;
    }
    dim3 grid(g.chunks, p->B, g.ctiles);
    const size_t shd = (size_t)4 * g.tx * g.vec * sizeof(double);
    if (g.vec == 1)
      instnorm_apply_kernel<T, 1><<<grid, NORM_THREADS, shd, stream>>>((const T*)p->x, p->ldx, (const T*)p->res, p->ldres, (T*)p->y, p->ldy, p->S, p->C, g.cv, g.tx,
                                                                      g.ty, g.rpb, (const double*)p->stat, p->eps, p->styles, sp, p->act, p->slope,
                                                                      (const double*)p->res_stat, rsp, (const T*)p->r1x, p->ldr1x, (const T*)p->r1w);
    else
      instnorm_apply_kernel<T, V><<<grid, NORM_THREADS, shd, stream>>>((const T*)p->x, p->ldx, (const T*)p->res, p->ldres, (T*)p->y, p->ldy, p->S, p->C, g.cv, g.tx,
                                                                      g.ty, g.rpb, (const double*)p->stat, p->eps, p->styles, sp, p->act, p->slope,
                                                                      (const double*)p->res_stat, rsp, (const T*)p->r1x, p->ldr1x, (const T*)p->r1w);
    MISEG_LAUNCH_CHECK("instnorm_apply");
    return MISEG_OK;
  });
}

extern "C" int miseg_instnorm_fused_max_rows(void) { return NORM_FUSED_MAX_ROWS; }

// statistics + normalisation: the two kernels above, or one fused launch for the small tensors of the deep stages
static int instnorm_fwd_impl(const miseg_instnorm_apply_params* p, const float* slabs, int nslabs, int64_t slab_stride, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->x && p->y && p->stat, MISEG_E_BADARG, "instnorm_fwd: null pointer");
  MISEG_REQUIRE(!p->res_stat, MISEG_E_UNSUPPORTED, "instnorm_fwd: res_stat (shortcut norm on the fly) is a miseg_instnorm_apply feature");
  if (slabs) {
    MISEG_REQUIRE(p->S <= NORM_FUSED_MAX_ROWS, MISEG_E_UNSUPPORTED, "instnorm_fwd_slabs: %d rows per sample (at most %d: miseg_instnorm_fused_max_rows)", p->S,
                  NORM_FUSED_MAX_ROWS);
    MISEG_REQUIRE(nslabs >= 1 && slab_stride >= (int64_t)p->B * p->S * p->C, MISEG_E_BADARG, "instnorm_fwd_slabs: nslabs / slab_stride");
  }
  if (p->S > NORM_FUSED_MAX_ROWS) {
    miseg_instnorm_stats_params sp_{p->x, p->ldx, p->B, p->S, p->C, p->dtype, const_cast<void*>(p->stat)};
    const int rc = miseg_instnorm_stats(&sp_, stream_);
    return rc ? rc : miseg_instnorm_apply(p, stream_);
  }
  MISEG_REQUIRE(p->num_styles >= 1 && p->num_styles <= MISEG_MAX_STYLES, MISEG_E_BADARG, "instnorm_fwd: num_styles %d", p->num_styles);
  MISEG_REQUIRE(p->act == MISEG_ACT_NONE || p->act == MISEG_ACT_LEAKY, MISEG_E_UNSUPPORTED, "instnorm_fwd: act %d", p->act);
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    constexpr int V = Vec16<T>::N;
    const int64_t ldor = p->ldx | p->ldy | (p->res ? p->ldres : 0) | (slabs ? p->C : 0);
    const bool al = aligned16(p->x) && aligned16(p->y) && (!p->res || aligned16(p->res)) && ldor % V == 0 && (!slabs || (aligned16(slabs) && slab_stride % 4 == 0));
    const FusedGeom g = fused_geom(p->S, p->C, al, V);
    StylePtrs sp;
    for (int s = 0; s < MISEG_MAX_STYLES; ++s) { sp.gamma[s] = s < p->num_styles ? p->gamma[s] : nullptr; sp.beta[s] = s < p->num_styles ? p->beta[s] : nullptr; }
    dim3 grid(cdiv(g.cv, g.tx), p->B);
    const size_t sh = fused_smem(g, 2, 0);
#define FWD_LAUNCH(VV)                                                                                                                                       \
    if (slabs)                                                                                                                                               \
      instnorm_fused_fwd_kernel<T, VV, true><<<grid, NORM_THREADS, sh, stream>>>((T*)p->x, p->ldx, (const T*)p->res, p->ldres, (T*)p->y, p->ldy, p->S, p->C, g.cv, \
                                                                                 g.tx, g.ty, (double*)p->stat, p->eps, p->styles, sp, p->act, p->slope, slabs,       \
                                                                                 nslabs, slab_stride);                                                               \
    else                                                                                                                                                     \
      instnorm_fused_fwd_kernel<T, VV, false><<<grid, NORM_THREADS, sh, stream>>>((T*)p->x, p->ldx, (const T*)p->res, p->ldres, (T*)p->y, p->ldy, p->S, p->C, g.cv, \
                                                                                  g.tx, g.ty, (double*)p->stat, p->eps, p->styles, sp, p->act, p->slope)
    FUSED_DISPATCH(g, FWD_LAUNCH);
#undef FWD_LAUNCH
    MISEG_LAUNCH_CHECK("instnorm_fwd");
    return MISEG_OK;
  });
}

extern "C" int miseg_instnorm_fwd(const miseg_instnorm_apply_params* p, miseg_stream_t stream) { return instnorm_fwd_impl(p, nullptr, 0, 0, stream); }

extern "C" int miseg_instnorm_fwd_slabs(const miseg_instnorm_apply_params* p, const float* slabs, int nslabs, int64_t slab_stride, miseg_stream_t stream) {
  MISEG_REQUIRE(slabs, MISEG_E_BADARG, "instnorm_fwd_slabs: null slabs");
  return instnorm_fwd_impl(p, slabs, nslabs, slab_stride, stream);
}

// apply_only: p->dstat already holds the backward sums (the epilogue of the GEMM that produced dy added them: miseg_gemm_params.stat_mode 2,
// miseg_mlp_params.bs_dstat) - the reduction launch is skipped and tensors of any size take the streaming apply kernel
static int instnorm_bwd_impl(const miseg_instnorm_bwd_params* p, const float* slabs, int nslabs, int64_t slab_stride, miseg_stream_t stream_, bool apply_only = false) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && (p->dy || slabs) && p->x && p->dx && p->stat && p->dstat, MISEG_E_BADARG, "instnorm_bwd: null pointer");
  if (slabs) {
    MISEG_REQUIRE(p->S <= NORM_FUSED_MAX_ROWS, MISEG_E_UNSUPPORTED, "instnorm_bwd_slabs: %d rows per sample (at most %d: miseg_instnorm_fused_max_rows)", p->S,
                  NORM_FUSED_MAX_ROWS);
    MISEG_REQUIRE(nslabs >= 1 && slab_stride >= (int64_t)p->B * p->S * p->C, MISEG_E_BADARG, "instnorm_bwd_slabs: nslabs / slab_stride");
  }
  MISEG_REQUIRE(p->act == MISEG_ACT_NONE || p->act == MISEG_ACT_LEAKY, MISEG_E_BADARG, "instnorm_bwd: act %d", p->act);
  MISEG_REQUIRE(p->num_styles >= 1 && p->num_styles <= MISEG_MAX_STYLES, MISEG_E_BADARG, "instnorm_bwd: num_styles %d", p->num_styles);
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    constexpr int V = Vec16<T>::N;
    const int64_t ldor = (slabs ? p->C : p->lddy) | p->ldx | p->lddx | (p->act ? p->ldy : 0) | (p->dres ? p->lddres : 0) | (p->gadd ? p->ldgadd : 0);
    const bool al = (slabs ? aligned16(slabs) && slab_stride % 4 == 0 : aligned16(p->dy)) && aligned16(p->x) && aligned16(p->dx) && (!p->act || aligned16(p->y)) &&
                    (!p->dres || aligned16(p->dres)) && (!p->gadd || aligned16(p->gadd)) && ldor % V == 0;
    NormGeom g = norm_geom(p->S, p->C, al, V);
    StylePtrs sp;
    StyleGradPtrs gp;
    for (int s = 0; s < MISEG_MAX_STYLES; ++s) {
      sp.gamma[s] = s < p->num_styles ? p->gamma[s] : nullptr;
      sp.beta[s] = s < p->num_styles ? p->beta[s] : nullptr;
      gp.dgamma[s] = s < p->num_styles ? p->dgamma[s] : nullptr;
      gp.dbeta[s] = s < p->num_styles ? p->dbeta[s] : nullptr;
    }
    if (p->S <= NORM_FUSED_MAX_ROWS && !apply_only) {
      const FusedGeom f = fused_geom(p->S, p->C, al, V);
      dim3 fgrid(cdiv(f.cv, f.tx), p->B);
      const size_t fsh = fused_smem(f, 2, 1);
#define FBWD_LAUNCH(VV)                                                                                                                                      \
      if (slabs)                                                                                                                                             \
        instnorm_fused_bwd_kernel<T, VV, true><<<fgrid, NORM_THREADS, fsh, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->x, p->ldx,         \
                                                                                     (T*)p->dx, p->lddx, (T*)p->dres, p->lddres, p->S, p->C, f.cv, f.tx, f.ty,        \
                                                                                     (const double*)p->stat, p->eps, p->styles, sp, gp, p->act, p->slope,             \
                                                                                     (const T*)p->gadd, p->ldgadd, slabs, nslabs, slab_stride);                      \
      else                                                                                                                                                   \
        instnorm_fused_bwd_kernel<T, VV, false><<<fgrid, NORM_THREADS, fsh, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->x, p->ldx,        \
                                                                                      (T*)p->dx, p->lddx, (T*)p->dres, p->lddres, p->S, p->C, f.cv, f.tx, f.ty,       \
                                                                                      (const double*)p->stat, p->eps, p->styles, sp, gp, p->act, p->slope,            \
                                                                                      (const T*)p->gadd, p->ldgadd)
      FUSED_DISPATCH(f, FBWD_LAUNCH);
#undef FBWD_LAUNCH
      MISEG_LAUNCH_CHECK("instnorm_bwd");
      return MISEG_OK;
    }
    dim3 grid(g.chunks, p->B, g.ctiles);
    size_t sh = (size_t)g.ty * 2 * g.tx * g.vec * sizeof(float);
    if (sh < (size_t)2 * g.tx * g.vec * sizeof(double)) sh = (size_t)2 * g.tx * g.vec * sizeof(double);
    const size_t shd = (size_t)4 * g.tx * g.vec * sizeof(double);
    const double* stat = (const double*)p->stat;
    double* dstat = (double*)p->dstat;
#define BWD_LAUNCH(VV)                                                                                                                                          \
    if (!apply_only)                                                                                                                                           \
      instnorm_bwd_reduce_kernel<T, VV><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->x, p->ldx, p->S, p->C,  \
                                                                             g.cv, g.tx, g.ty, g.rpb, stat, p->eps, p->act, p->slope, dstat, p->styles, sp);             \
    instnorm_bwd_apply_kernel<T, VV><<<grid, NORM_THREADS, shd, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->x, p->ldx, (T*)p->dx,    \
                                                                         p->lddx, (T*)p->dres, p->lddres, p->S, p->C, g.cv, g.tx, g.ty, g.rpb, stat, p->eps, p->styles, \
                                                                         sp, p->act, p->slope, dstat, gp, (const T*)p->gadd, p->ldgadd);
    if (g.vec == 1) { BWD_LAUNCH(1) } else { BWD_LAUNCH(V) }
#undef BWD_LAUNCH
    MISEG_LAUNCH_CHECK("instnorm_bwd");
    return MISEG_OK;
  });
}

extern "C" int miseg_instnorm_bwd(const miseg_instnorm_bwd_params* p, miseg_stream_t stream) { return instnorm_bwd_impl(p, nullptr, 0, 0, stream); }

extern "C" int miseg_instnorm_bwd_apply(const miseg_instnorm_bwd_params* p, miseg_stream_t stream) {
  MISEG_REQUIRE(p && p->act == MISEG_ACT_NONE, MISEG_E_UNSUPPORTED, "instnorm_bwd_apply: no activation (the producer's sums know nothing of one)");
  return instnorm_bwd_impl(p, nullptr, 0, 0, stream, true);
}

extern "C" int miseg_instnorm_bwd_slabs(const miseg_instnorm_bwd_params* p, const float* slabs, int nslabs, int64_t slab_stride, miseg_stream_t stream) {
  MISEG_REQUIRE(slabs, MISEG_E_BADARG, "instnorm_bwd_slabs: null slabs");
  return instnorm_bwd_impl(p, slabs, nslabs, slab_stride, stream);
}

extern "C" int miseg_instnorm_bwd_reduce(const miseg_instnorm_bwd_params* p, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->dy && p->x && p->stat && p->dstat, MISEG_E_BADARG, "instnorm_bwd_reduce: null pointer");
  MISEG_REQUIRE(p->act == MISEG_ACT_NONE, MISEG_E_UNSUPPORTED, "instnorm_bwd_reduce: no activation (compose it outside)");
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    constexpr int V = Vec16<T>::N;
    const bool al = aligned16(p->dy) && aligned16(p->x) && (p->lddy | p->ldx) % V == 0;
    NormGeom g = norm_geom(p->S, p->C, al, V);
    StylePtrs sp;
    for (int s = 0; s < MISEG_MAX_STYLES; ++s) { sp.gamma[s] = nullptr; sp.beta[s] = nullptr; }
    dim3 grid(g.chunks, p->B, g.ctiles);
    size_t sh = (size_t)g.ty * 2 * g.tx * g.vec * sizeof(float);
    if (sh < (size_t)2 * g.tx * g.vec * sizeof(double)) sh = (size_t)2 * g.tx * g.vec * sizeof(double);
    if (g.vec == 1)
      instnorm_bwd_reduce_kernel<T, 1><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->dy, p->lddy, nullptr, 0, (const T*)p->x, p->ldx, p->S, p->C, g.cv, g.tx, g.ty,
                                                                           g.rpb, (const double*)p->stat, p->eps, MISEG_ACT_NONE, 0.f, (double*)p->dstat, nullptr, sp);
    else
      instnorm_bwd_reduce_kernel<T, V><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->dy, p->lddy, nullptr, 0, (const T*)p->x, p->ldx, p->S, p->C, g.cv, g.tx, g.ty,
                                                                           g.rpb, (const double*)p->stat, p->eps, MISEG_ACT_NONE, 0.f, (double*)p->dstat, nullptr, sp);
    MISEG_LAUNCH_CHECK("instnorm_bwd_reduce");
    return MISEG_OK;
  });
}

extern "C" int miseg_instnorm_pair_bwd(const miseg_instnorm_pair_bwd_params* p, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->dy && p->xa && p->dxa && p->stat_a && p->stat_b && p->dstat_a && p->dstat_b, MISEG_E_BADARG, "instnorm_pair_bwd: null pointer");
  MISEG_REQUIRE(p->r1x ? (p->r1w && p->r1dw && !p->xb && !p->dxb && !p->y) : (p->xb && p->dxb), MISEG_E_BADARG,
                "instnorm_pair_bwd: either xb / dxb, or the rank-1 shortcut (r1x, r1w, r1dw; no y)");
  MISEG_REQUIRE(p->num_styles >= 1 && p->num_styles <= MISEG_MAX_STYLES, MISEG_E_BADARG, "instnorm_pair_bwd: num_styles %d", p->num_styles);
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    constexpr int V = Vec16<T>::N;
    const int64_t ldor = p->lddy | (p->y ? p->ldy : 0) | p->ldxa | (p->xb ? p->ldxb | p->lddxb : 0) | p->lddxa;
    const bool al = aligned16(p->dy) && (!p->y || aligned16(p->y)) && aligned16(p->xa) && (!p->xb || (aligned16(p->xb) && aligned16(p->dxb))) && aligned16(p->dxa) &&
                    ldor % V == 0 && (!p->r1x || p->C % V == 0);
    NormGeom g = norm_geom(p->S, p->C, al, V);
    StylePtrs spa, spb;
    StyleGradPtrs gpa, gpb;
    for (int s = 0; s < MISEG_MAX_STYLES; ++s) {
      const bool on = s < p->num_styles;
      spa.gamma[s] = on ? p->gamma_a[s] : nullptr; spa.beta[s] = on ? p->beta_a[s] : nullptr;      // betas: only read when y is absent
      spb.gamma[s] = on ? p->gamma_b[s] : nullptr; spb.beta[s] = on ? p->beta_b[s] : nullptr;
      gpa.dgamma[s] = on ? p->dgamma_a[s] : nullptr; gpa.dbeta[s] = on ? p->dbeta_a[s] : nullptr;
      gpb.dgamma[s] = on ? p->dgamma_b[s] : nullptr; gpb.dbeta[s] = on ? p->dbeta_b[s] : nullptr;
    }
    if (!p->r1x && p->S <= NORM_FUSED_MAX_ROWS) {      // small tensors: one register-resident launch
      const FusedGeom f = fused_geom(p->S, p->C, al, V);
      dim3 fgrid(cdiv(f.cv, f.tx), p->B);
      const size_t fsh = fused_smem(f, 3, 2);
#define PFBWD_LAUNCH(VV)                                                                                                                                     \
      instnorm_pair_fused_bwd_kernel<T, VV><<<fgrid, NORM_THREADS, fsh, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->xa, p->ldxa,         \
                                                                                  (const T*)p->xb, p->ldxb, (T*)p->dxa, p->lddxa, (T*)p->dxb, p->lddxb, p->S, p->C,  \
                                                                                  f.cv, f.tx, f.ty, (const double*)p->stat_a, (const double*)p->stat_b, p->eps,   \
                                                                                  p->styles, spa, spb, p->slope, gpa, gpb)
      FUSED_DISPATCH(f, PFBWD_LAUNCH);
#undef PFBWD_LAUNCH
      MISEG_LAUNCH_CHECK("instnorm_pair_bwd");
      return MISEG_OK;
    }
    dim3 grid(g.chunks, p->B, g.ctiles);
    size_t sh = (size_t)g.ty * 2 * g.tx * g.vec * sizeof(float);
    if (sh < (size_t)4 * g.tx * g.vec * sizeof(double)) sh = (size_t)4 * g.tx * g.vec * sizeof(double);
    size_t shd = (size_t)8 * g.tx * g.vec * sizeof(double);
    if (p->r1x && shd < (size_t)NORM_THREADS * g.vec * sizeof(float)) shd = (size_t)NORM_THREADS * g.vec * sizeof(float);
#define PAIR_LAUNCH(VV)                                                                                                                                          \
    instnorm_pair_bwd_reduce_kernel<T, VV><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->xa, p->ldxa,         \
                                                                               (const T*)p->xb, p->ldxb, p->S, p->C, g.cv, g.tx, g.ty, g.rpb,                    \
                                                                               (const double*)p->stat_a, (const double*)p->stat_b, p->eps, p->slope,            \
                                                                               (double*)p->dstat_a, (double*)p->dstat_b, p->styles, spa, spb,                   \
                                                                               (const T*)p->r1x, p->ldr1x, (const T*)p->r1w);                                   \
    instnorm_pair_bwd_apply_kernel<T, VV><<<grid, NORM_THREADS, shd, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->xa, p->ldxa,         \
                                                                               (const T*)p->xb, p->ldxb, (T*)p->dxa, p->lddxa, (T*)p->dxb, p->lddxb, p->S, p->C, \
                                                                               g.cv, g.tx, g.ty, g.rpb, (const double*)p->stat_a, (const double*)p->stat_b,     \
                                                                               p->eps, p->styles, spa, spb, p->slope, (const double*)p->dstat_a,               \
                                                                               (const double*)p->dstat_b, gpa, gpb, (const T*)p->r1x, p->ldr1x,                 \
                                                                               (const T*)p->r1w, p->r1dw);
    if (g.vec == 1) { PAIR_LAUNCH(1) } else { PAIR_LAUNCH(V) }
#undef PAIR_LAUNCH
    MISEG_LAUNCH_CHECK("instnorm_pair_bwd");
    return MISEG_OK;
  });
}

extern "C" int miseg_layernorm_fwd(const miseg_layernorm_fwd_params* p, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->x && p->y && p->mean && p->rstd && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "layernorm_fwd: bad args");
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    layernorm_fwd_kernel<T><<<cdiv(p->rows, 4), 256, 0, stream>>>((const T*)p->x, p->ldx, (T*)p->y, p->ldy, p->rows, p->C, p->eps, p->gamma, p->beta, p->mean, p->rstd);
    MISEG_LAUNCH_CHECK("layernorm_fwd");
    return MISEG_OK;
  });
}

extern "C" int miseg_layernorm_bwd(const miseg_layernorm_bwd_params* p, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->dy && p->x && p->dx && p->mean && p->rstd && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "layernorm_bwd: bad args");
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    layernorm_bwd_kernel<T><<<cdiv(p->rows, 4), 256, 0, stream>>>((const T*)p->dy, p->lddy, (const T*)p->x, p->ldx, (T*)p->dx, p->lddx, p->rows, p->C, p->gamma,
                                                                  p->mean, p->rstd);
    if (p->dgamma || p->dbeta) {
      const int rpb = 256;
      layernorm_bwd_param_kernel<T><<<cdiv(p->rows, rpb), 256, 0, stream>>>((const T*)p->dy, p->lddy, (const T*)p->x, p->ldx, p->rows, p->C, p->mean, p->rstd,
                                                                           p->dgamma, p->dbeta, rpb);
    }
    MISEG_LAUNCH_CHECK("layernorm_bwd");
    return MISEG_OK;
  });
}
