// (Conditional) InstanceNorm and LayerNorm over channels-last rows -- HBM-bound kernels.
// Roofline: bytes only.  stats = 1 read of x; apply = 1 read (+1 residual) + 1 write;
// backward = reduce (dy, y, x reads) + apply (dy, y, x reads, dx [+dres] writes).
// Thread layout: tx owns one 16-byte channel vector for the whole block (its affine/statistics live in
// registers), ty walks rows, so every wave-instruction reads contiguous row segments.
#include "common.h"

namespace miseg {

static constexpr int NORM_THREADS = 256;
static constexpr int NORM_ROWS_PER_BLOCK = 1024;

struct NormGeom {
  int vec;      // elements per lane access (Vec16<T>::N or 1)
  int cv;       // channel vectors per row
  int tx, ty;   // thread grid inside a block
  int chunks;   // row chunks per sample
};

static NormGeom norm_geom(int S, int C, int64_t ld_all_or, const void* ptr_or, int vecN) {
  NormGeom g;
  bool vec_ok = (C % vecN == 0) && (ld_all_or % vecN == 0) && (((uintptr_t)ptr_or) % 16 == 0);
  g.vec = vec_ok ? vecN : 1;
  g.cv = C / g.vec;
  g.tx = g.cv < NORM_THREADS ? g.cv : NORM_THREADS;
  g.ty = NORM_THREADS / g.tx;
  g.chunks = cdiv(S, NORM_ROWS_PER_BLOCK);
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

// ---------------------------------------------------------------------------------------------------
// stats: partial (sum, sumsq) per (b, chunk, c) -> workspace; finalize in fp64 -> mean, rstd
// workspace layout: float ws[B][chunks][2][C]
// ---------------------------------------------------------------------------------------------------
template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_stats_kernel(const T* __restrict__ x, int64_t ldx, int S, int C,
                                                                      int cv, int tx_n, int ty_n, float* __restrict__ ws,
                                                                      int chunks) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [ty][2][tx*VEC] per cv-iteration
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int r0 = chunk * NORM_ROWS_PER_BLOCK;
  const int r1 = min(S, r0 + NORM_ROWS_PER_BLOCK);
  const T* xb = x + (int64_t)b * S * ldx;
  float* out = ws + ((int64_t)(b * chunks + chunk) * 2) * C;
  for (int c0 = 0; c0 < cv; c0 += tx_n) {
    const int c = c0 + tx;
    float s[VEC], q[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = q[i] = 0.f;
    if (ty < ty_n && c < cv) {
      for (int r = r0 + ty; r < r1; r += ty_n) {
        RowVec<T, VEC> v;
        v.load(xb + (int64_t)r * ldx + c * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s[i] += v.v[i]; q[i] = fmaf(v.v[i], v.v[i], q[i]); }
      }
    }
    // reduce over ty through LDS
    __syncthreads();
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
      if (ch < C) out[which * C + ch] = acc;
    }
  }
}

__global__ void instnorm_finalize_kernel(const float* __restrict__ ws, int chunks, int S, int C, float eps,
                                         float* __restrict__ mean, float* __restrict__ rstd, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (b, c)
  if (i >= total) return;
  const int b = i / C, c = i % C;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < chunks; ++k) {
    s += (double)ws[((int64_t)(b * chunks + k) * 2 + 0) * C + c];
    q += (double)ws[((int64_t)(b * chunks + k) * 2 + 1) * C + c];
  }
  const double m = s / S;
  double var = q / S - m * m;
  if (var < 0.0) var = 0.0;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---------------------------------------------------------------------------------------------------
// apply
// ---------------------------------------------------------------------------------------------------
struct StylePtrs {
  const float* gamma[MISEG_MAX_STYLES];
  const float* beta[MISEG_MAX_STYLES];
};

template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_apply_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ res,
                                                                      int64_t ldres, T* __restrict__ y, int64_t ldy, int S, int C, int cv,
                                                                      int tx_n, int ty_n, const float* __restrict__ mean,
                                                                      const float* __restrict__ rstd, const int32_t* __restrict__ styles,
                                                                      StylePtrs sp, int act, float slope) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  if (ty >= ty_n) return;
  const int r0 = chunk * NORM_ROWS_PER_BLOCK;
  const int r1 = min(S, r0 + NORM_ROWS_PER_BLOCK);
  const int st = styles ? styles[b] : 0;
  const float* g = sp.gamma[st];
  const float* be = sp.beta[st];
  const int64_t boff = (int64_t)b * S;
  for (int c = tx; c < cv; c += tx_n) {
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int ch = c * VEC + i;
      const float m = mean[b * C + ch], rs = rstd[b * C + ch];
      const float gg = g ? g[ch] : 1.f, bb = be ? be[ch] : 0.f;
      sc[i] = rs * gg;
      sh[i] = bb - m * sc[i];
    }
    for (int r = r0 + ty; r < r1; r += ty_n) {
      RowVec<T, VEC> v, o;
      v.load(x + (boff + r) * ldx + c * VEC);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.v[i] = fmaf(v.v[i], sc[i], sh[i]);
      if (res) {
        RowVec<T, VEC> rr;
        rr.load(res + (boff + r) * ldres + c * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.v[i] += rr.v[i];
      }
      if (act == MISEG_ACT_LEAKY) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.v[i] = o.v[i] > 0.f ? o.v[i] : o.v[i] * slope;
      }
      o.store(y + (boff + r) * ldy + c * VEC);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// backward: reduce  (sum dz, sum dz*xhat) -> ws ; finalize -> a,b per (b,c) + dgamma/dbeta atomics ; apply
// ---------------------------------------------------------------------------------------------------
template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_bwd_reduce_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ yact,
                                                                           int64_t ldy, const T* __restrict__ x, int64_t ldx, int S, int C, int cv,
                                                                           int tx_n, int ty_n, const float* __restrict__ mean,
                                                                           const float* __restrict__ rstd, int act, float slope,
                                                                           float* __restrict__ ws, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int r0 = chunk * NORM_ROWS_PER_BLOCK;
  const int r1 = min(S, r0 + NORM_ROWS_PER_BLOCK);
  const int64_t boff = (int64_t)b * S;
  float* out = ws + ((int64_t)(b * chunks + chunk) * 2) * C;
  for (int c0 = 0; c0 < cv; c0 += tx_n) {
    const int c = c0 + tx;
    float s[VEC], q[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = q[i] = 0.f;
    if (ty < ty_n && c < cv) {
      float m[VEC], rs[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) { m[i] = mean[b * C + c * VEC + i]; rs[i] = rstd[b * C + c * VEC + i]; }
      for (int r = r0 + ty; r < r1; r += ty_n) {
        RowVec<T, VEC> g, xv;
        g.load(dy + (boff + r) * lddy + c * VEC);
        xv.load(x + (boff + r) * ldx + c * VEC);
        if (act == MISEG_ACT_LEAKY) {
          RowVec<T, VEC> yv;
          yv.load(yact + (boff + r) * ldy + c * VEC);
#pragma unroll
          for (int i = 0; i < VEC; ++i) g.v[i] = yv.v[i] > 0.f ? g.v[i] : g.v[i] * slope;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          s[i] += g.v[i];
          q[i] = fmaf(g.v[i], (xv.v[i] - m[i]) * rs[i], q[i]);
        }
      }
    }
    __syncthreads();
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
      for (int yy = 0; yy < ty_n; ++yy) acc += red[(yy * 2 + which) * tx_n * VEC + col];
      const int ch = c0 * VEC + col;
      if (ch < C) out[which * C + ch] = acc;
    }
  }
}

struct StyleGradPtrs {
  float* dgamma[MISEG_MAX_STYLES];
  float* dbeta[MISEG_MAX_STYLES];
};

// writes the per-(b,c) means a = sum(dz)/S, bq = sum(dz*xhat)/S over ws[b][0][0..1][c] (in place) and
// accumulates dgamma / dbeta of the sample's style.
__global__ void instnorm_bwd_finalize_kernel(float* __restrict__ ws, int chunks, int S, int C, const int32_t* __restrict__ styles,
                                             StyleGradPtrs gp, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = i / C, c = i % C;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < chunks; ++k) {
    s += (double)ws[((int64_t)(b * chunks + k) * 2 + 0) * C + c];
    q += (double)ws[((int64_t)(b * chunks + k) * 2 + 1) * C + c];
  }
  const int st = styles ? styles[b] : 0;
  if (gp.dgamma[st]) atomicAdd(gp.dgamma[st] + c, (float)q);
  if (gp.dbeta[st]) atomicAdd(gp.dbeta[st] + c, (float)s);
  ws[((int64_t)(b * chunks) * 2 + 0) * C + c] = (float)(s / S);
  ws[((int64_t)(b * chunks) * 2 + 1) * C + c] = (float)(q / S);
}

template <class T, int VEC>
__global__ void __launch_bounds__(NORM_THREADS) instnorm_bwd_apply_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ yact,
                                                                          int64_t ldy, const T* __restrict__ x, int64_t ldx, T* __restrict__ dx,
                                                                          int64_t lddx, T* __restrict__ dres, int64_t lddres, int S, int C, int cv,
                                                                          int tx_n, int ty_n, const float* __restrict__ mean,
                                                                          const float* __restrict__ rstd, const int32_t* __restrict__ styles,
                                                                          StylePtrs sp, int act, float slope, const float* __restrict__ ws,
                                                                          int chunks) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  if (ty >= ty_n) return;
  const int r0 = chunk * NORM_ROWS_PER_BLOCK;
  const int r1 = min(S, r0 + NORM_ROWS_PER_BLOCK);
  const int st = styles ? styles[b] : 0;
  const float* g = sp.gamma[st];
  const int64_t boff = (int64_t)b * S;
  const float* ab = ws + ((int64_t)(b * chunks) * 2) * C;
  for (int c = tx; c < cv; c += tx_n) {
    float m[VEC], rs[VEC], sc[VEC], a[VEC], bq[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int ch = c * VEC + i;
      m[i] = mean[b * C + ch];
      rs[i] = rstd[b * C + ch];
      sc[i] = rs[i] * (g ? g[ch] : 1.f);
      a[i] = ab[ch];
      bq[i] = ab[C + ch];
    }
    for (int r = r0 + ty; r < r1; r += ty_n) {
      RowVec<T, VEC> gv, xv, o;
      gv.load(dy + (boff + r) * lddy + c * VEC);
      xv.load(x + (boff + r) * ldx + c * VEC);
      if (act == MISEG_ACT_LEAKY) {
        RowVec<T, VEC> yv;
        yv.load(yact + (boff + r) * ldy + c * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) gv.v[i] = yv.v[i] > 0.f ? gv.v[i] : gv.v[i] * slope;
      }
      if (dres) gv.store(dres + (boff + r) * lddres + c * VEC);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xh = (xv.v[i] - m[i]) * rs[i];
        o.v[i] = sc[i] * (gv.v[i] - a[i] - xh * bq[i]);
      }
      o.store(dx + (boff + r) * lddx + c * VEC);
    }
  }
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
  const float rs = rsqrtf(q / C + eps);
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

extern "C" size_t miseg_instnorm_workspace_bytes(int B, int S, int C) {
  return (size_t)B * cdiv(S, NORM_ROWS_PER_BLOCK) * 2 * C * sizeof(float);
}

extern "C" int miseg_instnorm_stats(const miseg_instnorm_stats_params* p, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->x && p->mean && p->rstd && p->workspace, MISEG_E_BADARG, "instnorm_stats: null pointer");
  MISEG_REQUIRE(p->B > 0 && p->S > 0 && p->C > 0 && p->ldx >= p->C, MISEG_E_BADARG, "instnorm_stats: bad shape B=%d S=%d C=%d ld=%ld",
                p->B, p->S, p->C, (long)p->ldx);
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    NormGeom g = norm_geom(p->S, p->C, p->ldx, p->x, Vec16<T>::N);
    dim3 grid(g.chunks, p->B);
    size_t sh = (size_t)g.ty * 2 * g.tx * g.vec * sizeof(float);
    if (g.vec == 1)
      instnorm_stats_kernel<T, 1><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->x, p->ldx, p->S, p->C, g.cv, g.tx, g.ty, (float*)p->workspace, g.chunks);
    else
      instnorm_stats_kernel<T, Vec16<T>::N><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->x, p->ldx, p->S, p->C, g.cv, g.tx, g.ty, (float*)p->workspace, g.chunks);
    const int total = p->B * p->C;
    instnorm_finalize_kernel<<<cdiv(total, 256), 256, 0, stream>>>((const float*)p->workspace, g.chunks, p->S, p->C, p->eps, p->mean, p->rstd, total);
    MISEG_LAUNCH_CHECK("instnorm_stats");
    return MISEG_OK;
  });
}

static bool aligned16(const void* p) { return ((uintptr_t)p % 16) == 0; }

extern "C" int miseg_instnorm_apply(const miseg_instnorm_apply_params* p, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->x && p->y && p->mean && p->rstd, MISEG_E_BADARG, "instnorm_apply: null pointer");
  MISEG_REQUIRE(p->num_styles >= 1 && p->num_styles <= MISEG_MAX_STYLES, MISEG_E_BADARG, "instnorm_apply: num_styles %d", p->num_styles);
  MISEG_REQUIRE(p->act == MISEG_ACT_NONE || p->act == MISEG_ACT_LEAKY, MISEG_E_UNSUPPORTED, "instnorm_apply: act %d", p->act);
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    int64_t ldor = p->ldx | p->ldy | (p->res ? p->ldres : 0);
    bool al = aligned16(p->x) && aligned16(p->y) && (!p->res || aligned16(p->res));
    NormGeom g = norm_geom(p->S, p->C, al ? ldor : 1, p->x, Vec16<T>::N);
    StylePtrs sp;
    for (int s = 0; s < MISEG_MAX_STYLES; ++s) { sp.gamma[s] = s < p->num_styles ? p->gamma[s] : nullptr; sp.beta[s] = s < p->num_styles ? p->beta[s] : nullptr; }
    dim3 grid(g.chunks, p->B);
    if (g.vec == 1)
      instnorm_apply_kernel<T, 1><<<grid, NORM_THREADS, 0, stream>>>((const T*)p->x, p->ldx, (const T*)p->res, p->ldres, (T*)p->y, p->ldy, p->S, p->C, g.cv,
                                                                      g.tx, g.ty, p->mean, p->rstd, p->styles, sp, p->act, p->slope);
    else
      instnorm_apply_kernel<T, Vec16<T>::N><<<grid, NORM_THREADS, 0, stream>>>((const T*)p->x, p->ldx, (const T*)p->res, p->ldres, (T*)p->y, p->ldy, p->S,
                                                                                p->C, g.cv, g.tx, g.ty, p->mean, p->rstd, p->styles, sp, p->act, p->slope);
    MISEG_LAUNCH_CHECK("instnorm_apply");
    return MISEG_OK;
  });
}

extern "C" int miseg_instnorm_bwd(const miseg_instnorm_bwd_params* p, miseg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MISEG_REQUIRE(p && p->dy && p->x && p->dx && p->mean && p->rstd && p->workspace, MISEG_E_BADARG, "instnorm_bwd: null pointer");
  MISEG_REQUIRE(p->act == MISEG_ACT_NONE || (p->act == MISEG_ACT_LEAKY && p->y), MISEG_E_BADARG, "instnorm_bwd: act %d needs y", p->act);
  MISEG_REQUIRE(p->num_styles >= 1 && p->num_styles <= MISEG_MAX_STYLES, MISEG_E_BADARG, "instnorm_bwd: num_styles %d", p->num_styles);
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    int64_t ldor = p->lddy | p->ldx | p->lddx | (p->act ? p->ldy : 0) | (p->dres ? p->lddres : 0);
    bool al = aligned16(p->dy) && aligned16(p->x) && aligned16(p->dx) && (!p->act || aligned16(p->y)) && (!p->dres || aligned16(p->dres));
    NormGeom g = norm_geom(p->S, p->C, al ? ldor : 1, p->x, Vec16<T>::N);
    StylePtrs sp;
    StyleGradPtrs gp;
    for (int s = 0; s < MISEG_MAX_STYLES; ++s) {
      sp.gamma[s] = s < p->num_styles ? p->gamma[s] : nullptr;
      sp.beta[s] = nullptr;
      gp.dgamma[s] = s < p->num_styles ? p->dgamma[s] : nullptr;
      gp.dbeta[s] = s < p->num_styles ? p->dbeta[s] : nullptr;
    }
    dim3 grid(g.chunks, p->B);
    size_t sh = (size_t)g.ty * 2 * g.tx * g.vec * sizeof(float);
    float* ws = (float*)p->workspace;
    const int total = p->B * p->C;
    if (g.vec == 1) {
      instnorm_bwd_reduce_kernel<T, 1><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->x, p->ldx, p->S, p->C,
                                                                            g.cv, g.tx, g.ty, p->mean, p->rstd, p->act, p->slope, ws, g.chunks);
      instnorm_bwd_finalize_kernel<<<cdiv(total, 256), 256, 0, stream>>>(ws, g.chunks, p->S, p->C, p->styles, gp, total);
      instnorm_bwd_apply_kernel<T, 1><<<grid, NORM_THREADS, 0, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->x, p->ldx, (T*)p->dx,
                                                                          p->lddx, (T*)p->dres, p->lddres, p->S, p->C, g.cv, g.tx, g.ty, p->mean, p->rstd,
                                                                          p->styles, sp, p->act, p->slope, ws, g.chunks);
    } else {
      constexpr int V = Vec16<T>::N;
      instnorm_bwd_reduce_kernel<T, V><<<grid, NORM_THREADS, sh, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->x, p->ldx, p->S, p->C,
                                                                            g.cv, g.tx, g.ty, p->mean, p->rstd, p->act, p->slope, ws, g.chunks);
      instnorm_bwd_finalize_kernel<<<cdiv(total, 256), 256, 0, stream>>>(ws, g.chunks, p->S, p->C, p->styles, gp, total);
      instnorm_bwd_apply_kernel<T, V><<<grid, NORM_THREADS, 0, stream>>>((const T*)p->dy, p->lddy, (const T*)p->y, p->ldy, (const T*)p->x, p->ldx, (T*)p->dx,
                                                                          p->lddx, (T*)p->dres, p->lddres, p->S, p->C, g.cv, g.tx, g.ty, p->mean, p->rstd,
                                                                          p->styles, sp, p->act, p->slope, ws, g.chunks);
    }
    MISEG_LAUNCH_CHECK("instnorm_bwd");
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
