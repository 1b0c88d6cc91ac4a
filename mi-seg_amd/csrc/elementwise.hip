// Data-movement and thin-channel kernels (all HBM-bound): add, copy/cast, GELU', 2x2x2 space<->channel,
// PatchEmbed (k2 s2), the 1-channel 3x3x3 stem conv, the 1x1x1 output head, im2col for tiny grids.
#include "common.h"

namespace miseg {

// ----------------------------------------------------------------------------- generic row-wise helpers
template <class T, int VEC, class F>
__global__ void __launch_bounds__(256) rowwise_kernel(int64_t rows, int cv, F f) {
  const int64_t total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    f(r, c);
  }
}

static inline int ew_grid(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

template <class T, int VEC> struct V {
  float v[VEC];
  __device__ __forceinline__ void load(const T* p) {
    if constexpr (VEC == 1) v[0] = to_f32(p[0]);
    else {
      typename Vec16<T>::type t = *reinterpret_cast<const typename Vec16<T>::type*>(p);
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = to_f32(t[i]);
    }
  }
  __device__ __forceinline__ void store(T* p) const {
    if constexpr (VEC == 1) p[0] = from_f32<T>(v[0]);
    else {
      typename Vec16<T>::type t;
#pragma unroll
      for (int i = 0; i < VEC; ++i) t[i] = from_f32<T>(v[i]);
      *reinterpret_cast<typename Vec16<T>::type*>(p) = t;
    }
  }
};

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ----------------------------------------------------------------------------- add
template <class T, int VEC>
__global__ void __launch_bounds__(256) add_kernel(const T* a, int64_t lda, const T* b, int64_t ldb, T* y, int64_t ldy, int64_t rows, int cv) {
  const int64_t total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    V<T, VEC> va, vb;
    va.load(a + r * lda + c);
    vb.load(b + r * ldb + c);
#pragma unroll
    for (int k = 0; k < VEC; ++k) va.v[k] += vb.v[k];
    va.store(y + r * ldy + c);
  }
}

// ----------------------------------------------------------------------------- copy2d with conversion
template <class TS, class TD>
__global__ void __launch_bounds__(256) copy2d_kernel(const TS* s, int64_t lds, TD* d, int64_t ldd, int64_t rows, int C) {
  const int64_t total = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int c = (int)(i % C);
    d[r * ldd + c] = from_f32<TD>(to_f32(s[r * lds + c]));
  }
}

// ----------------------------------------------------------------------------- cast (+transpose) fp32 matrix
template <class T>
__global__ void __launch_bounds__(256) cast_kernel(const float* s, T* d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = from_f32<T>(s[i]);
}
template <class T>
__global__ void __launch_bounds__(256) cast_transpose_kernel(const float* s, T* d, int R, int C) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8)
    if (by + j < R && bx + tx < C) tile[j][tx] = s[(int64_t)(by + j) * C + bx + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < C && by + tx < R) d[(int64_t)(bx + j) * R + by + tx] = from_f32<T>(tile[tx][j]);
}

// ----------------------------------------------------------------------------- GELU backward
template <class T, int VEC>
__global__ void __launch_bounds__(256) gelu_bwd_kernel(const T* dy, int64_t lddy, const T* x, int64_t ldx, T* dx, int64_t lddx, int64_t rows, int cv) {
  const int64_t total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    V<T, VEC> g, xv;
    g.load(dy + r * lddy + c);
    xv.load(x + r * ldx + c);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float t = xv.v[k];
      const float cdf = 0.5f * (1.f + erff(t * 0.70710678118654752f));
      const float pdf = 0.39894228040143268f * (sizeof(T) == 4 ? expf(-0.5f * t * t) : __expf(-0.5f * t * t));
      g.v[k] *= cdf + t * pdf;
    }
    g.store(dx + r * lddx + c);
  }
}

template <class T, int VEC>
__global__ void __launch_bounds__(256) gelu_fwd_kernel(const T* x, int64_t ldx, T* y, int64_t ldy, int64_t rows, int cv) {
  const int64_t total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    V<T, VEC> xv;
    xv.load(x + r * ldx + c);
#pragma unroll
    for (int k = 0; k < VEC; ++k) xv.v[k] = 0.5f * xv.v[k] * (1.f + erff(xv.v[k] * 0.70710678118654752f));
    xv.store(y + r * ldy + c);
  }
}

// ----------------------------------------------------------------------------- 2x2x2 space <-> channel
struct Off8 { int8_t o[24]; };

// IDX: the item index type.  64-bit division is a ~100-instruction software routine on the GPU and these kernels do 4-5 of them per
// 16-byte item: with fewer than 2^31 items (every shape of the reference configs) the launchers pick the 32-bit instantiation.
template <class T, int VEC, class IDX = int64_t>
__global__ void __launch_bounds__(256) s2c_kernel(const T* src, int64_t lds, T* dst, int64_t ldd, int B, int D, int H, int W, int C, Off8 off) {
  const int D2 = (D + 1) / 2, H2 = (H + 1) / 2, W2 = (W + 1) / 2, cv = C / VEC;
  const IDX total = (IDX)((int64_t)B * D2 * H2 * W2 * 8 * cv);
  for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    IDX t = i / cv;
    const int j = (int)(t % 8); t /= 8;
    const int w2 = (int)(t % W2); t /= W2;
    const int h2 = (int)(t % H2); t /= H2;
    const int d2 = (int)(t % D2);
    const int b = (int)(t / D2);
    const int d = 2 * d2 + off.o[3 * j], h = 2 * h2 + off.o[3 * j + 1], w = 2 * w2 + off.o[3 * j + 2];
    V<T, VEC> v;
    if (d < D && h < H && w < W) v.load(src + ((((int64_t)b * D + d) * H + h) * W + w) * lds + c);
    else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) v.v[k] = 0.f;
    }
    v.store(dst + ((((int64_t)b * D2 + d2) * H2 + h2) * W2 + w2) * ldd + (int64_t)j * C + c);
  }
}

template <class T, int VEC, class IDX = int64_t>
__global__ void __launch_bounds__(256) c2s_kernel(const T* src, int64_t lds, T* dst, int64_t ldd, int B, int D, int H, int W, int C, Off8 off) {
  const int D2 = (D + 1) / 2, H2 = (H + 1) / 2, W2 = (W + 1) / 2, cv = C / VEC;
  const IDX total = (IDX)((int64_t)B * D * H * W * cv);
  for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    IDX t = i / cv;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D);
    const int b = (int)(t / D);
    const int pd = d & 1, ph = h & 1, pw = w & 1;
    const T* srow = src + ((((int64_t)b * D2 + (d >> 1)) * H2 + (h >> 1)) * W2 + (w >> 1)) * lds + c;
    V<T, VEC> acc;
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc.v[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (off.o[3 * j] == pd && off.o[3 * j + 1] == ph && off.o[3 * j + 2] == pw) {
        V<T, VEC> v;
        v.load(srow + (int64_t)j * C);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc.v[k] += v.v[k];
      }
    }
    acc.store(dst + ((((int64_t)b * D + d) * H + h) * W + w) * ldd + c);
  }
}

// ----------------------------------------------------------------------------- PatchEmbed k2 s2 (+bias)
// thread = (coarse voxel, group of 8 output channels)
// IDX: the type of the item arithmetic - `unsigned` where the item count and the input's element count fit 31 bits (the 64-bit divisions of the
// item decode and the eight 64-bit address chains were most of this kernel's instructions: round 5)
template <class T, class IDX = int64_t>
__global__ void __launch_bounds__(256) patch_embed_fwd_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t ldy, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int B, int Cin, int D, int H, int W, int Cout) {
  extern __shared__ float ws[];  // [Cin*8][Cout] transposed weights + bias[Cout]
  const int K = Cin * 8;
  for (int i = threadIdx.x; i < K * Cout; i += blockDim.x) {
    const int co = i / K, k = i % K;
    ws[k * Cout + co] = w[i];
  }
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) ws[K * Cout + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int D2 = D / 2, H2 = H / 2, W2 = W / 2, cg = (Cout + 7) / 8;
  const IDX total = (IDX)((int64_t)B * D2 * H2 * W2 * cg);
  for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
    const int g = (int)(i % (IDX)cg);
    IDX t = i / (IDX)cg;
    const int w2 = (int)(t % (IDX)W2); t /= (IDX)W2;
    const int h2 = (int)(t % (IDX)H2); t /= (IDX)H2;
    const int d2 = (int)(t % (IDX)D2);
    const int b = (int)(t / (IDX)D2);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (g * 8 + k < Cout) ? ws[K * Cout + g * 8 + k] : 0.f;
    for (int ci = 0; ci < Cin; ++ci)
      for (int a = 0; a < 2; ++a)
        for (int bb = 0; bb < 2; ++bb)
          for (int c = 0; c < 2; ++c) {
            const float xv = x[((((IDX)b * Cin + ci) * D + 2 * d2 + a) * H + 2 * h2 + bb) * W + 2 * w2 + c];
            const float* wr = ws + ((ci * 2 + a) * 2 + bb) * 2 * Cout + c * Cout + g * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (g * 8 + k < Cout) acc[k] = fmaf(xv, wr[k], acc[k]);
          }
    T* yr = y + (int64_t)((((IDX)b * D2 + d2) * H2 + h2) * W2 + w2) * ldy + g * 8;
    // whole group, 16-byte aligned: one or two vector stores (round 5: the eight 2-byte stores per thread - eight partial-line writes per
    // wave and group - cost 3.3 of the kernel's 18.2 us in front of the Swin stages, where nothing runs beside it)
    if (g * 8 + 8 <= Cout && (reinterpret_cast<uintptr_t>(yr) & 15) == 0) {
      constexpr int N = Vec16<T>::N;
#pragma unroll
      for (int h = 0; h < 8 / N; ++h) {
        V<T, N> o;
#pragma unroll
        for (int k = 0; k < N; ++k) o.v[k] = acc[h * N + k];
        o.store(yr + h * N);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (g * 8 + k < Cout) yr[k] = from_f32<T>(acc[k]);
    }
  }
}

// dw[co][k] += sum_v dy[v][co] * xpatch[v][k],  dbias[co] += sum_v dy[v][co];   k in [0, Cin*8)
template <class T>
__global__ void __launch_bounds__(256) patch_embed_bwd_kernel(const float* __restrict__ x, const T* __restrict__ dy, int64_t lddy, float* __restrict__ dw,
                                                              float* __restrict__ dbias, int B, int Cin, int D, int H, int W, int Cout, int vox_per_block) {
  extern __shared__ float sm[];  // xs[VT][K+1], ds[VT][Cout]
  const int K = Cin * 8, VT = 64;
  float* xs = sm;
  float* ds = sm + VT * (K + 1);
  const int D2 = D / 2, H2 = H / 2, W2 = W / 2;
  const int64_t nv = (int64_t)B * D2 * H2 * W2;
  const int64_t v0 = (int64_t)blockIdx.x * vox_per_block, v1 = min(nv, v0 + vox_per_block);
  const int nout = Cout * (K + 1);
  float acc[8];  // outputs owned: o = threadIdx.x + j*256
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int64_t vb = v0; vb < v1; vb += VT) {
    const int nvt = (int)min((int64_t)VT, v1 - vb);
    __syncthreads();
    for (int i = threadIdx.x; i < nvt * (K + 1); i += blockDim.x) {
      const int vi = i / (K + 1), k = i % (K + 1);
      float val = 1.f;  // column K is the bias "input"
      if (k < K) {
        int64_t t = vb + vi;
        const int w2 = (int)(t % W2); t /= W2;
        const int h2 = (int)(t % H2); t /= H2;
        const int d2 = (int)(t % D2);
        const int b = (int)(t / D2);
        const int ci = k / 8, a = (k >> 2) & 1, bb = (k >> 1) & 1, c = k & 1;
        val = x[((((int64_t)b * Cin + ci) * D + 2 * d2 + a) * H + 2 * h2 + bb) * W + 2 * w2 + c];
      }
      xs[vi * (K + 1) + k] = val;
    }
    for (int i = threadIdx.x; i < nvt * Cout; i += blockDim.x) {
      const int vi = i / Cout, co = i % Cout;
      ds[vi * Cout + co] = to_f32(dy[(vb + vi) * lddy + co]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int o = threadIdx.x + j * 256;
      if (o < nout) {
        const int co = o / (K + 1), k = o % (K + 1);
        float a = acc[j];
        for (int vi = 0; vi < nvt; ++vi) a = fmaf(ds[vi * Cout + co], xs[vi * (K + 1) + k], a);
        acc[j] = a;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int o = threadIdx.x + j * 256;
    if (o < nout) {
      const int co = o / (K + 1), k = o % (K + 1);
      if (k < K) atomicAdd(dw + co * K + k, acc[j]);
      else if (dbias) atomicAdd(dbias + co, acc[j]);
    }
  }
}

// ----------------------------------------------------------------------------- thin 3x3x3 stem conv (Cin <= 4, NCDHW fp32 in)
// thread = voxel; weights transposed in LDS as [ci*27+tap][Cout]; loops output channels in groups of 8.
template <class T>
__global__ void __launch_bounds__(256) conv3_thin_fwd_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t ldy, const float* __restrict__ w, int B,
                                                             int Cin, int D, int H, int W, int Cout) {
  extern __shared__ float ws[];  // [Cin*27][Cout]
  const int K = Cin * 27;
  for (int i = threadIdx.x; i < K * Cout; i += blockDim.x) {
    const int co = i / K, k = i % K;
    ws[k * Cout + co] = w[i];
  }
  __syncthreads();
  const int64_t nv = (int64_t)B * D * H * W;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = v;
    const int wq = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D);
    const int b = (int)(t / D);
    for (int c0 = 0; c0 < Cout; c0 += 8) {
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.f;
      for (int ci = 0; ci < Cin; ++ci) {
        const float* xc = x + ((int64_t)b * Cin + ci) * D * H * W;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
          const int dd = d + tap / 9 - 1, hh = h + (tap / 3) % 3 - 1, ww = wq + tap % 3 - 1;
          float xv = 0.f;
          if (dd >= 0 && dd < D && hh >= 0 && hh < H && ww >= 0 && ww < W) xv = xc[((int64_t)dd * H + hh) * W + ww];
          const float* wr = ws + (ci * 27 + tap) * Cout + c0;
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] = fmaf(xv, (c0 + k < Cout) ? wr[k] : 0.f, acc[k]);
        }
      }
      T* yr = y + v * ldy + c0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (c0 + k < Cout) yr[k] = from_f32<T>(acc[k]);
    }
  }
}

// dw[co][ci][tap] += sum_v dy[v][co] x[ci][v+tap].  Block = 8x8x8 voxel brick; x halo + dy brick staged in LDS;
// thread = (tap, group of 8 output channels).
template <class T>
__global__ void __launch_bounds__(256) conv3_thin_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy, int64_t lddy, float* __restrict__ dw,
                                                               int B, int Cin, int D, int H, int W, int Cout) {
  extern __shared__ float sm[];
  constexpr int BT = 8, HT = BT + 2;
  float* xs = sm;                       // [HT^3]
  float* ds = sm + HT * HT * HT;        // [512][Cout]
  const int nbw = cdiv(W, BT), nbh = cdiv(H, BT), nbd = cdiv(D, BT);
  int bid = blockIdx.x;
  const int bw = bid % nbw; bid /= nbw;
  const int bh = bid % nbh; bid /= nbh;
  const int bd = bid % nbd;
  const int b = bid / nbd;
  const int d0 = bd * BT, h0 = bh * BT, w0 = bw * BT;
  for (int i = threadIdx.x; i < BT * BT * BT * Cout; i += blockDim.x) {
    const int co = i % Cout, vi = i / Cout;
    const int d = d0 + vi / 64, h = h0 + (vi / 8) % 8, w = w0 + vi % 8;
    float val = 0.f;
    if (d < D && h < H && w < W) val = to_f32(dy[((((int64_t)b * D + d) * H + h) * W + w) * lddy + co]);
    ds[vi * Cout + co] = val;
  }
  const int ncg = (Cout + 7) / 8;
  const int tap = threadIdx.x % 27, cg = threadIdx.x / 27;
  for (int ci = 0; ci < Cin; ++ci) {
    __syncthreads();
    for (int i = threadIdx.x; i < HT * HT * HT; i += blockDim.x) {
      const int d = d0 - 1 + i / (HT * HT), h = h0 - 1 + (i / HT) % HT, w = w0 - 1 + i % HT;
      float val = 0.f;
      if (d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W) val = x[((((int64_t)b * Cin + ci) * D + d) * H + h) * W + w];
      xs[i] = val;
    }
    __syncthreads();
    for (int g = cg; g < ncg; g += 256 / 27) {
      if (cg >= 256 / 27) break;
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.f;
      const int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
      for (int vi = 0; vi < 512; ++vi) {
        const int vd = vi >> 6, vh = (vi >> 3) & 7, vw = vi & 7;
        const float xv = xs[((vd + td) * HT + vh + th) * HT + vw + tw];
        const float* dr = ds + vi * Cout + g * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fmaf(xv, (g * 8 + k < Cout) ? dr[k] : 0.f, acc[k]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (g * 8 + k < Cout) atomicAdd(dw + ((int64_t)(g * 8 + k) * Cin + ci) * 27 + tap, acc[k]);
    }
  }
}

// ----------------------------------------------------------------------------- one-channel stem conv, brick form
// Cin == 1 (the CT / MR image): 27 x Cout multiply-adds per voxel - not matrix-core shaped (K = 27), so it stays on the VALU, but
// blocked like the implicit-GEMM kernels: a 4x4x16 voxel brick, its 6x6x18 halo of the fp32 image in LDS.
static constexpr int SB_D = 4, SB_H = 4, SB_W = 16, SH_H = SB_H + 2, SH_W = SB_W + 2, SH_N = (SB_D + 2) * SH_H * SH_W;   // 648

// forward: thread = voxel, its 27 taps in registers, weights [tap][Cout] as LDS broadcasts, 8 output channels per pass
// (two voxels per thread sharing the weight reads: 87 -> 107 us)
template <class T>
__global__ void __launch_bounds__(256) conv3_stem_fwd_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t ldy, const float* __restrict__ w, int B, int D,
                                                             int H, int W, int Cout, int nbricks) {
  __shared__ __attribute__((aligned(16))) float ws[27 * 64];
  __shared__ float xs[SH_N];
  const int tid = threadIdx.x;
  for (int i = tid; i < 27 * Cout; i += 256) {
    const int tap = i / Cout, co = i - tap * Cout;
    ws[i] = w[co * 27 + tap];
  }
  const int nbw = cdiv(W, SB_W), nbh = cdiv(H, SB_H), nbd = cdiv(D, SB_D);
  const int vd = tid >> 6, vh = (tid >> 4) & 3, vw = tid & 15;
  for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
    int bid = brick;
    const int bw = bid % nbw; bid /= nbw;
    const int bh = bid % nbh; bid /= nbh;
    const int bd = bid % nbd;
    const int b = bid / nbd;
    const int d0 = bd * SB_D, h0 = bh * SB_H, w0 = bw * SB_W;
    __syncthreads();
    for (int i = tid; i < SH_N; i += 256) {
      const int hd = i / (SH_H * SH_W), rem = i - hd * (SH_H * SH_W), hh = rem / SH_W, hw = rem - hh * SH_W;
      const int d = d0 - 1 + hd, h = h0 - 1 + hh, ww = w0 - 1 + hw;
      xs[i] = (d >= 0 && d < D && h >= 0 && h < H && ww >= 0 && ww < W) ? x[(((int64_t)b * D + d) * H + h) * W + ww] : 0.f;
    }
    __syncthreads();
    float xt[27];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) xt[tap] = xs[((vd + tap / 9) * SH_H + vh + (tap / 3) % 3) * SH_W + vw + tap % 3];
    const int d = d0 + vd, h = h0 + vh, ww = w0 + vw;
    if (d < D && h < H && ww < W) {
      T* yr = y + ((((int64_t)b * D + d) * H + h) * W + ww) * ldy;
      for (int c0 = 0; c0 < Cout; c0 += 8) {
        // weights straight from memory at wave-uniform addresses: scalar loads, the multiply-adds take them as SGPR operands
        // (as LDS broadcasts the 54 reads per pass cost as much as the 216 multiply-adds)
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = 0.f;
        const float* wc = w + c0 * 27;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int tap = 0; tap < 27; ++tap) a[k] = fmaf(xt[tap], wc[k * 27 + tap], a[k]);
        const f32x4 a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]};
        if constexpr (std::is_same<T, float>::value) {
          *reinterpret_cast<f32x4*>(yr + c0) = a0;
          *reinterpret_cast<f32x4*>(yr + c0 + 4) = a1;
        } else {
          *reinterpret_cast<bf16x8*>(yr + c0) = bf16x8{(bf16)a0[0], (bf16)a0[1], (bf16)a0[2], (bf16)a0[3], (bf16)a1[0], (bf16)a1[1], (bf16)a1[2], (bf16)a1[3]};
        }
      }
    }
  }
}

// weight gradient: dw[co][tap] += sum_v dy[v][co] x[v + tap].  thread = (kd, kh, 8 output channels, depth slice of the brick): one
// 18-float halo row serves the three kw taps of 16 voxels (384 multiply-adds per 5 + 16 LDS reads); workgroups are persistent and
// keep their 3 x 8 sums in registers over all their bricks.
template <class T>
__global__ void __launch_bounds__(256) conv3_stem_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy, int64_t lddy, float* __restrict__ dw, int B,
                                                               int D, int H, int W, int Cout, int nbricks) {
  extern __shared__ __attribute__((aligned(16))) char stem_sm[];
  char* sm = stem_sm;
  float* xs = reinterpret_cast<float*>(sm);                  // [648] (+ pad to 16 B)
  T* ds = reinterpret_cast<T*>(sm + 2608);                   // [256 voxels][Cout]
  const int tid = threadIdx.x, ncg = Cout / 8;
  const int cg = tid % ncg, r = tid / ncg;
  const int th = r % 3, td = (r / 3) % 3, vd = r / 9;
  const bool worker = vd < SB_D;
  float acc[3][8];          // (packed f32x2 accumulators were tried: 156 -> 229 us)
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[t][k] = 0.f;
  const int nbw = cdiv(W, SB_W), nbh = cdiv(H, SB_H), nbd = cdiv(D, SB_D);
  constexpr int VN = Vec16<T>::N;
  const int cvn = Cout / VN;
  for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
    int bid = brick;
    const int bw = bid % nbw; bid /= nbw;
    const int bh = bid % nbh; bid /= nbh;
    const int bd = bid % nbd;
    const int b = bid / nbd;
    const int d0 = bd * SB_D, h0 = bh * SB_H, w0 = bw * SB_W;
    __syncthreads();
    for (int i = tid; i < SH_N; i += 256) {
      const int hd = i / (SH_H * SH_W), rem = i - hd * (SH_H * SH_W), hh = rem / SH_W, hw = rem - hh * SH_W;
      const int d = d0 - 1 + hd, h = h0 - 1 + hh, ww = w0 - 1 + hw;
      xs[i] = (d >= 0 && d < D && h >= 0 && h < H && ww >= 0 && ww < W) ? x[(((int64_t)b * D + d) * H + h) * W + ww] : 0.f;
    }
    for (int i = tid; i < 256 * cvn; i += 256) {
      const int vi = i / cvn, c = (i - vi * cvn) * VN;
      const int d = d0 + (vi >> 6), h = h0 + ((vi >> 4) & 3), ww = w0 + (vi & 15);
      typename Vec16<T>::type v;
#pragma unroll
      for (int e = 0; e < VN; ++e) v[e] = from_f32<T>(0.f);
      if (d < D && h < H && ww < W) v = *reinterpret_cast<const typename Vec16<T>::type*>(dy + ((((int64_t)b * D + d) * H + h) * W + ww) * lddy + c);
      *reinterpret_cast<typename Vec16<T>::type*>(ds + vi * Cout + c) = v;
    }
    __syncthreads();
    if (worker) {
#pragma unroll 1
      for (int vh = 0; vh < SB_H; ++vh) {
        float xr[SH_W];
        const float* xrow = xs + ((vd + td) * SH_H + vh + th) * SH_W;
#pragma unroll
        for (int i = 0; i < SH_W; ++i) xr[i] = xrow[i];
        const T* drow = ds + ((vd * SB_H + vh) * SB_W) * Cout + cg * 8;
#pragma unroll
        for (int vw = 0; vw < SB_W; ++vw) {
          float g[8];
          if constexpr (std::is_same<T, float>::value) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(drow + vw * Cout), g1 = *reinterpret_cast<const f32x4*>(drow + vw * Cout + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { g[k] = g0[k]; g[4 + k] = g1[k]; }
          } else {
            const bf16x8 gv = *reinterpret_cast<const bf16x8*>(drow + vw * Cout);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = (float)gv[k];
          }
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[t][k] = fmaf(xr[vw + t], g[k], acc[t][k]);
        }
      }
    }
  }
  // the four depth slices meet in LDS, then one atomic per (co, tap) per workgroup
  __syncthreads();
  float* red = reinterpret_cast<float*>(sm);                 // [256][24]
  if (worker) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 8; ++k) red[tid * 24 + t * 8 + k] = acc[t][k];
  }
  __syncthreads();
  for (int o = tid; o < 9 * ncg * 24; o += 256) {
    const int q = o / 24, e = o - q * 24;          // q = (td * 3 + th) * ncg + cg  (the r, cg pair without the depth slice)
    const int qcg = q % ncg, qr = q / ncg;
    float a = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < SB_D; ++s4) a += red[((s4 * 9 + qr) * ncg + qcg) * 24 + e];
    const int t = e / 8, k = e - t * 8;
    const int qth = qr % 3, qtd = qr / 3;
    atomicAdd(dw + (int64_t)(qcg * 8 + k) * 27 + qtd * 9 + qth * 3 + t, a);
  }
}

// forward of the one-channel stem conv on the matrix cores (bf16, Cout == 48): y[v][co] = sum_tap X[v][tap] w[co][tap] is ONE k-step
// of 32 (27 taps + padding) per 16-voxel tile: 12 MFMAs per wave and brick.  The weight fragments live in registers for the whole
// kernel; a lane builds its X fragment from 8 halo reads (its k-group's 8 taps of voxel w = fi).  Output layout as in the
// implicit-GEMM kernel: lane holds channels 16nt + 4fq .. +3 of voxel (d0 + wave, h0 + mt, w0 + fi).
__global__ void __launch_bounds__(256) conv3_stem_fwd_mfma_kernel(const float* __restrict__ x, bf16* __restrict__ y, int64_t ldy, const float* __restrict__ w, int B,
                                                                  int D, int H, int W, int nbricks) {
  __shared__ float xs[SH_N];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fq = lane >> 4;
  bf16x8 wf[3];
  int toff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int tap = 8 * fq + j;
    toff[j] = tap < 27 ? ((tap / 9) * SH_H + (tap / 3) % 3) * SH_W + tap % 3 : -1;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) wf[nt][j] = (bf16)(tap < 27 ? w[(nt * 16 + fi) * 27 + tap] : 0.f);
  }
  const int nbw = cdiv(W, SB_W), nbh = cdiv(H, SB_H), nbd = cdiv(D, SB_D);
  for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
    int bid = brick;
    const int bw = bid % nbw; bid /= nbw;
    const int bh = bid % nbh; bid /= nbh;
    const int bd = bid % nbd;
    const int b = bid / nbd;
    const int d0 = bd * SB_D, h0 = bh * SB_H, w0 = bw * SB_W;
    __syncthreads();
    for (int i = tid; i < SH_N; i += 256) {
      const int hd = i / (SH_H * SH_W), rem = i - hd * (SH_H * SH_W), hh = rem / SH_W, hw = rem - hh * SH_W;
      const int d = d0 - 1 + hd, h = h0 - 1 + hh, ww = w0 - 1 + hw;
      xs[i] = (d >= 0 && d < D && h >= 0 && h < H && ww >= 0 && ww < W) ? x[(((int64_t)b * D + d) * H + h) * W + ww] : 0.f;
    }
    __syncthreads();
    const int d = d0 + wave, ww = w0 + fi;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float* xp = xs + (wave * SH_H + mt) * SH_W + fi;
      bf16x8 xf;
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = (bf16)(toff[j] < 0 ? 0.f : xp[toff[j] < 0 ? 0 : toff[j]]);
      f32x4 acc[3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const int h = h0 + mt;
      if (d < D && h < H && ww < W) {
        bf16* yr = y + ((((int64_t)b * D + d) * H + h) * W + ww) * ldy + fq * 4;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) *reinterpret_cast<bf16x4*>(yr + nt * 16) = bf16x4{(bf16)acc[nt][0], (bf16)acc[nt][1], (bf16)acc[nt][2], (bf16)acc[nt][3]};
      }
    }
  }
}

// weight gradient of the one-channel stem conv on the matrix cores (bf16): dw[co][tap] = sum_v dy[v][co] X[v][tap] with X the
// im2col of the image (27 -> 32 columns), i.e. per 4x4x16 brick 8 k-steps of 32 voxels x (3 co tiles x 2 tap tiles) = 48 MFMAs
// instead of 331 K multiply-adds.  The dy^T operand comes from the staged brick by transposed LDS reads (as in the 3x3x3
// weight-gradient kernel), the X operand is 8 consecutive floats of a halo row per lane (k-group = (h row, w half), element = w).
// Each wave takes two of the eight k-steps; the waves' tiles meet in LDS once per workgroup.  Needs Cout == 48.
// out[e] += sum over b of partial[b][e]: the second launch of the kernels whose workgroups leave partial sums instead of adding fp32 atomics
// onto a small output (a workgroup = 32 consecutive elements x 8 slices of the workgroup list)
__global__ void __launch_bounds__(256) partial_sum_add_kernel(const float* __restrict__ partial, float* __restrict__ out, int n, int nblk) {
  __shared__ float red[8][32];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  float a = 0.f;
  if (e < n) {
#pragma unroll 8
    for (int b = sl; b < nblk; b += 8) a += partial[(int64_t)b * n + e];
  }
  red[sl][threadIdx.x & 31] = a;
  __syncthreads();
  if (sl == 0 && e < n) {
#pragma unroll
    for (int i = 1; i < 8; ++i) a += red[i][threadIdx.x];
    out[e] += a;
  }
}

__global__ void __launch_bounds__(256) conv3_stem_wgrad_mfma_kernel(const float* __restrict__ x, const bf16* __restrict__ dy, int64_t lddy, float* __restrict__ dw,
                                                                    int B, int D, int H, int W, int nbricks, float* __restrict__ partial) {
  constexpr int Cout = 48, ROWB = Cout * 2 + 16;
  __shared__ __attribute__((aligned(16))) char ds[256 * ROWB];      // dy brick, padded rows; reused for the final reduction
  __shared__ float xs[SH_N];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fq = lane >> 4, qq = fi >> 2, p4 = (fi & 3) * 4;
  f32x4 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this lane's tap per tap tile (n index = fi): halo displacement, or -1 for the padding columns 27..31
  int toff[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int tap = nt * 16 + fi;
    toff[nt] = tap < 27 ? ((tap / 9) * SH_H + (tap / 3) % 3) * SH_W + tap % 3 : -1;
  }
  const int nbw = cdiv(W, SB_W), nbh = cdiv(H, SB_H), nbd = cdiv(D, SB_D);
  for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
    int bid = brick;
    const int bw = bid % nbw; bid /= nbw;
    const int bh = bid % nbh; bid /= nbh;
    const int bd = bid % nbd;
    const int b = bid / nbd;
    const int d0 = bd * SB_D, h0 = bh * SB_H, w0 = bw * SB_W;
    __syncthreads();
    for (int i = tid; i < SH_N; i += 256) {
      const int hd = i / (SH_H * SH_W), rem = i - hd * (SH_H * SH_W), hh = rem / SH_W, hw = rem - hh * SH_W;
      const int d = d0 - 1 + hd, h = h0 - 1 + hh, ww = w0 - 1 + hw;
      xs[i] = (d >= 0 && d < D && h >= 0 && h < H && ww >= 0 && ww < W) ? x[(((int64_t)b * D + d) * H + h) * W + ww] : 0.f;
    }
    for (int i = tid; i < 256 * 6; i += 256) {
      const int vi = i / 6, c = (i - vi * 6) * 8;
      const int d = d0 + (vi >> 6), h = h0 + ((vi >> 4) & 3), ww = w0 + (vi & 15);
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
      if (d < D && h < H && ww < W) v = *reinterpret_cast<const bf16x8*>(dy + ((((int64_t)b * D + d) * H + h) * W + ww) * lddy + c);
      *reinterpret_cast<bf16x8*>(ds + vi * ROWB + c * 2) = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ks = wave + 4 * kk;                                    // k-step: depth ks >> 1, h rows 2 (ks & 1) .. +1
      const int vd = ks >> 1, vh = 2 * (ks & 1) + (fq >> 1), wh = 8 * (fq & 1);
      const int vrow = (vd * SB_H + vh) * SB_W + wh + qq;
      bf16x8 af[3], bfr[2];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        const char* a1 = ds + vrow * ROWB + (mt * 16 + p4) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1 + 4 * ROWB));
        af[mt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float* xp = xs + (vd * SH_H + vh) * SH_W + wh + (toff[nt] < 0 ? 0 : toff[nt]);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)(toff[nt] < 0 ? 0.f : xp[e]);
        bfr[nt] = v;
      }
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
    }
  }
  // lane holds tap = 16nt + 4fq + r, co = 16mt + fi
  __syncthreads();
  float* red = reinterpret_cast<float*>(ds);                  // [4 waves][48 co][32 taps]
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      *reinterpret_cast<f32x4*>(red + ((wave * 48 + mt * 16 + fi) * 32 + nt * 16 + fq * 4)) = acc[mt][nt];
  __syncthreads();
  for (int o = tid; o < 48 * 27; o += 256) {
    const int co = o / 27, tap = o - co * 27;
    const float v = red[(0 * 48 + co) * 32 + tap] + red[(1 * 48 + co) * 32 + tap] + red[(2 * 48 + co) * 32 + tap] + red[(3 * 48 + co) * 32 + tap];
    if (partial) partial[(int64_t)blockIdx.x * (48 * 27) + o] = v;      // summed by partial_sum_add_kernel
    else atomicAdd(dw + o, v);
  }
}

// ----------------------------------------------------------------------------- output head (1x1x1 + bias -> NCDHW fp32)
// thread = voxel: the channels-last row is read once in 16-byte pieces; weights [Cout][Cin] are LDS broadcasts.
template <class T, int VEC>
__global__ void __launch_bounds__(256) head_fwd_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ y, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int B, int S, int Cin, int Cout) {
  extern __shared__ float ws[];  // [Cout][Cin] + bias
  for (int i = threadIdx.x; i < Cout * Cin; i += blockDim.x) ws[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) ws[Cout * Cin + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int64_t nv = (int64_t)B * S;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(v / S);
    const int64_t s = v % S;
    const T* xr = x + v * ldx;
    float acc[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = co < Cout ? ws[Cout * Cin + co] : 0.f;
    for (int c0 = 0; c0 < Cin; c0 += VEC) {
      V<T, VEC> xv;
      xv.load(xr + c0);
#pragma unroll
      for (int co = 0; co < 16; ++co) {
        if (co < Cout) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[co] = fmaf(xv.v[k], ws[co * Cin + c0 + k], acc[co]);
        }
      }
    }
#pragma unroll
    for (int co = 0; co < 16; ++co)
      if (co < Cout) y[((int64_t)b * Cout + co) * S + s] = acc[co];
  }
}

template <class T, int VEC>
__global__ void __launch_bounds__(256) head_bwd_dx_kernel(const float* __restrict__ dy, T* __restrict__ dx, int64_t lddx, const float* __restrict__ w, int B, int S,
                                                          int Cin, int Cout) {
  extern __shared__ float ws[];  // [Cout][Cin]
  for (int i = threadIdx.x; i < Cout * Cin; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const int64_t nv = (int64_t)B * S;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(v / S);
    const int64_t s = v % S;
    float g[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) g[co] = co < Cout ? dy[((int64_t)b * Cout + co) * S + s] : 0.f;
    T* dr = dx + v * lddx;
    for (int c0 = 0; c0 < Cin; c0 += VEC) {
      V<T, VEC> o;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int co = 0; co < 16; ++co)
          if (co < Cout) acc = fmaf(g[co], ws[co * Cin + c0 + k], acc);
        o.v[k] = acc;
      }
      o.store(dr + c0);
    }
  }
}

// dw[co][ci] += sum_v dy[co][v] x[v][ci] ; dbias[co] += sum_v dy[co][v].  thread = (co, ci or bias column)
template <class T>
__global__ void __launch_bounds__(256) head_bwd_dw_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ dy, float* __restrict__ dw,
                                                          float* __restrict__ dbias, int B, int S, int Cin, int Cout, int vox_per_block) {
  extern __shared__ float sm[];
  constexpr int VT = 64;
  float* xs = sm;                  // [VT][Cin+1]
  float* ds = sm + VT * (Cin + 1); // [Cout][VT]
  const int64_t nv = (int64_t)B * S;
  const int64_t v0 = (int64_t)blockIdx.x * vox_per_block, v1 = min(nv, v0 + vox_per_block);
  const int nout = Cout * (Cin + 1);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t vb = v0; vb < v1; vb += VT) {
    const int nvt = (int)min((int64_t)VT, v1 - vb);
    __syncthreads();
    for (int i = threadIdx.x; i < nvt * (Cin + 1); i += blockDim.x) {
      const int vi = i / (Cin + 1), ci = i % (Cin + 1);
      xs[i] = ci < Cin ? to_f32(x[(vb + vi) * ldx + ci]) : 1.f;
    }
    for (int i = threadIdx.x; i < nvt * Cout; i += blockDim.x) {
      const int co = i / nvt, vi = i % nvt;
      const int64_t v = vb + vi;
      ds[co * VT + vi] = dy[((v / S) * Cout + co) * S + v % S];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = threadIdx.x + j * 256;
      if (o < nout) {
        const int co = o / (Cin + 1), ci = o % (Cin + 1);
        float a = acc[j];
        for (int vi = 0; vi < nvt; ++vi) a = fmaf(ds[co * VT + vi], xs[vi * (Cin + 1) + ci], a);
        acc[j] = a;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = threadIdx.x + j * 256;
    if (o < nout) {
      const int co = o / (Cin + 1), ci = o % (Cin + 1);
      if (ci < Cin) atomicAdd(dw + co * Cin + ci, acc[j]);
      else if (dbias) atomicAdd(dbias + co, acc[j]);
    }
  }
}

// ----------------------------------------------------------------------------- head / patch-embed, row-team kernels
// Vector variants of the four kernels above for channel counts that are multiples of the 16-byte vector: a voxel row is
// read by Cin/VEC lanes with one 16-byte load each (coalesced), weights come from LDS as float4, and the reductions over
// voxels keep [R][VEC] register accumulators per lane that meet in LDS once per workgroup.

// sum over ty of acc[R][VEC] -> atomicAdd(out[r * ostride + (tx * VEC + i)]); red: [ty_n][R * tx_n * VEC] floats
template <int R, int VEC>
__device__ __forceinline__ void team_reduce_atomic(float* red, const float (&acc)[R][VEC], int tx, int ty, int tx_n, int ty_n, float* out, int ostride, int nrows,
                                                   int ncols) {
  const int width = R * tx_n * VEC;
  __syncthreads();
  if (ty < ty_n) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[ty * width + (r * tx_n + tx) * VEC + i] = acc[r][i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < width; e += blockDim.x) {
    float a = 0.f;
    for (int y = 0; y < ty_n; ++y) a += red[y * width + e];
    const int r = e / (tx_n * VEC), col = e - r * (tx_n * VEC);
    if (r < nrows && col < ncols) atomicAdd(out + r * ostride + col, a);
  }
}

// y[b][co][s] = sum_ci x[v][ci] w[co][ci] + bias[co]: 256-voxel tiles staged in LDS (padded rows), one lane per voxel
template <class T, int VEC>
__global__ void __launch_bounds__(256) head_fwd_tile_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ y, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int B, int S, int Cin, int Cout) {
  typedef typename Vec16<T>::type VT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cv = Cin / VEC, rowb = Cin * (int)sizeof(T) + 16;
  float* ws = reinterpret_cast<float*>(smem);          // [Cout][Cin] + bias[Cout]
  char* xt = smem + ((Cout * Cin + Cout + 3) & ~3) * 4; // [256][rowb]
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) ws[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += 256) ws[Cout * Cin + i] = bias ? bias[i] : 0.f;
  const int64_t nv = (int64_t)B * S;
  for (int64_t v0 = (int64_t)blockIdx.x * 256; v0 < nv; v0 += (int64_t)gridDim.x * 256) {
    __syncthreads();
    for (int i = threadIdx.x; i < 256 * cv; i += 256) {
      const int r = i / cv, c = i - r * cv;
      if (v0 + r < nv) *reinterpret_cast<VT*>(xt + r * rowb + c * 16) = *reinterpret_cast<const VT*>(x + (v0 + r) * ldx + c * VEC);
    }
    __syncthreads();
    const int64_t v = v0 + threadIdx.x;
    if (v < nv) {
      float acc[16];
#pragma unroll
      for (int co = 0; co < 16; ++co) acc[co] = co < Cout ? ws[Cout * Cin + co] : 0.f;
      for (int c = 0; c < cv; ++c) {
        const VT xv = *reinterpret_cast<const VT*>(xt + threadIdx.x * rowb + c * 16);
        float xf[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) xf[k] = to_f32(xv[k]);
#pragma unroll
        for (int co = 0; co < 16; ++co) {
          if (co < Cout) {
#pragma unroll
            for (int k4 = 0; k4 < VEC / 4; ++k4) {
              const f32x4 w4 = *reinterpret_cast<const f32x4*>(ws + co * Cin + c * VEC + 4 * k4);
              acc[co] = fmaf(xf[4 * k4 + 0], w4[0], acc[co]); acc[co] = fmaf(xf[4 * k4 + 1], w4[1], acc[co]);
              acc[co] = fmaf(xf[4 * k4 + 2], w4[2], acc[co]); acc[co] = fmaf(xf[4 * k4 + 3], w4[3], acc[co]);
            }
          }
        }
      }
      const int b = (int)(v / S);
      const int64_t sidx = v - (int64_t)b * S;
#pragma unroll
      for (int co = 0; co < 16; ++co)
        if (co < Cout) y[((int64_t)b * Cout + co) * S + sidx] = acc[co];
    }
  }
}

// dx[v][ci] = sum_co dy[b][co][s] w[co][ci]: one lane per (voxel, 16-byte channel vector)
template <class T, int VEC, class IDX = int64_t>
__global__ void __launch_bounds__(256) head_bwd_dx_team_kernel(const float* __restrict__ dy, T* __restrict__ dx, int64_t lddx, const float* __restrict__ w, int B, int S,
                                                               int Cin, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float ws[];  // [Cout][Cin]
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) ws[i] = w[i];
  __syncthreads();
  const int cv = Cin / VEC;
  const IDX total = (IDX)((int64_t)B * S * cv), stride = (IDX)gridDim.x * 256;
  constexpr int U = 4;      // items per trip: their dy loads are all in flight before the first FMA
  for (IDX i0 = (IDX)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += U * stride) {
    float g[U][16];
    int64_t vv[U];
    int cc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const IDX i = min((IDX)(i0 + u * stride), (IDX)(total - 1));
      const IDX v = i / cv;
      vv[u] = (int64_t)v;
      cc[u] = (int)(i - v * cv);
      const int b = (int)(v / (IDX)S);
      const int64_t sidx = (int64_t)v - (int64_t)b * S;
#pragma unroll
      for (int co = 0; co < 16; ++co) g[u][co] = co < Cout ? dy[((int64_t)b * Cout + co) * S + sidx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * stride >= total) break;
      V<T, VEC> o;
#pragma unroll
      for (int k = 0; k < VEC; ++k) o.v[k] = 0.f;
#pragma unroll
      for (int co = 0; co < 16; ++co) {
        if (co < Cout) {
#pragma unroll
          for (int k4 = 0; k4 < VEC / 4; ++k4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(ws + co * Cin + cc[u] * VEC + 4 * k4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o.v[4 * k4 + e] = fmaf(g[u][co], w4[e], o.v[4 * k4 + e]);
          }
        }
      }
      o.store(dx + vv[u] * lddx + cc[u] * VEC);
    }
  }
}

// The same for up to 8 classes and whole 256-voxel tiles: the lane-per-item kernel above reads the six dy planes with 43-byte runs (six
// lanes share a voxel) and re-reads its weights from LDS for every item - 57 us for 106 MB on the 96^3 head.  Here a workgroup stages the
// dy tile with one coalesced 1 KB run per class, a lane keeps the weights of ITS channel vector in registers for all its tiles, and the
// 3 KB of a pass (32 voxels x 96 bytes at Cin 48) leave as one contiguous run.
template <class T, int VEC>
__global__ void __launch_bounds__(256) head_bwd_dx_tile_kernel(const float* __restrict__ dy, T* __restrict__ dx, int64_t lddx, const float* __restrict__ w, int S,
                                                               int Cin, int Cout, int64_t ntiles) {
  __shared__ __attribute__((aligned(16))) float gl[256 * 8];      // [voxel][class, padded to 8]
  const int cv = Cin / VEC, vpp = 256 / cv, tid = threadIdx.x;
  const int vl = tid / cv, cvec = tid - vl * cv;
  const bool worker = vl < vpp;
  float wr[8][VEC];
#pragma unroll
  for (int co = 0; co < 8; ++co)
#pragma unroll
    for (int k = 0; k < VEC; ++k) wr[co][k] = (worker && co < Cout) ? w[co * Cin + cvec * VEC + k] : 0.f;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t v0 = t * 256;
    const int b = (int)(v0 / S);
    const int64_t s0 = v0 - (int64_t)b * S;
    float g8[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) g8[co] = co < Cout ? dy[((int64_t)b * Cout + co) * S + s0 + tid] : 0.f;
    __syncthreads();      // the previous tile's readers are done
    *reinterpret_cast<f32x4*>(gl + tid * 8) = f32x4{g8[0], g8[1], g8[2], g8[3]};
    *reinterpret_cast<f32x4*>(gl + tid * 8 + 4) = f32x4{g8[4], g8[5], g8[6], g8[7]};
    __syncthreads();
    if (worker) {
      for (int vox = vl; vox < 256; vox += vpp) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gl + vox * 8), gb = *reinterpret_cast<const f32x4*>(gl + vox * 8 + 4);
        const float g[8] = {ga[0], ga[1], ga[2], ga[3], gb[0], gb[1], gb[2], gb[3]};
        V<T, VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = 0.f;
#pragma unroll
        for (int co = 0; co < 8; ++co)
#pragma unroll
          for (int k = 0; k < VEC; ++k) o.v[k] = fmaf(g[co], wr[co][k], o.v[k]);
        o.store(dx + (v0 + vox) * lddx + cvec * VEC);
      }
    }
  }
}

// dw[co][ci] += sum_v dy[b][co][s] x[v][ci], dbias[co] += sum_v dy: lane = (row ty, channel vector tx), CO <= 8 per pass
template <class T, int VEC>
__global__ void __launch_bounds__(256) head_bwd_dw_team_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ dy, float* __restrict__ dw,
                                                               float* __restrict__ dbias, int B, int S, int Cin, int Cout, int co0, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int tx_n = Cin / VEC, ty_n = 256 / tx_n;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int64_t nv = (int64_t)B * S;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(nv, r0 + rows_per_block);
  float acc[8][VEC], bsum[8];
#pragma unroll
  for (int co = 0; co < 8; ++co)
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[co][k] = 0.f;
#pragma unroll
  for (int co = 0; co < 8; ++co) bsum[co] = 0.f;
  const int nco = min(8, Cout - co0);
  if (ty < ty_n) {
    // four rows per trip, all their loads issued before the first FMA (one row per trip waited a full memory round trip per row:
    // 96 us for 106 MB)
    constexpr int U = 4;
    int64_t v = r0 + ty;
    for (; v + (int64_t)(U - 1) * ty_n < r1; v += (int64_t)U * ty_n) {
      V<T, VEC> xv[U];
      float g[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t vv = v + (int64_t)u * ty_n;
        xv[u].load(x + vv * ldx + tx * VEC);
        const int b = (int)(vv / S);
        const int64_t sidx = vv - (int64_t)b * S;
#pragma unroll
        for (int co = 0; co < 8; ++co) g[u][co] = co < nco ? dy[((int64_t)b * Cout + co0 + co) * S + sidx] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int co = 0; co < 8; ++co) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[co][k] = fmaf(g[u][co], xv[u].v[k], acc[co][k]);
          bsum[co] += g[u][co];
        }
    }
    for (; v < r1; v += ty_n) {
      V<T, VEC> xv;
      xv.load(x + v * ldx + tx * VEC);
      const int b = (int)(v / S);
      const int64_t sidx = v - (int64_t)b * S;
#pragma unroll
      for (int co = 0; co < 8; ++co) {
        if (co < nco) {
          const float g = dy[((int64_t)b * Cout + co0 + co) * S + sidx];
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[co][k] = fmaf(g, xv.v[k], acc[co][k]);
          bsum[co] += g;
        }
      }
    }
  }
  team_reduce_atomic<8, VEC>(red, acc, tx, ty, tx_n, ty_n, dw + (int64_t)co0 * Cin, Cin, nco, Cin);
  if (dbias) {   // every lane of a row holds the same sums: take lane tx == 0
    __syncthreads();
    if (tx == 0 && ty < ty_n) {
#pragma unroll
      for (int co = 0; co < 8; ++co) red[ty * 8 + co] = bsum[co];
    }
    __syncthreads();
    if ((int)threadIdx.x < nco) {
      float a = 0.f;
      for (int yy = 0; yy < ty_n; ++yy) a += red[yy * 8 + threadIdx.x];
      atomicAdd(dbias + co0 + threadIdx.x, a);
    }
  }
}

// Head weight gradient on the matrix cores (bf16 activations, Cin == 48, Cout <= 16, S % 256 == 0): dw[co][ci] = sum_v dy[co][v] x[v][ci]
// is a 16 x 48 x V GEMM.  Tiles of 256 voxel rows of x are staged in LDS and read transposed (`ds_read_b64_tr_b16`, as the conv
// weight-gradient kernels do); dy is already voxel-contiguous per output channel: a lane converts 8 consecutive fp32 values of its
// channel.  Each wave takes two of a tile's eight 32-voxel k-steps; 3 MFMAs per k-step.  The team kernel (one 16-byte load per lane and
// row, then 6 x 8 multiply-adds) ran at 1.1 TB/s.
__global__ void __launch_bounds__(256) head_bwd_dw_mfma_kernel(const bf16* __restrict__ x, int64_t ldx, const float* __restrict__ dy, float* __restrict__ dw,
                                                               float* __restrict__ dbias, int S, int Cout, int64_t ntiles) {
  constexpr int Cin = 48, ROWB = Cin * 2 + 16;
  __shared__ __attribute__((aligned(16))) char xs[256 * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fq = lane >> 4, qq = fi >> 2, p4 = (fi & 3) * 4;
  f32x4 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool has_co = fi < Cout;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t v0 = tile * 256;
    const int b = (int)(v0 / S);
    const int64_t s0 = v0 - (int64_t)b * S;
    __syncthreads();
    for (int i = tid; i < 256 * 6; i += 256) {
      const int r = i / 6, c = (i - r * 6) * 8;
      *reinterpret_cast<bf16x8*>(xs + r * ROWB + c * 2) = *reinterpret_cast<const bf16x8*>(x + (v0 + r) * ldx + c);
    }
    // this lane's dy values of both its k-steps (issued before the barrier: in flight while the tile is staged)
    f32x4 g[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const float* gp = dy + ((int64_t)b * Cout + (has_co ? fi : 0)) * S + s0 + (wave + 4 * kk) * 32 + fq * 8;
      g[kk][0] = has_co ? *reinterpret_cast<const f32x4*>(gp) : f32x4{0.f, 0.f, 0.f, 0.f};
      g[kk][1] = has_co ? *reinterpret_cast<const f32x4*>(gp + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int vrow = (wave + 4 * kk) * 32 + fq * 8 + qq;
      bf16x8 gf;
#pragma unroll
      for (int e = 0; e < 4; ++e) { gf[e] = (bf16)g[kk][0][e]; gf[4 + e] = (bf16)g[kk][1][e]; }
      bsum += (g[kk][0][0] + g[kk][0][1]) + (g[kk][0][2] + g[kk][0][3]) + (g[kk][1][0] + g[kk][1][1]) + (g[kk][1][2] + g[kk][1][3]);
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        const char* a1 = xs + vrow * ROWB + (mt * 16 + p4) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1 + 4 * ROWB));
        const bf16x8 xf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, xf, acc[mt], 0, 0, 0);      // lane: co = 4fq + r, ci = 16mt + fi
      }
    }
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(xs);          // [4 waves][16 co][48 ci] + [4 waves][64 lanes] bias partials
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 16 + fq * 4 + r) * 48 + mt * 16 + fi] = acc[mt][r];
  float* bred = red + 4 * 16 * 48;
  bred[wave * 64 + lane] = bsum;
  __syncthreads();
  for (int o = tid; o < Cout * 48; o += 256) {
    const int co = o / 48, ci = o - co * 48;
    atomicAdd(dw + o, red[(0 * 16 + co) * 48 + ci] + red[(1 * 16 + co) * 48 + ci] + red[(2 * 16 + co) * 48 + ci] + red[(3 * 16 + co) * 48 + ci]);
  }
  if (dbias && tid < Cout) {
    float a = 0.f;
    for (int w4 = 0; w4 < 4; ++w4)
      for (int q = 0; q < 4; ++q) a += bred[w4 * 64 + q * 16 + tid];
    atomicAdd(dbias + tid, a);
  }
}

// dw[co][ci][a][b][c] += sum over coarse voxels of dy[cv][co] x[b][ci][2d+a][2h+b][2w+c], dbias[co] += sum dy[cv][co]
// lane = (coarse voxel row ty, output-channel vector tx); one input channel per pass (blockIdx.y)
template <class T, int VEC>
__global__ void __launch_bounds__(256) patch_embed_bwd_team_kernel(const float* __restrict__ x, const T* __restrict__ dy, int64_t lddy, float* __restrict__ dw,
                                                                   float* __restrict__ dbias, int B, int Cin, int D, int H, int W, int Cout, int rows_per_block,
                                                                   float* __restrict__ partial) {
  // partial != nullptr: the workgroup's 9 * Cout sums go to partial[ci][block][tap][co] (plain stores) and patch_embed_bwd_reduce_kernel adds
  // them up: 256 workgroups x 432 fp32 atomics onto 14 cache lines cost 46 of this kernel's 53 us
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int tx_n = Cout / VEC, ty_n = 256 / tx_n;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int ci = blockIdx.y;
  const int D2 = D / 2, H2 = H / 2, W2 = W / 2;
  const int64_t nv = (int64_t)B * D2 * H2 * W2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(nv, r0 + rows_per_block);
  float acc[9][VEC];   // 8 taps + the bias row
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[t][k] = 0.f;
  if (ty < ty_n) {
#pragma unroll 4
    for (int64_t v = r0 + ty; v < r1; v += ty_n) {
      V<T, VEC> g;
      g.load(dy + v * lddy + tx * VEC);
      int64_t t = v;
      const int w2 = (int)(t % W2); t /= W2;
      const int h2 = (int)(t % H2); t /= H2;
      const int d2 = (int)(t % D2);
      const int b = (int)(t / D2);
      const float* xp = x + ((((int64_t)b * Cin + ci) * D + 2 * d2) * H + 2 * h2) * W + 2 * w2;
#pragma unroll
      for (int tap = 0; tap < 8; ++tap) {
        const float xv = xp[((int64_t)(tap >> 2) * H + ((tap >> 1) & 1)) * W + (tap & 1)];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[tap][k] = fmaf(xv, g.v[k], acc[tap][k]);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[8][k] += g.v[k];
    }
  }
  // out[tap][co] lives at dw[(co * Cin + ci) * 8 + tap]: reduce into LDS, then scatter
  const int width = 9 * tx_n * VEC;
  __syncthreads();
  if (ty < ty_n) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int k = 0; k < VEC; ++k) red[ty * width + (t * tx_n + tx) * VEC + k] = acc[t][k];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < width; e += 256) {
    float a = 0.f;
    for (int yy = 0; yy < ty_n; ++yy) a += red[yy * width + e];
    const int tap = e / (tx_n * VEC), co = e - tap * (tx_n * VEC);
    if (partial) partial[((int64_t)ci * gridDim.x + blockIdx.x) * width + e] = a;
    else if (tap < 8) atomicAdd(dw + ((int64_t)co * Cin + ci) * 8 + tap, a);
    else if (dbias && ci == 0) atomicAdd(dbias + co, a);
  }
}

// sums partial[ci][nblk][9 * Cout] over the workgroups: a workgroup = 32 consecutive elements x 8 slices of the workgroup list
__global__ void __launch_bounds__(256) patch_embed_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ dbias, int Cin,
                                                                     int Cout, int nblk) {
  __shared__ float red[8][32];
  const int width = 9 * Cout, ci = blockIdx.y;
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  float a = 0.f;
  if (e < width) {
    const float* p = partial + (int64_t)ci * nblk * width + e;
#pragma unroll 8
    for (int b = sl; b < nblk; b += 8) a += p[(int64_t)b * width];
  }
  red[sl][threadIdx.x & 31] = a;
  __syncthreads();
  if (sl == 0 && e < width) {
#pragma unroll
    for (int i = 1; i < 8; ++i) a += red[i][threadIdx.x];
    const int tap = e / Cout, co = e - tap * Cout;
    if (tap < 8) dw[((int64_t)co * Cin + ci) * 8 + tap] += a;        // single writer per element
    else if (dbias && ci == 0) dbias[co] += a;
  }
}

// ----------------------------------------------------------------------------- im2col / col2im (3x3x3 pad 1)
template <class T, int VEC>
__global__ void __launch_bounds__(256) im2col3_kernel(const T* src, int64_t lds, T* dst, int64_t ldd, int B, int D, int H, int W, int C) {
  const int cv = C / VEC;
  const int64_t total = (int64_t)B * D * H * W * 27 * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    int64_t t = i / cv;
    const int tap = (int)(t % 27); t /= 27;
    const int64_t v = t;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D);
    const int b = (int)(t / D);
    const int dd = d + tap / 9 - 1, hh = h + (tap / 3) % 3 - 1, ww = w + tap % 3 - 1;
    V<T, VEC> val;
    if (dd >= 0 && dd < D && hh >= 0 && hh < H && ww >= 0 && ww < W) val.load(src + ((((int64_t)b * D + dd) * H + hh) * W + ww) * lds + c);
    else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) val.v[k] = 0.f;
    }
    val.store(dst + v * ldd + (int64_t)tap * C + c);
  }
}

template <class T, int VEC>
__global__ void __launch_bounds__(256) col2im3_kernel(const T* col, int64_t ldc, T* dst, int64_t ldd, int B, int D, int H, int W, int C) {
  const int cv = C / VEC;
  const int64_t total = (int64_t)B * D * H * W * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    int64_t t = i / cv;
    const int64_t v = t;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D);
    const int b = (int)(t / D);
    V<T, VEC> acc;
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc.v[k] = 0.f;
    for (int tap = 0; tap < 27; ++tap) {
      // col[u][tap] holds src[u + off(tap)]  =>  dst[v] += col[v - off(tap)][tap]
      const int dd = d - (tap / 9 - 1), hh = h - ((tap / 3) % 3 - 1), ww = w - (tap % 3 - 1);
      if (dd >= 0 && dd < D && hh >= 0 && hh < H && ww >= 0 && ww < W) {
        V<T, VEC> val;
        val.load(col + ((((int64_t)b * D + dd) * H + hh) * W + ww) * ldc + (int64_t)tap * C + c);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc.v[k] += val.v[k];
      }
    }
    acc.store(dst + v * ldd + c);
  }
}

// ----------------------------------------------------------------------------- column sums
// grid (row chunks, channel tiles); tx = 16-byte channel vector, ty walks rows; LDS reduce; one atomic per channel.
template <class T, int VEC>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int C, int cv, int tx_n, int ty_n,
                                                     float* __restrict__ out, int rows_per_block) {
  extern __shared__ float red[];
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int c0 = blockIdx.y * tx_n, c = c0 + tx;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.f;
  if (ty < ty_n && c < cv) {
    for (int64_t r = r0 + ty; r < r1; r += ty_n) {
      V<T, VEC> v;
      v.load(x + r * ldx + c * VEC);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s[i] += v.v[i];
    }
  }
  if (ty < ty_n) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[ty * tx_n * VEC + tx * VEC + i] = s[i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < tx_n * VEC; e += 256) {
    float acc = 0.f;
    for (int y = 0; y < ty_n; ++y) acc += red[y * tx_n * VEC + e];
    const int ch = c0 * VEC + e;
    if (ch < C) atomicAdd(out + ch, acc);
  }
}

struct ColsumBatch { miseg_colsum_desc d[MISEG_COLSUM_BATCH]; int n; };

// one launch for a list of accumulate-mode column sums: workgroup -> (descriptor, 512-row chunk, tile of 64 channels);
// a lane reads one 16-byte channel vector per row (vec descriptors) or single elements (any C / alignment)
static constexpr int CSB_ROWS = 512, CSB_CH = 64;

template <class T>
__global__ void __launch_bounds__(256) colsum_batch_kernel(ColsumBatch b) {
  typedef typename Vec16<T>::type VT;
  constexpr int N = Vec16<T>::N;
  __shared__ float red[32][CSB_CH + 1];
  int k = 0;
  while (k + 1 < b.n && b.d[k + 1].block0 <= (int)blockIdx.x) ++k;
  const miseg_colsum_desc d = b.d[k];
  const int ctiles = (d.C + CSB_CH - 1) / CSB_CH;
  const int local = blockIdx.x - d.block0;
  const int c0 = (local % ctiles) * CSB_CH;
  const int64_t r0 = (int64_t)(local / ctiles) * CSB_ROWS, r1 = min(d.rows, r0 + CSB_ROWS);
  const T* x = reinterpret_cast<const T*>(d.x);
  const bool vec = d.C % N == 0 && d.ldx % N == 0 && ((uintptr_t)d.x % 16 == 0);
  for (int i = threadIdx.x; i < 32 * (CSB_CH + 1); i += 256) (&red[0][0])[i] = 0.f;
  __syncthreads();
  if (vec) {
    constexpr int TX = CSB_CH / N, TY = 256 / TX;     // bf16: 8 x 32, fp32: 16 x 16
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = c0 + tx * N;
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.f;
    if (c < d.C) {
      // eight rows in flight per lane (round 4: one dependent load per iteration streamed at 2.4 TB/s; rows beyond the chunk re-read its
      // last row and are not added)
      constexpr int U = 8;
      for (int64_t r = r0 + ty; r < r1; r += U * TY) {
        VT v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t ru = r + (int64_t)u * TY;
          v[u] = *reinterpret_cast<const VT*>(x + (ru < r1 ? ru : r1 - 1) * d.ldx + c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float keep = (r + (int64_t)u * TY) < r1 ? 1.f : 0.f;
#pragma unroll
          for (int e = 0; e < N; ++e) acc[e] = fmaf(keep, to_f32(v[u][e]), acc[e]);
        }
      }
    }
    // TY <= 32: one row of `red` per ty
#pragma unroll
    for (int e = 0; e < N; ++e) red[ty][tx * N + e] = acc[e];
  } else {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 channels x 4 row lanes
    const int c = c0 + tx;
    float acc = 0.f;
    if (c < d.C)
      for (int64_t r = r0 + ty; r < r1; r += 4) acc += to_f32(x[r * d.ldx + c]);
    red[ty][tx] = acc;
  }
  __syncthreads();
  if (threadIdx.x < CSB_CH && c0 + (int)threadIdx.x < d.C) {
    float t = 0.f;
#pragma unroll
    for (int y = 0; y < 32; ++y) t += red[y][threadIdx.x];
    atomicAdd(d.out + c0 + threadIdx.x, t);
  }
}

}  // namespace miseg

using namespace miseg;

#define DT(p, ...)                                                          \
  return dispatch_dtype((p)->dtype, [&](auto* tag) -> int {                 \
    typedef typename std::remove_pointer<decltype(tag)>::type T;            \
    __VA_ARGS__;                                                            \
    return MISEG_OK;                                                        \
  })

namespace miseg {
template <class T, int VEC>
__global__ void __launch_bounds__(256) affine2_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                      const float* __restrict__ coef, int B, int S, int cv, int C) {
  const int64_t total = (int64_t)B * S * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    const int b = (int)(r / S);
    V<T, VEC> av, xv;
    av.load(a + r * lda + c);
    xv.load(x + r * ldx + c);
    const float* k = coef + ((int64_t)b * C + c) * 3;
#pragma unroll
    for (int e = 0; e < VEC; ++e) av.v[e] = fmaf(k[3 * e], av.v[e], fmaf(k[3 * e + 1], xv.v[e], k[3 * e + 2]));
    av.store(y + r * ldy + c);
  }
}
}  // namespace miseg

extern "C" int miseg_affine2(const miseg_affine2_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_affine2_params), MISEG_E_BADARG, "affine2: struct_size");
  MISEG_REQUIRE(p->a && p->x && p->y && p->coef && p->B > 0 && p->S > 0 && p->C > 0, MISEG_E_BADARG, "affine2: bad args");
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->lda % N == 0 && p->ldx % N == 0 && p->ldy % N == 0 && al16(p->a) && al16(p->x) && al16(p->y);
    const int cv = vec ? p->C / N : p->C;
    int64_t n = (int64_t)p->B * p->S * cv;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (vec) miseg::affine2_kernel<T, N><<<grid, 256, 0, s>>>((const T*)p->a, p->lda, (const T*)p->x, p->ldx, (T*)p->y, p->ldy, p->coef, p->B, p->S, cv, p->C);
    else miseg::affine2_kernel<T, 1><<<grid, 256, 0, s>>>((const T*)p->a, p->lda, (const T*)p->x, p->ldx, (T*)p->y, p->ldy, p->coef, p->B, p->S, cv, p->C);
    MISEG_LAUNCH_CHECK("affine2");
  });
}

extern "C" int miseg_add(const miseg_add_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->a && p->b && p->y && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "add: bad args");
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->lda % N == 0 && p->ldb % N == 0 && p->ldy % N == 0 && al16(p->a) && al16(p->b) && al16(p->y);
    if (vec) add_kernel<T, N><<<ew_grid(p->rows * (p->C / N)), 256, 0, s>>>((const T*)p->a, p->lda, (const T*)p->b, p->ldb, (T*)p->y, p->ldy, p->rows, p->C / N);
    else add_kernel<T, 1><<<ew_grid(p->rows * p->C), 256, 0, s>>>((const T*)p->a, p->lda, (const T*)p->b, p->ldb, (T*)p->y, p->ldy, p->rows, p->C);
    MISEG_LAUNCH_CHECK("add");
  });
}

extern "C" int miseg_copy2d(const miseg_copy2d_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->src && p->dst && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "copy2d: bad args");
  const int g = ew_grid(p->rows * p->C);
  if (p->sdtype == MISEG_F32 && p->ddtype == MISEG_F32) copy2d_kernel<float, float><<<g, 256, 0, s>>>((const float*)p->src, p->lds, (float*)p->dst, p->ldd, p->rows, p->C);
  else if (p->sdtype == MISEG_F32 && p->ddtype == MISEG_BF16) copy2d_kernel<float, bf16><<<g, 256, 0, s>>>((const float*)p->src, p->lds, (bf16*)p->dst, p->ldd, p->rows, p->C);
  else if (p->sdtype == MISEG_BF16 && p->ddtype == MISEG_F32) copy2d_kernel<bf16, float><<<g, 256, 0, s>>>((const bf16*)p->src, p->lds, (float*)p->dst, p->ldd, p->rows, p->C);
  else if (p->sdtype == MISEG_BF16 && p->ddtype == MISEG_BF16) copy2d_kernel<bf16, bf16><<<g, 256, 0, s>>>((const bf16*)p->src, p->lds, (bf16*)p->dst, p->ldd, p->rows, p->C);
  else return set_error(MISEG_E_BADARG, "copy2d: dtypes %d -> %d", p->sdtype, p->ddtype);
  MISEG_LAUNCH_CHECK("copy2d");
  return MISEG_OK;
}

extern "C" int miseg_cast_matrix(const miseg_cast_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->src && p->dst && p->R > 0 && p->C > 0, MISEG_E_BADARG, "cast_matrix: bad args");
  DT(p, {
    if (p->transpose) cast_transpose_kernel<T><<<dim3(cdiv(p->C, 32), cdiv(p->R, 32)), 256, 0, s>>>(p->src, (T*)p->dst, p->R, p->C);
    else cast_kernel<T><<<ew_grid((int64_t)p->R * p->C), 256, 0, s>>>(p->src, (T*)p->dst, (int64_t)p->R * p->C);
    MISEG_LAUNCH_CHECK("cast_matrix");
  });
}

extern "C" int miseg_gelu_bwd(const miseg_gelu_bwd_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->dy && p->x && p->dx && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "gelu_bwd: bad args");
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->lddy % N == 0 && p->ldx % N == 0 && p->lddx % N == 0 && al16(p->dy) && al16(p->x) && al16(p->dx);
    if (vec) gelu_bwd_kernel<T, N><<<ew_grid(p->rows * (p->C / N)), 256, 0, s>>>((const T*)p->dy, p->lddy, (const T*)p->x, p->ldx, (T*)p->dx, p->lddx, p->rows, p->C / N);
    else gelu_bwd_kernel<T, 1><<<ew_grid(p->rows * p->C), 256, 0, s>>>((const T*)p->dy, p->lddy, (const T*)p->x, p->ldx, (T*)p->dx, p->lddx, p->rows, p->C);
    MISEG_LAUNCH_CHECK("gelu_bwd");
  });
}

extern "C" int miseg_gelu_fwd(const miseg_gelu_fwd_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->y && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "gelu_fwd: bad args");
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->ldx % N == 0 && p->ldy % N == 0 && al16(p->x) && al16(p->y);
    if (vec) gelu_fwd_kernel<T, N><<<ew_grid(p->rows * (p->C / N)), 256, 0, s>>>((const T*)p->x, p->ldx, (T*)p->y, p->ldy, p->rows, p->C / N);
    else gelu_fwd_kernel<T, 1><<<ew_grid(p->rows * p->C), 256, 0, s>>>((const T*)p->x, p->ldx, (T*)p->y, p->ldy, p->rows, p->C);
    MISEG_LAUNCH_CHECK("gelu_fwd");
  });
}

static int s2c_common(const miseg_s2c_params* p, hipStream_t s, bool gather) {
  MISEG_REQUIRE(p && p->src && p->dst && p->B > 0 && p->D > 0 && p->H > 0 && p->W > 0 && p->C > 0, MISEG_E_BADARG, "space/channel: bad args");
  Off8 off;
  for (int i = 0; i < 24; ++i) {
    off.o[i] = p->offsets[i];
    MISEG_REQUIRE(off.o[i] == 0 || off.o[i] == 1, MISEG_E_BADARG, "space/channel: offsets must be 0/1");
  }
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->lds % N == 0 && p->ldd % N == 0 && al16(p->src) && al16(p->dst);
    const int D2 = (p->D + 1) / 2, H2 = (p->H + 1) / 2, W2 = (p->W + 1) / 2;
    if (gather) {
      const int64_t tot = (int64_t)p->B * D2 * H2 * W2 * 8;
      if (vec && tot * (p->C / N) < (1LL << 30)) s2c_kernel<T, N, unsigned><<<ew_grid(tot * (p->C / N)), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C, off);
      else if (vec) s2c_kernel<T, N><<<ew_grid(tot * (p->C / N)), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C, off);
      else s2c_kernel<T, 1><<<ew_grid(tot * p->C), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C, off);
    } else {
      const int64_t tot = (int64_t)p->B * p->D * p->H * p->W;
      if (vec && tot * (p->C / N) < (1LL << 30)) c2s_kernel<T, N, unsigned><<<ew_grid(tot * (p->C / N)), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C, off);
      else if (vec) c2s_kernel<T, N><<<ew_grid(tot * (p->C / N)), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C, off);
      else c2s_kernel<T, 1><<<ew_grid(tot * p->C), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C, off);
    }
    MISEG_LAUNCH_CHECK("space/channel");
  });
}
extern "C" int miseg_space_to_channel(const miseg_s2c_params* p, miseg_stream_t s) { return s2c_common(p, (hipStream_t)s, true); }
extern "C" int miseg_channel_to_space(const miseg_s2c_params* p, miseg_stream_t s) { return s2c_common(p, (hipStream_t)s, false); }

extern "C" int miseg_patch_embed_fwd(const miseg_patch_embed_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->y && p->w, MISEG_E_BADARG, "patch_embed_fwd: null pointer");
  MISEG_REQUIRE(p->D % 2 == 0 && p->H % 2 == 0 && p->W % 2 == 0, MISEG_E_BADARG, "patch_embed_fwd: grid %dx%dx%d must be even (pad on host)", p->D, p->H, p->W);
  MISEG_REQUIRE(p->Cin * 8 * p->Cout <= 12000, MISEG_E_UNSUPPORTED, "patch_embed_fwd: Cin*8*Cout too large");
  DT(p, {
    const int64_t tot = (int64_t)p->B * (p->D / 2) * (p->H / 2) * (p->W / 2) * ((p->Cout + 7) / 8);
    size_t sh = ((size_t)p->Cin * 8 * p->Cout + p->Cout) * sizeof(float);
    // (grid sweep, round 5: 2592 = 1296 workgroups 15.2 us, 864: 17.7, 648: 20.7 - the kernel is its instruction count, not its launch shape)
    if (tot + 256 * 8192 < (1LL << 31) && (int64_t)p->B * p->Cin * p->D * p->H * p->W < (1LL << 31))
      patch_embed_fwd_kernel<T, unsigned><<<ew_grid(tot), 256, sh, s>>>(p->x, (T*)p->y, p->ldy, p->w, p->bias, p->B, p->Cin, p->D, p->H, p->W, p->Cout);
    else
      patch_embed_fwd_kernel<T><<<ew_grid(tot), 256, sh, s>>>(p->x, (T*)p->y, p->ldy, p->w, p->bias, p->B, p->Cin, p->D, p->H, p->W, p->Cout);
    MISEG_LAUNCH_CHECK("patch_embed_fwd");
  });
}

extern "C" size_t miseg_patch_embed_bwd_workspace_bytes(const miseg_patch_embed_bwd_params* p) {
  if (!p) return 0;
  return (size_t)256 * p->Cin * 9 * p->Cout * sizeof(float);       // at most 256 workgroups per input channel
}

extern "C" int miseg_patch_embed_bwd(const miseg_patch_embed_bwd_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->dy && p->dw, MISEG_E_BADARG, "patch_embed_bwd: null pointer");
  MISEG_REQUIRE(p->Cout * (p->Cin * 8 + 1) <= 2048, MISEG_E_UNSUPPORTED, "patch_embed_bwd: Cout*(8Cin+1) > 2048");
  DT(p, {
    const int64_t nv = (int64_t)p->B * (p->D / 2) * (p->H / 2) * (p->W / 2);
    constexpr int N = Vec16<T>::N;
    const int tx_n = p->Cout / N;
    if (p->Cout % N == 0 && p->lddy % N == 0 && al16(p->dy) && tx_n >= 1 && tx_n <= 64) {
      const int ty_n = 256 / tx_n;
      int rpb = (int)((nv + 255) / 256);
      if (rpb < 4 * ty_n) rpb = 4 * ty_n;
      const size_t sh = (size_t)ty_n * 9 * tx_n * N * sizeof(float);
      hipFuncSetAttribute((const void*)patch_embed_bwd_team_kernel<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      const int nblk = cdiv(nv, rpb);
      float* part = (p->workspace && nblk >= 16) ? (float*)p->workspace : nullptr;
      patch_embed_bwd_team_kernel<T, N><<<dim3(nblk, p->Cin), 256, sh, s>>>(p->x, (const T*)p->dy, p->lddy, p->dw, p->dbias, p->B, p->Cin, p->D, p->H, p->W,
                                                                            p->Cout, rpb, part);
      if (part) patch_embed_bwd_reduce_kernel<<<dim3(cdiv(9 * p->Cout, 32), p->Cin), 256, 0, s>>>(part, p->dw, p->dbias, p->Cin, p->Cout, nblk);
    } else {
      const int vpb = 1024;
      size_t sh = ((size_t)64 * (p->Cin * 8 + 1) + 64 * p->Cout) * sizeof(float);
      patch_embed_bwd_kernel<T><<<cdiv(nv, vpb), 256, sh, s>>>(p->x, (const T*)p->dy, p->lddy, p->dw, p->dbias, p->B, p->Cin, p->D, p->H, p->W, p->Cout, vpb);
    }
    MISEG_LAUNCH_CHECK("patch_embed_bwd");
  });
}

extern "C" int miseg_conv3_thin_fwd(const miseg_conv3_thin_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->y && p->w, MISEG_E_BADARG, "conv3_thin_fwd: null pointer");
  MISEG_REQUIRE(p->Cin >= 1 && p->Cin <= 4 && p->Cin * 27 * p->Cout <= 15000, MISEG_E_UNSUPPORTED, "conv3_thin_fwd: Cin %d Cout %d", p->Cin, p->Cout);
  DT(p, {
    const int64_t nv = (int64_t)p->B * p->D * p->H * p->W;
    if constexpr (std::is_same<T, bf16>::value) {
      if (p->Cin == 1 && p->Cout == 48 && p->ldy % 4 == 0 && ((uintptr_t)p->y & 7) == 0) {           // matrix-core form (see conv3_stem_fwd_mfma_kernel)
        const int nbricks = p->B * cdiv(p->D, miseg::SB_D) * cdiv(p->H, miseg::SB_H) * cdiv(p->W, miseg::SB_W);
        conv3_stem_fwd_mfma_kernel<<<nbricks < 2048 ? nbricks : 2048, 256, 0, s>>>(p->x, (bf16*)p->y, p->ldy, p->w, p->B, p->D, p->H, p->W, nbricks);
        MISEG_LAUNCH_CHECK("conv3_stem_fwd_mfma");
        return MISEG_OK;
      }
    }
    if (p->Cin == 1 && p->Cout % 8 == 0 && p->Cout <= 64 && p->ldy % 8 == 0 && al16(p->y)) {       // brick kernel (see conv3_stem_fwd_kernel)
      const int nbricks = p->B * cdiv(p->D, miseg::SB_D) * cdiv(p->H, miseg::SB_H) * cdiv(p->W, miseg::SB_W);
      conv3_stem_fwd_kernel<T><<<nbricks < 4096 ? nbricks : 4096, 256, 0, s>>>(p->x, (T*)p->y, p->ldy, p->w, p->B, p->D, p->H, p->W, p->Cout, nbricks);
      MISEG_LAUNCH_CHECK("conv3_stem_fwd");
      return MISEG_OK;
    }
    size_t sh = (size_t)p->Cin * 27 * p->Cout * sizeof(float);
    conv3_thin_fwd_kernel<T><<<ew_grid(nv), 256, sh, s>>>(p->x, (T*)p->y, p->ldy, p->w, p->B, p->Cin, p->D, p->H, p->W, p->Cout);
    MISEG_LAUNCH_CHECK("conv3_thin_fwd");
  });
}

extern "C" size_t miseg_conv3_thin_wgrad_workspace_bytes(const miseg_conv3_thin_wgrad_params* p) {
  return (p && p->dtype == MISEG_BF16 && p->Cin == 1 && p->Cout == 48) ? (size_t)512 * 48 * 27 * sizeof(float) : 0;      // the matrix-core form only
}

extern "C" int miseg_conv3_thin_wgrad(const miseg_conv3_thin_wgrad_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->dy && p->dw, MISEG_E_BADARG, "conv3_thin_wgrad: null pointer");
  MISEG_REQUIRE(p->Cin >= 1 && p->Cin <= 4 && p->Cout <= 64, MISEG_E_UNSUPPORTED, "conv3_thin_wgrad: Cin %d Cout %d", p->Cin, p->Cout);
  DT(p, {
    constexpr int VN = Vec16<T>::N;
    if constexpr (std::is_same<T, bf16>::value) {
      if (p->Cin == 1 && p->Cout == 48 && p->lddy % 8 == 0 && al16(p->dy)) {      // matrix-core form (see conv3_stem_wgrad_mfma_kernel)
        const int nbricks = p->B * cdiv(p->D, miseg::SB_D) * cdiv(p->H, miseg::SB_H) * cdiv(p->W, miseg::SB_W);
        const int nblk = nbricks < 512 ? nbricks : 512;
        float* part = (p->workspace && nblk >= 16) ? (float*)p->workspace : nullptr;
        conv3_stem_wgrad_mfma_kernel<<<nblk, 256, 0, s>>>(p->x, (const bf16*)p->dy, p->lddy, p->dw, p->B, p->D, p->H, p->W, nbricks, part);
        if (part) partial_sum_add_kernel<<<cdiv(48 * 27, 32), 256, 0, s>>>(part, p->dw, 48 * 27, nblk);
        MISEG_LAUNCH_CHECK("conv3_stem_wgrad_mfma");
        return MISEG_OK;
      }
    }
    if (p->Cin == 1 && p->Cout % 8 == 0 && p->Cout % VN == 0 && p->Cout / 8 * 36 <= 256 && p->lddy % VN == 0 && al16(p->dy)) {   // see conv3_stem_wgrad_kernel
      const int nbricks = p->B * cdiv(p->D, miseg::SB_D) * cdiv(p->H, miseg::SB_H) * cdiv(p->W, miseg::SB_W);
      size_t sh = 2608 + (size_t)256 * p->Cout * sizeof(T);
      if (sh < 256 * 24 * 4) sh = 256 * 24 * 4;
      (void)hipFuncSetAttribute((const void*)conv3_stem_wgrad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      conv3_stem_wgrad_kernel<T><<<nbricks < 512 ? nbricks : 512, 256, sh, s>>>(p->x, (const T*)p->dy, p->lddy, p->dw, p->B, p->D, p->H, p->W, p->Cout, nbricks);
      MISEG_LAUNCH_CHECK("conv3_stem_wgrad");
      return MISEG_OK;
    }
    const int nb = p->B * cdiv(p->D, 8) * cdiv(p->H, 8) * cdiv(p->W, 8);
    size_t sh = ((size_t)1000 + 512 * p->Cout) * sizeof(float);
    hipFuncSetAttribute((const void*)conv3_thin_wgrad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    conv3_thin_wgrad_kernel<T><<<nb, 256, sh, s>>>(p->x, (const T*)p->dy, p->lddy, p->dw, p->B, p->Cin, p->D, p->H, p->W, p->Cout);
    MISEG_LAUNCH_CHECK("conv3_thin_wgrad");
  });
}

// workgroups of the head's persistent tile loops (256-voxel tiles; 3456 of them at 96^3).  Round 5, scripts/micro/head_bench.py on one box, forward /
// backward (dx + dw): 2048 / 512 workgroups (until then; 640 of the 2048 idle through their second round) 35.0 / 66.4 us, balanced 1728 / 494:
// 36.5 / 65.8, **1152 / 494: 32.6 / 62.8**, 864: 37.0 / 63.1, 576: 47.6 / 63.4, 3456: 33.8 / 82; weight gradient at 256 / 864: + 10 / + 2.5 us
static constexpr int HEAD_GRID_CAP = 1152, HEAD_DW_GRID_CAP = 512;

extern "C" int miseg_head_fwd(const miseg_head_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->y && p->w, MISEG_E_BADARG, "head_fwd: null pointer");
  MISEG_REQUIRE(p->Cout <= 16 && p->Cin * p->Cout <= 8192, MISEG_E_UNSUPPORTED, "head_fwd: Cout %d (max 16)", p->Cout);
  DT(p, {
    size_t sh = ((size_t)p->Cin * p->Cout + p->Cout) * sizeof(float);
    constexpr int N = Vec16<T>::N;
    if (p->Cin % N == 0 && p->ldx % N == 0 && al16(p->x)) {
      const int64_t nv = (int64_t)p->B * p->S;
      const size_t sh2 = (size_t)((p->Cout * p->Cin + p->Cout + 3) & ~3) * 4 + (size_t)256 * (p->Cin * sizeof(T) + 16);
      const int grid = balanced_grid((nv + 255) / 256, HEAD_GRID_CAP);
      hipFuncSetAttribute((const void*)head_fwd_tile_kernel<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2);
      head_fwd_tile_kernel<T, N><<<grid, 256, sh2, s>>>((const T*)p->x, p->ldx, p->y, p->w, p->bias, p->B, p->S, p->Cin, p->Cout);
    } else
      head_fwd_kernel<T, 1><<<ew_grid((int64_t)p->B * p->S), 256, sh, s>>>((const T*)p->x, p->ldx, p->y, p->w, p->bias, p->B, p->S, p->Cin, p->Cout);
    MISEG_LAUNCH_CHECK("head_fwd");
  });
}

extern "C" int miseg_head_bwd(const miseg_head_bwd_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->dy && p->w, MISEG_E_BADARG, "head_bwd: null pointer");
  MISEG_REQUIRE(p->Cout <= 16 && p->Cout * (p->Cin + 1) <= 1024, MISEG_E_UNSUPPORTED, "head_bwd: Cout %d Cin %d", p->Cout, p->Cin);
  DT(p, {
    const int64_t nv = (int64_t)p->B * p->S;
    constexpr int N = Vec16<T>::N;
    if (p->dx) {
      size_t sh = (size_t)p->Cin * p->Cout * sizeof(float);
      if (p->Cin % N == 0 && p->lddx % N == 0 && al16(p->dx) && p->Cout <= 8 && p->S % 256 == 0 && p->Cin / N <= 32 && N == 8) {
        const int64_t ntiles = nv / 256;
        head_bwd_dx_tile_kernel<T, N><<<balanced_grid(ntiles, HEAD_GRID_CAP), 256, 0, s>>>(p->dy, (T*)p->dx, p->lddx, p->w, p->S, p->Cin, p->Cout, ntiles);
      } else if (p->Cin % N == 0 && p->lddx % N == 0 && al16(p->dx))
        if (nv * (p->Cin / N) < (1LL << 29))      // 32-bit item arithmetic (see s2c_kernel)
          head_bwd_dx_team_kernel<T, N, unsigned><<<ew_grid(nv * (p->Cin / N)), 256, sh, s>>>(p->dy, (T*)p->dx, p->lddx, p->w, p->B, p->S, p->Cin, p->Cout);
        else
          head_bwd_dx_team_kernel<T, N><<<ew_grid(nv * (p->Cin / N)), 256, sh, s>>>(p->dy, (T*)p->dx, p->lddx, p->w, p->B, p->S, p->Cin, p->Cout);
      else
        head_bwd_dx_kernel<T, 1><<<ew_grid(nv), 256, sh, s>>>(p->dy, (T*)p->dx, p->lddx, p->w, p->B, p->S, p->Cin, p->Cout);
    }
    bool dw_done = false;
    if constexpr (std::is_same<T, bf16>::value) {
      if (p->dw && p->Cin == 48 && p->Cout <= 16 && p->S % 256 == 0 && p->ldx % 8 == 0 && al16(p->x) && al16(p->dy)) {
        const int64_t ntiles = nv / 256;
        head_bwd_dw_mfma_kernel<<<balanced_grid(ntiles, HEAD_DW_GRID_CAP), 256, 0, s>>>((const bf16*)p->x, p->ldx, p->dy, p->dw, p->dbias, p->S, p->Cout, ntiles);
        dw_done = true;
      }
    }
    if (p->dw && !dw_done) {
      const int tx_n = p->Cin / N;
      if (p->Cin % N == 0 && p->ldx % N == 0 && al16(p->x) && tx_n >= 1 && tx_n <= 64) {
        const int ty_n = 256 / tx_n;
        int rpb = (int)((nv + 2047) / 2048);
        if (rpb < 4 * ty_n) rpb = 4 * ty_n;
        const size_t sh = (size_t)ty_n * 8 * tx_n * N * sizeof(float);
        hipFuncSetAttribute((const void*)head_bwd_dw_team_kernel<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        for (int co0 = 0; co0 < p->Cout; co0 += 8)
          head_bwd_dw_team_kernel<T, N><<<cdiv(nv, rpb), 256, sh, s>>>((const T*)p->x, p->ldx, p->dy, p->dw, p->dbias, p->B, p->S, p->Cin, p->Cout, co0, rpb);
      } else {
        const int vpb = 2048;
        size_t sh = ((size_t)64 * (p->Cin + 1) + (size_t)p->Cout * 64) * sizeof(float);
        head_bwd_dw_kernel<T><<<cdiv(nv, vpb), 256, sh, s>>>((const T*)p->x, p->ldx, p->dy, p->dw, p->dbias, p->B, p->S, p->Cin, p->Cout, vpb);
      }
    }
    MISEG_LAUNCH_CHECK("head_bwd");
  });
}

static int im2col_common(const miseg_im2col3_params* p, hipStream_t s, bool fwd) {
  MISEG_REQUIRE(p && p->src && p->dst && p->B > 0 && p->C > 0, MISEG_E_BADARG, "im2col3: bad args");
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->lds % N == 0 && p->ldd % N == 0 && al16(p->src) && al16(p->dst);
    const int64_t nv = (int64_t)p->B * p->D * p->H * p->W;
    if (fwd) {
      if (vec) im2col3_kernel<T, N><<<ew_grid(nv * 27 * (p->C / N)), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C);
      else im2col3_kernel<T, 1><<<ew_grid(nv * 27 * p->C), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C);
    } else {
      if (vec) col2im3_kernel<T, N><<<ew_grid(nv * (p->C / N)), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C);
      else col2im3_kernel<T, 1><<<ew_grid(nv * p->C), 256, 0, s>>>((const T*)p->src, p->lds, (T*)p->dst, p->ldd, p->B, p->D, p->H, p->W, p->C);
    }
    MISEG_LAUNCH_CHECK("im2col3");
  });
}
extern "C" int miseg_im2col3(const miseg_im2col3_params* p, miseg_stream_t s) { return im2col_common(p, (hipStream_t)s, true); }
extern "C" int miseg_col2im3(const miseg_im2col3_params* p, miseg_stream_t s) { return im2col_common(p, (hipStream_t)s, false); }

extern "C" int miseg_colsum(const miseg_colsum_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->out && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "colsum: bad args");
  if (!p->accumulate) {
    hipError_t e = miseg::fill_words_async(p->out, 0, (size_t)p->C, s);
    MISEG_REQUIRE(e == hipSuccess, MISEG_E_LAUNCH, "colsum: memset failed");
  }
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->ldx % N == 0 && al16(p->x);
    const int cv = vec ? p->C / N : p->C;
    const int tx = cv < 32 ? cv : 32, ty = 256 / tx;
    const int rpb = p->rows >= 65536 ? 512 : 128;
    dim3 grid(cdiv(p->rows, rpb), cdiv(cv, tx));
    const size_t sh = (size_t)ty * tx * (vec ? N : 1) * sizeof(float);
    if (vec) colsum_kernel<T, N><<<grid, 256, sh, s>>>((const T*)p->x, p->ldx, p->rows, p->C, cv, tx, ty, p->out, rpb);
    else colsum_kernel<T, 1><<<grid, 256, sh, s>>>((const T*)p->x, p->ldx, p->rows, p->C, cv, tx, ty, p->out, rpb);
    MISEG_LAUNCH_CHECK("colsum");
  });
}

namespace miseg {
template <class T>
__global__ void __launch_bounds__(256) ncdhw_to_rows_kernel(const float* __restrict__ x, T* __restrict__ y, int Cin, int64_t S, int64_t total) {
  typedef typename Vec16<T>::type VT;
  constexpr int N = Vec16<T>::N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / S, v = i - b * S;
    VT o;
#pragma unroll
    for (int c = 0; c < N; ++c) o[c] = from_f32<T>(c < Cin ? x[(b * Cin + c) * S + v] : 0.f);
    *reinterpret_cast<VT*>(y + i * N) = o;
  }
}
}  // namespace miseg

extern "C" int miseg_ncdhw_to_rows(const float* x, void* y, int B, int Cin, int64_t S, int CP, int dtype, miseg_stream_t s_) {
  MISEG_REQUIRE(x && y && B > 0 && S > 0 && Cin > 0, MISEG_E_BADARG, "ncdhw_to_rows: bad arguments");
  return dispatch_dtype(dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    MISEG_REQUIRE(CP == Vec16<T>::N && Cin <= CP, MISEG_E_UNSUPPORTED, "ncdhw_to_rows: CP must be %d and Cin <= CP", Vec16<T>::N);
    MISEG_REQUIRE((uintptr_t)y % 16 == 0, MISEG_E_BADARG, "ncdhw_to_rows: y must be 16-byte aligned");
    const int64_t total = (int64_t)B * S;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    miseg::ncdhw_to_rows_kernel<T><<<grid, 256, 0, (hipStream_t)s_>>>(x, (T*)y, Cin, S, total);
    MISEG_LAUNCH_CHECK("ncdhw_to_rows");
    return MISEG_OK;
  });
}

namespace miseg {
template <class T>
__global__ void __launch_bounds__(256) param_cast_batch_kernel(const miseg_cast_desc* __restrict__ descs, int ndesc, int total_tiles,
                                                               const int64_t* __restrict__ params_version, int64_t* __restrict__ state) {
  __shared__ float tile[32][33];
  constexpr int NS = 512;
  __shared__ int s_tile0[NS];      // the descriptors' first tiles: the per-tile search reads LDS, not a chain of dependent global loads (round 4)
  // versioned refresh (miseg_hip.h): nothing to do when the copies were made from the current parameters
  const int64_t pv = params_version ? *params_version : 0;
  if (params_version && state[0] == pv) return;
  const bool in_lds = ndesc <= NS;
  if (in_lds) {
    for (int i = threadIdx.x; i < ndesc; i += 256) s_tile0[i] = descs[i].tile0;
    __syncthreads();
  }
  for (int tl = blockIdx.x; tl < total_tiles; tl += gridDim.x) {
    // binary search: last descriptor with tile0 <= tl
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((in_lds ? s_tile0[mid] : descs[mid].tile0) <= tl) lo = mid; else hi = mid - 1;
    }
    const miseg_cast_desc d = descs[lo];
    const int t = tl - d.tile0, tc = (d.C + 31) / 32;
    const int by = (t / tc) * 32, bx = (t % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    T* dst = (T*)d.dst;
    // whole 32 x 32 tiles of 16-byte aligned rows (round 4): one float4 load per thread; the plain cast stores 4 converted values per thread
    // straight from registers, the transposed one 4 consecutive rows of a column from the LDS tile - element-wise 4-byte loads and 2-byte
    // stores streamed at 1.7 TB/s (120 us per live refresh of C-Swin-UNETR's 25 M non-conv weights)
    const bool whole = by + 32 <= d.R && bx + 32 <= d.C && (d.C & 3) == 0;
    if (whole) {
      const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(d.src + (int64_t)(by + r) * d.C + bx + c4);
      if (!d.transpose && d.inner == 1) {
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
        T* q = dst + (int64_t)(by + r) * d.C + bx + c4;
        if (sizeof(T) == 2 && (reinterpret_cast<uintptr_t>(q) & 7) == 0) *reinterpret_cast<uint2*>(q) = *reinterpret_cast<const uint2*>(o);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = o[e];
        }
        continue;
      }
      __syncthreads();                                       // the tile buffer of the previous iteration has been read
      tile[r][c4] = v[0]; tile[r][c4 + 1] = v[1]; tile[r][c4 + 2] = v[2]; tile[r][c4 + 3] = v[3];
      __syncthreads();
      if (d.transpose && (d.R & 3) == 0) {
        const int j = threadIdx.x >> 3, r4 = (threadIdx.x & 7) * 4, c = bx + j;      // column c, rows by + r4 .. + 3
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(tile[r4 + e][j]);
        T* q = dst + (int64_t)((c % d.inner) * d.outer + c / d.inner) * d.R + by + r4;
        if (sizeof(T) == 2 && (reinterpret_cast<uintptr_t>(q) & 7) == 0) *reinterpret_cast<uint2*>(q) = *reinterpret_cast<const uint2*>(o);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = o[e];
        }
        continue;
      }
    } else {
      __syncthreads();                                       // the tile buffer of the previous iteration has been read
      for (int j = ty; j < 32; j += 8)
        if (by + j < d.R && bx + tx < d.C) tile[j][tx] = d.src[(int64_t)(by + j) * d.C + bx + tx];
      __syncthreads();
    }
    if (d.transpose) {
      for (int j = ty; j < 32; j += 8) {
        const int c = bx + j;
        if (c < d.C && by + tx < d.R) dst[(int64_t)((c % d.inner) * d.outer + c / d.inner) * d.R + by + tx] = from_f32<T>(tile[tx][j]);
      }
    } else {
      for (int j = ty; j < 32; j += 8) {
        const int c = bx + tx;
        if (by + j < d.R && c < d.C) dst[(int64_t)(by + j) * d.C + (c % d.inner) * d.outer + c / d.inner] = from_f32<T>(tile[j][tx]);
      }
    }
  }
  refresh_done(params_version, state, pv);
}
}  // namespace miseg

extern "C" int miseg_param_cast_batch(const miseg_cast_desc* descs, int ndesc, int total_tiles, int dtype, const int64_t* params_version, int64_t* state,
                                      miseg_stream_t s_) {
  MISEG_REQUIRE(descs && ndesc > 0 && total_tiles > 0, MISEG_E_BADARG, "param_cast_batch: bad arguments");
  MISEG_REQUIRE((params_version == nullptr) == (state == nullptr), MISEG_E_BADARG, "param_cast_batch: params_version and state go together");
  return dispatch_dtype(dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    static const int cap = miseg::refresh_max_wg("MISEG_CAST_WG", miseg::REFRESH_CAST_WG);
    miseg::param_cast_batch_kernel<T><<<total_tiles < cap ? total_tiles : cap, 256, 0, (hipStream_t)s_>>>(descs, ndesc, total_tiles, params_version, state);
    MISEG_LAUNCH_CHECK("param_cast_batch");
    return MISEG_OK;
  });
}

namespace miseg {
// dir 0: y[coarse] = x[2 * coarse]; dir 1: y[fine] = (all coordinates even) ? x[fine / 2] : 0.   One thread per output element vector.
template <class T, int VEC>
__global__ void __launch_bounds__(256) resample2_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int B, int D, int H, int W, int cv,
                                                        int dir) {
  const int D2 = (D + 1) / 2, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const int od = dir ? D : D2, oh = dir ? H : H2, ow = dir ? W : W2;
  const int64_t total = (int64_t)B * od * oh * ow * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv) * VEC;
    int64_t t = i / cv;
    const int w = (int)(t % ow); t /= ow;
    const int h = (int)(t % oh); t /= oh;
    const int d = (int)(t % od);
    const int b = (int)(t / od);
    const int64_t orow = ((((int64_t)b * od + d) * oh + h) * ow + w);
    V<T, VEC> v;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v.v[k] = 0.f;
    if (!dir) v.load(x + ((((int64_t)b * D + 2 * d) * H + 2 * h) * W + 2 * w) * ldx + c);
    else if (!((d | h | w) & 1)) v.load(x + ((((int64_t)b * D2 + d / 2) * H2 + h / 2) * W2 + w / 2) * ldx + c);
    v.store(y + orow * ldy + c);
  }
}

template <class T, int VEC>
__global__ void __launch_bounds__(256) rowbias_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ bias, T* __restrict__ y, int64_t ldy, int64_t rows,
                                                      int cv) {
  const int64_t total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    V<T, VEC> v;
    v.load(x + r * ldx + c);
#pragma unroll
    for (int k = 0; k < VEC; ++k) v.v[k] += bias[c + k];
    v.store(y + r * ldy + c);
  }
}

template <class T, int VEC>
__global__ void __launch_bounds__(256) prelu_fwd_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ slope, T* __restrict__ y, int64_t ldy,
                                                        int64_t rows, int cv) {
  const float a = slope[0];
  const int64_t total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    V<T, VEC> v;
    v.load(x + r * ldx + c);
#pragma unroll
    for (int k = 0; k < VEC; ++k) v.v[k] = v.v[k] > 0.f ? v.v[k] : a * v.v[k];
    v.store(y + r * ldy + c);
  }
}

template <class T, int VEC>
__global__ void __launch_bounds__(256) prelu_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx, const float* __restrict__ slope,
                                                        T* __restrict__ dx, int64_t lddx, float* __restrict__ dslope, int64_t rows, int cv,
                                                        double* __restrict__ scratch) {
  __shared__ float red[4];
  const float a = slope[0];
  const int64_t total = rows * cv;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    V<T, VEC> g, xv;
    g.load(dy + r * lddy + c);
    xv.load(x + r * ldx + c);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (!(xv.v[k] > 0.f)) { acc = fmaf(g.v[k], xv.v[k], acc); g.v[k] *= a; }
    }
    g.store(dx + r * lddx + c);
  }
  if (dslope) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      // the slope gradient is ONE number, a sum over every voxel with heavy cancellation: the workgroup partials meet in float64 (fp32 atomics
      // in arrival order were off by up to 60 % on the 64^3 UNet and not reproducible), the last workgroup to arrive hands the total over.
      // The add returns its old value and the wave waits for it (s_waitcnt: the add has been performed) before it takes its arrival
      // ticket - a dependency written in C++ ("cond ? 1 : 1") is folded away and the ticket could overtake the add.
      const double part = ((double)red[0] + (double)red[1]) + ((double)red[2] + (double)red[3]);
      const double before = atomicAdd(scratch, part);
      asm volatile("s_waitcnt vmcnt(0)" : : "v"(before) : "memory");
      unsigned long long* ticket = reinterpret_cast<unsigned long long*>(scratch + 1);
      const unsigned long long arrived = atomicAdd(ticket, 1ull);
      if (arrived == (unsigned long long)gridDim.x - 1) atomicAdd(dslope, (float)atomicAdd(scratch, 0.0));
    }
  }
}

// rows [B][S][C] <-> NCDHW fp32; the tile transpose keeps both sides coalesced
template <class T>
__global__ void __launch_bounds__(256) layout_ncdhw_kernel(const T* __restrict__ rows_in, T* __restrict__ rows_out, int64_t ld, const float* __restrict__ nc_in,
                                                           float* __restrict__ nc_out, int C, int64_t S, int dir) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int64_t s0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  if (!dir) {
    for (int j = ty; j < 32; j += 8)
      if (s0 + j < S && c0 + tx < C) tile[j][tx] = to_f32(rows_in[((int64_t)b * S + s0 + j) * ld + c0 + tx]);
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
      if (c0 + j < C && s0 + tx < S) nc_out[((int64_t)b * C + c0 + j) * S + s0 + tx] = tile[tx][j];
  } else {
    for (int j = ty; j < 32; j += 8)
      if (c0 + j < C && s0 + tx < S) tile[j][tx] = nc_in[((int64_t)b * C + c0 + j) * S + s0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
      if (s0 + j < S && c0 + tx < C) rows_out[((int64_t)b * S + s0 + j) * ld + c0 + tx] = from_f32<T>(tile[tx][j]);
  }
}
}  // namespace miseg

extern "C" int miseg_resample2(const miseg_resample2_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->y && p->B > 0 && p->D > 0 && p->H > 0 && p->W > 0 && p->C > 0, MISEG_E_BADARG, "resample2: bad args");
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->ldx % N == 0 && p->ldy % N == 0 && al16(p->x) && al16(p->y);
    const int64_t vox = p->dir ? (int64_t)p->B * p->D * p->H * p->W : (int64_t)p->B * ((p->D + 1) / 2) * ((p->H + 1) / 2) * ((p->W + 1) / 2);
    if (vec) miseg::resample2_kernel<T, N><<<ew_grid(vox * (p->C / N)), 256, 0, s>>>((const T*)p->x, p->ldx, (T*)p->y, p->ldy, p->B, p->D, p->H, p->W, p->C / N, p->dir);
    else miseg::resample2_kernel<T, 1><<<ew_grid(vox * p->C), 256, 0, s>>>((const T*)p->x, p->ldx, (T*)p->y, p->ldy, p->B, p->D, p->H, p->W, p->C, p->dir);
    MISEG_LAUNCH_CHECK("resample2");
  });
}

extern "C" int miseg_rowbias_add(const miseg_rowbias_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->y && p->bias && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "rowbias_add: bad args");
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->ldx % N == 0 && p->ldy % N == 0 && al16(p->x) && al16(p->y);
    if (vec) miseg::rowbias_kernel<T, N><<<ew_grid(p->rows * (p->C / N)), 256, 0, s>>>((const T*)p->x, p->ldx, p->bias, (T*)p->y, p->ldy, p->rows, p->C / N);
    else miseg::rowbias_kernel<T, 1><<<ew_grid(p->rows * p->C), 256, 0, s>>>((const T*)p->x, p->ldx, p->bias, (T*)p->y, p->ldy, p->rows, p->C);
    MISEG_LAUNCH_CHECK("rowbias_add");
  });
}

extern "C" int miseg_prelu_fwd(const miseg_prelu_fwd_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->x && p->y && p->slope && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "prelu_fwd: bad args");
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->ldx % N == 0 && p->ldy % N == 0 && al16(p->x) && al16(p->y);
    if (vec) miseg::prelu_fwd_kernel<T, N><<<ew_grid(p->rows * (p->C / N)), 256, 0, s>>>((const T*)p->x, p->ldx, p->slope, (T*)p->y, p->ldy, p->rows, p->C / N);
    else miseg::prelu_fwd_kernel<T, 1><<<ew_grid(p->rows * p->C), 256, 0, s>>>((const T*)p->x, p->ldx, p->slope, (T*)p->y, p->ldy, p->rows, p->C);
    MISEG_LAUNCH_CHECK("prelu_fwd");
  });
}

extern "C" int miseg_prelu_bwd(const miseg_prelu_bwd_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->dy && p->x && p->dx && p->slope && p->rows > 0 && p->C > 0, MISEG_E_BADARG, "prelu_bwd: bad args");
  MISEG_REQUIRE(!p->dslope || p->scratch, MISEG_E_BADARG, "prelu_bwd: dslope needs the zeroed double[2] scratch");
  DT(p, {
    constexpr int N = Vec16<T>::N;
    const bool vec = p->C % N == 0 && p->lddy % N == 0 && p->ldx % N == 0 && p->lddx % N == 0 && al16(p->dy) && al16(p->x) && al16(p->dx);
    int64_t n = vec ? p->rows * (p->C / N) : p->rows * p->C;
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;       // one atomic per workgroup on the single slope gradient
    if (vec) miseg::prelu_bwd_kernel<T, N><<<grid, 256, 0, s>>>((const T*)p->dy, p->lddy, (const T*)p->x, p->ldx, p->slope, (T*)p->dx, p->lddx, p->dslope, p->rows, p->C / N, p->scratch);
    else miseg::prelu_bwd_kernel<T, 1><<<grid, 256, 0, s>>>((const T*)p->dy, p->lddy, (const T*)p->x, p->ldx, p->slope, (T*)p->dx, p->lddx, p->dslope, p->rows, p->C, p->scratch);
    MISEG_LAUNCH_CHECK("prelu_bwd");
  });
}

extern "C" int miseg_layout_ncdhw(const void* rows_, int64_t ld, float* ncdhw, int B, int C, int64_t S, int dtype, int dir, miseg_stream_t s_) {
  MISEG_REQUIRE(rows_ && ncdhw && B > 0 && C > 0 && S > 0 && ld >= C, MISEG_E_BADARG, "layout_ncdhw: bad args");
  return dispatch_dtype(dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    dim3 grid((unsigned)((S + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    miseg::layout_ncdhw_kernel<T><<<grid, 256, 0, (hipStream_t)s_>>>((const T*)rows_, (T*)const_cast<void*>(rows_), ld, ncdhw, ncdhw, C, S, dir);
    MISEG_LAUNCH_CHECK("layout_ncdhw");
    return MISEG_OK;
  });
}

extern "C" int miseg_colsum_batch(const miseg_colsum_desc* descs, int n, int dtype, miseg_stream_t s_) {
  MISEG_REQUIRE(descs && n > 0 && n <= MISEG_COLSUM_BATCH, MISEG_E_BADARG, "colsum_batch: 1..%d descriptors", MISEG_COLSUM_BATCH);
  miseg::ColsumBatch b;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    MISEG_REQUIRE(descs[i].x && descs[i].out && descs[i].rows > 0 && descs[i].C > 0, MISEG_E_BADARG, "colsum_batch: descriptor %d", i);
    b.d[i] = descs[i];
    b.d[i].block0 = blocks;
    blocks += (int)((descs[i].rows + miseg::CSB_ROWS - 1) / miseg::CSB_ROWS) * ((descs[i].C + miseg::CSB_CH - 1) / miseg::CSB_CH);
  }
  b.n = n;
  return dispatch_dtype(dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    miseg::colsum_batch_kernel<T><<<blocks, 256, 0, (hipStream_t)s_>>>(b);
    MISEG_LAUNCH_CHECK("colsum_batch");
    return MISEG_OK;
  });
}

// up to MISEG_FILL_RANGES word ranges of one buffer in ONE launch (the gradient arena minus the slots a kernel will overwrite whole, round 5):
// ranges are 16-byte aligned (arena slots are) and a multiple of 4 words long except possibly the last
namespace miseg {
struct FillRanges { uint64_t off[MISEG_FILL_RANGES], len[MISEG_FILL_RANGES], start[MISEG_FILL_RANGES + 1]; int n; };
static __global__ void __launch_bounds__(256) fill_ranges_kernel(uint32_t* __restrict__ dst, uint32_t v, FillRanges r) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  const u32x4 vv = {v, v, v, v};
  const uint64_t total4 = r.start[r.n];                       // in 16-byte units over all ranges (each range rounded up)
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (uint64_t)gridDim.x * 256) {
    int k = 0;
    while (k + 1 < r.n && r.start[k + 1] <= i) ++k;
    const uint64_t w = (i - r.start[k]) * 4;                  // word offset inside range k
    uint32_t* p = dst + r.off[k] + w;
    if (w + 4 <= r.len[k]) *reinterpret_cast<u32x4*>(p) = vv;
    else for (uint64_t e = w; e < r.len[k]; ++e) dst[r.off[k] + e] = v;
  }
}
}  // namespace miseg

extern "C" int miseg_fill32_ranges(void* dst, uint32_t value, const uint64_t* ranges_host, int n, miseg_stream_t s_) {
  MISEG_REQUIRE(dst && ranges_host && n >= 1 && n <= MISEG_FILL_RANGES, MISEG_E_BADARG, "fill32_ranges: 1..%d ranges", MISEG_FILL_RANGES);
  MISEG_REQUIRE(((uintptr_t)dst % 16) == 0, MISEG_E_BADARG, "fill32_ranges: the buffer must be 16-byte aligned");
  miseg::FillRanges r;
  r.n = n;
  uint64_t acc = 0;
  for (int i = 0; i < n; ++i) {
    r.off[i] = ranges_host[2 * i];
    r.len[i] = ranges_host[2 * i + 1];
    MISEG_REQUIRE(r.off[i] % 4 == 0 && r.len[i] > 0, MISEG_E_BADARG, "fill32_ranges: range %d (offset %llu words, %llu words): offsets are multiples of 4 words", i,
                  (unsigned long long)r.off[i], (unsigned long long)r.len[i]);
    r.start[i] = acc;
    acc += (r.len[i] + 3) / 4;
  }
  r.start[n] = acc;
  uint64_t blocks = (acc + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  miseg::fill_ranges_kernel<<<(int)blocks, 256, 0, (hipStream_t)s_>>>((uint32_t*)dst, value, r);
  MISEG_LAUNCH_CHECK("fill32_ranges");
  return MISEG_OK;
}

extern "C" int miseg_fill32(void* dst, uint32_t value, size_t n, miseg_stream_t s_) {
  MISEG_REQUIRE(dst || n == 0, MISEG_E_BADARG, "fill32: null pointer");
  if (n == 0) return MISEG_OK;
  hipError_t e = miseg::fill_words_async(dst, value, n, (hipStream_t)s_);
  MISEG_REQUIRE(e == hipSuccess, MISEG_E_LAUNCH, "fill32: %s", hipGetErrorString(e));
  return MISEG_OK;
}
