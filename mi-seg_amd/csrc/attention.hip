// Fused 3D (shifted-)window attention core, forward and backward (round-1 version: fp32 VALU, one workgroup per
// (window, head), one lane per query row; K/V rows are LDS broadcasts, scores never leave registers).
// pad / roll / window partition / reverse / crop are pure index arithmetic here; the relative-position bias is
// gathered from a per-head LDS copy of the table with idx = code[i] - code[j] + centre, and the shift mask is
// derived from 27 region labels of the rolled grid (reference: swin_utils.py:107-143, window_attention.py:99-119).
#include "common.h"

namespace miseg {

struct WinGeom {
  int B, D, H, W, C, heads, hd;
  int wd, wh, ww, sd, sh, sw, tw;
  int Dp, Hp, Wp, nwd, nwh, nww, n;  // padded grid, windows per axis, tokens per window
  float scale;
};

// token t of window `win` -> row in the unpadded [B][D][H][W] grid, or -1 for a zero-padded token; also the
// region label of the rolled grid (for the shift mask) and the rel-pos code.
__device__ __forceinline__ void token_info(const WinGeom& g, int win, int t, int& row, int& label, int& code) {
  int wi = win;
  const int wx = wi % g.nww; wi /= g.nww;
  const int wy = wi % g.nwh; wi /= g.nwh;
  const int wz = wi % g.nwd;
  const int b = wi / g.nwd;
  const int tz = t / (g.wh * g.ww), ty = (t / g.ww) % g.wh, tx = t % g.ww;
  const int pz = wz * g.wd + tz, py = wy * g.wh + ty, px = wx * g.ww + tx;  // coords in the rolled padded grid
  // roll by -shift: rolled[p] = padded[(p + shift) mod P]
  int z = pz + g.sd; if (z >= g.Dp) z -= g.Dp;
  int y = py + g.sh; if (y >= g.Hp) y -= g.Hp;
  int x = px + g.sw; if (x >= g.Wp) x -= g.Wp;
  row = (z < g.D && y < g.H && x < g.W) ? ((b * g.D + z) * g.H + y) * g.W + x : -1;
  const int rz = g.sd ? (pz < g.Dp - g.wd ? 0 : (pz < g.Dp - g.sd ? 1 : 2)) : 0;
  const int ry = g.sh ? (py < g.Hp - g.wh ? 0 : (py < g.Hp - g.sh ? 1 : 2)) : 0;
  const int rx = g.sw ? (px < g.Wp - g.ww ? 0 : (px < g.Wp - g.sw ? 1 : 2)) : 0;
  label = (rz * 3 + ry) * 3 + rx;
  // reference quirk: the index table is built for a tw^3 window and sliced [:n,:n], so token t is decoded base tw
  const int tb = 2 * g.tw - 1;
  code = ((t / (g.tw * g.tw)) * tb + (t / g.tw) % g.tw) * tb + t % g.tw;
}

template <class T, int HD4>
__device__ __forceinline__ void load_head_row(const T* qkv, int64_t ldq, int row, int off, const float* bias, float* dst) {
  // dst[0 .. 4*HD4) = qkv[row][off ..] (fp32) or the bias row for a padded token
  constexpr int HD = HD4 * 4;
  if (row >= 0) {
    const T* p = qkv + (int64_t)row * ldq + off;
#pragma unroll
    for (int d = 0; d < HD; ++d) dst[d] = to_f32(p[d]);
  } else {
#pragma unroll
    for (int d = 0; d < HD; ++d) dst[d] = bias ? bias[off + d] : 0.f;
  }
}

struct AttnSmem {
  float* A;      // [n][HD]  (K in fwd / phase A, Q in phase B)
  float* Bm;     // [n][HD]  (V, then dO)
  float* table;  // [(2tw-1)^3] bias of this head
  float* dtable; // [(2tw-1)^3] (bwd)
  float* lse;    // [n]
  float* delta;  // [n]
  int* label;    // [n]
  int* code;     // [n]
  int* row;      // [n]
};

template <int HD4>
__device__ __forceinline__ AttnSmem carve(char* base, int n, int tsize, bool bwd) {
  AttnSmem s;
  constexpr int HD = HD4 * 4;
  float* f = reinterpret_cast<float*>(base);
  s.A = f; f += n * HD;
  s.Bm = f; f += n * HD;
  s.table = f; f += tsize;
  s.dtable = f; f += bwd ? tsize : 0;
  s.lse = f; f += n;
  s.delta = f; f += n;
  s.label = reinterpret_cast<int*>(f); f += n;
  s.code = reinterpret_cast<int*>(f); f += n;
  s.row = reinterpret_cast<int*>(f);
  return s;
}

static size_t attn_smem_bytes(int n, int hd, int tsize, bool bwd) {
  return ((size_t)2 * n * hd + (size_t)tsize * (bwd ? 2 : 1) + (size_t)5 * n) * sizeof(float);
}

template <class T, int HD4>
__global__ void __launch_bounds__(384) winattn_fwd_kernel(const T* __restrict__ qkv, int64_t ldq, T* __restrict__ out, int64_t ldo, const float* __restrict__ qkv_bias,
                                                          const float* __restrict__ bias_table, float* __restrict__ lse_out, WinGeom g, int tsize) {
  constexpr int HD = HD4 * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AttnSmem s = carve<HD4>(smem, g.n, tsize, false);
  const int win = blockIdx.x, head = blockIdx.y, t = threadIdx.x;
  const int n = g.n, C = g.C;
  const bool use_mask = (g.sd | g.sh | g.sw) != 0;
  int row = -1, label = 0, code = 0;
  float q[HD];
  if (t < n) {
    token_info(g, win, t, row, label, code);
    s.label[t] = label;
    s.code[t] = code;
    load_head_row<T, HD4>(qkv, ldq, row, C + head * HD, qkv_bias, s.A + t * HD);
    load_head_row<T, HD4>(qkv, ldq, row, 2 * C + head * HD, qkv_bias, s.Bm + t * HD);
    load_head_row<T, HD4>(qkv, ldq, row, head * HD, qkv_bias, q);
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] *= g.scale;
  }
  if (bias_table)
    for (int i = t; i < tsize; i += blockDim.x) s.table[i] = bias_table[(int64_t)i * g.heads + head];
  __syncthreads();
  if (t >= n) return;
  const int tb = 2 * g.tw - 1;
  const int centre = ((g.tw - 1) * tb + (g.tw - 1)) * tb + (g.tw - 1);
  float m = -INFINITY, l = 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  for (int j = 0; j < n; ++j) {
    const f32x4* kr = reinterpret_cast<const f32x4*>(s.A + j * HD);
    float sc = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < HD4; ++d4) {
      const f32x4 kv = kr[d4];
      sc = fmaf(q[4 * d4 + 0], kv[0], sc);
      sc = fmaf(q[4 * d4 + 1], kv[1], sc);
      sc = fmaf(q[4 * d4 + 2], kv[2], sc);
      sc = fmaf(q[4 * d4 + 3], kv[3], sc);
    }
    if (bias_table) sc += s.table[code - s.code[j] + centre];
    if (use_mask && s.label[j] != label) sc -= 100.f;
    const float mn = fmaxf(m, sc);
    const float alpha = __expf(m - mn), p = __expf(sc - mn);
    l = l * alpha + p;
    const f32x4* vr = reinterpret_cast<const f32x4*>(s.Bm + j * HD);
#pragma unroll
    for (int d4 = 0; d4 < HD4; ++d4) {
      const f32x4 vv = vr[d4];
      o[4 * d4 + 0] = fmaf(o[4 * d4 + 0], alpha, p * vv[0]);
      o[4 * d4 + 1] = fmaf(o[4 * d4 + 1], alpha, p * vv[1]);
      o[4 * d4 + 2] = fmaf(o[4 * d4 + 2], alpha, p * vv[2]);
      o[4 * d4 + 3] = fmaf(o[4 * d4 + 3], alpha, p * vv[3]);
    }
    m = mn;
  }
  const float inv = 1.f / l;
  lse_out[((int64_t)win * g.heads + head) * n + t] = m + __logf(l);
  if (row >= 0) {
    T* op = out + (int64_t)row * ldo + head * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[d] = from_f32<T>(o[d] * inv);
  }
}

template <class T, int HD4>
__global__ void __launch_bounds__(384) winattn_bwd_kernel(const T* __restrict__ qkv, int64_t ldq, const T* __restrict__ out, int64_t ldo, const T* __restrict__ dout,
                                                          int64_t lddo, T* __restrict__ dqkv, int64_t lddq, const float* __restrict__ qkv_bias,
                                                          const float* __restrict__ bias_table, const float* __restrict__ lse_in, float* __restrict__ dqkv_bias,
                                                          float* __restrict__ dbias_table, WinGeom g, int tsize) {
  constexpr int HD = HD4 * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AttnSmem s = carve<HD4>(smem, g.n, tsize, true);
  const int win = blockIdx.x, head = blockIdx.y, t = threadIdx.x;
  const int n = g.n, C = g.C;
  const bool use_mask = (g.sd | g.sh | g.sw) != 0;
  const int tb = 2 * g.tw - 1;
  const int centre = ((g.tw - 1) * tb + (g.tw - 1)) * tb + (g.tw - 1);
  int row = -1, label = 0, code = 0;
  float q[HD], go[HD];
  float lse = 0.f, delta = 0.f;
  if (t < n) {
    token_info(g, win, t, row, label, code);
    s.label[t] = label;
    s.code[t] = code;
    s.row[t] = row;
    load_head_row<T, HD4>(qkv, ldq, row, C + head * HD, qkv_bias, s.A + t * HD);       // K
    load_head_row<T, HD4>(qkv, ldq, row, 2 * C + head * HD, qkv_bias, s.Bm + t * HD);  // V
    load_head_row<T, HD4>(qkv, ldq, row, head * HD, qkv_bias, q);
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] *= g.scale;
    lse = lse_in[((int64_t)win * g.heads + head) * n + t];
    if (row >= 0) {  // padded rows: the forward output was cropped, so no gradient reaches them
      const T* gp = dout + (int64_t)row * lddo + head * HD;
      const T* op = out + (int64_t)row * ldo + head * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        go[d] = to_f32(gp[d]);
        delta = fmaf(go[d], to_f32(op[d]), delta);
      }
    } else {
#pragma unroll
      for (int d = 0; d < HD; ++d) go[d] = 0.f;
    }
    s.lse[t] = lse;
    s.delta[t] = delta;
  }
  for (int i = t; i < tsize; i += blockDim.x) {
    s.table[i] = bias_table ? bias_table[(int64_t)i * g.heads + head] : 0.f;
    s.dtable[i] = 0.f;
  }
  __syncthreads();
  // ---------------- phase A: lane = query i  ->  dq_i, dbias
  if (t < n) {
    float dq[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
    for (int j = 0; j < n; ++j) {
      const f32x4* kr = reinterpret_cast<const f32x4*>(s.A + j * HD);
      const f32x4* vr = reinterpret_cast<const f32x4*>(s.Bm + j * HD);
      float sc = 0.f, dp = 0.f;
      f32x4 kk[HD4];
#pragma unroll
      for (int d4 = 0; d4 < HD4; ++d4) {
        kk[d4] = kr[d4];
        const f32x4 vv = vr[d4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sc = fmaf(q[4 * d4 + e], kk[d4][e], sc);
          dp = fmaf(go[4 * d4 + e], vv[e], dp);
        }
      }
      const int bidx = code - s.code[j] + centre;
      if (bias_table) sc += s.table[bidx];
      if (use_mask && s.label[j] != label) sc -= 100.f;
      const float p = __expf(sc - lse);
      const float ds = p * (dp - delta);
      if (dbias_table) atomicAdd(&s.dtable[bidx], ds);
#pragma unroll
      for (int d4 = 0; d4 < HD4; ++d4)
#pragma unroll
        for (int e = 0; e < 4; ++e) dq[4 * d4 + e] = fmaf(ds, kk[d4][e], dq[4 * d4 + e]);
    }
    if (row >= 0) {
      T* p = dqkv + (int64_t)row * lddq + head * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) p[d] = from_f32<T>(dq[d] * g.scale);
    } else if (dqkv_bias) {
#pragma unroll
      for (int d = 0; d < HD; ++d) atomicAdd(dqkv_bias + head * HD + d, dq[d] * g.scale);
    }
  }
  __syncthreads();
  // ---------------- phase B: lane = key j  ->  dk_j, dv_j ; A <- Q (scaled), Bm <- dO
  float kreg[HD], vreg[HD];
  if (t < n) {
#pragma unroll
    for (int d = 0; d < HD; ++d) { kreg[d] = s.A[t * HD + d]; vreg[d] = s.Bm[t * HD + d]; }
  }
  __syncthreads();
  if (t < n) {
#pragma unroll
    for (int d = 0; d < HD; ++d) { s.A[t * HD + d] = q[d]; s.Bm[t * HD + d] = go[d]; }
  }
  __syncthreads();
  if (t < n) {
    float dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dk[d] = dv[d] = 0.f;
    for (int i = 0; i < n; ++i) {
      const f32x4* qr = reinterpret_cast<const f32x4*>(s.A + i * HD);
      const f32x4* gr = reinterpret_cast<const f32x4*>(s.Bm + i * HD);
      float sc = 0.f, dp = 0.f;
      f32x4 qq[HD4], gg[HD4];
#pragma unroll
      for (int d4 = 0; d4 < HD4; ++d4) {
        qq[d4] = qr[d4];
        gg[d4] = gr[d4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sc = fmaf(qq[d4][e], kreg[4 * d4 + e], sc);
          dp = fmaf(gg[d4][e], vreg[4 * d4 + e], dp);
        }
      }
      if (bias_table) sc += s.table[s.code[i] - code + centre];
      if (use_mask && s.label[i] != label) sc -= 100.f;
      const float p = __expf(sc - s.lse[i]);
      const float ds = p * (dp - s.delta[i]);
#pragma unroll
      for (int d4 = 0; d4 < HD4; ++d4)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dk[4 * d4 + e] = fmaf(ds, qq[d4][e], dk[4 * d4 + e]);   // q already carries the scale
          dv[4 * d4 + e] = fmaf(p, gg[d4][e], dv[4 * d4 + e]);
        }
    }
    if (row >= 0) {
      T* pk = dqkv + (int64_t)row * lddq + C + head * HD;
      T* pv = dqkv + (int64_t)row * lddq + 2 * C + head * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) { pk[d] = from_f32<T>(dk[d]); pv[d] = from_f32<T>(dv[d]); }
    } else if (dqkv_bias) {
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        atomicAdd(dqkv_bias + C + head * HD + d, dk[d]);
        atomicAdd(dqkv_bias + 2 * C + head * HD + d, dv[d]);
      }
    }
  }
  if (dbias_table) {
    __syncthreads();
    for (int i = t; i < tsize; i += blockDim.x) {
      const float v = s.dtable[i];
      if (v != 0.f) atomicAdd(dbias_table + (int64_t)i * g.heads + head, v);
    }
  }
}

}  // namespace miseg

using namespace miseg;

static int make_geom(const miseg_winattn_params* p, WinGeom* g) {
  MISEG_REQUIRE(p->qkv && p->out && p->lse, MISEG_E_BADARG, "winattn: null pointer");
  MISEG_REQUIRE(p->B > 0 && p->D > 0 && p->H > 0 && p->W > 0 && p->C > 0 && p->heads > 0 && p->C % p->heads == 0, MISEG_E_BADARG, "winattn: bad shape");
  MISEG_REQUIRE(p->wd > 0 && p->wh > 0 && p->ww > 0 && p->wd * p->wh * p->ww <= 384, MISEG_E_UNSUPPORTED, "winattn: window %dx%dx%d (max 384 tokens)", p->wd, p->wh,
                p->ww);
  MISEG_REQUIRE(p->sd >= 0 && p->sd < p->wd && p->sh >= 0 && p->sh < p->wh && p->sw >= 0 && p->sw < p->ww, MISEG_E_BADARG, "winattn: shift must be < window");
  g->B = p->B; g->D = p->D; g->H = p->H; g->W = p->W; g->C = p->C; g->heads = p->heads; g->hd = p->C / p->heads;
  g->wd = p->wd; g->wh = p->wh; g->ww = p->ww; g->sd = p->sd; g->sh = p->sh; g->sw = p->sw; g->tw = p->bias_table ? p->tw : 1;
  g->nwd = cdiv(p->D, p->wd); g->nwh = cdiv(p->H, p->wh); g->nww = cdiv(p->W, p->ww);
  g->Dp = g->nwd * p->wd; g->Hp = g->nwh * p->wh; g->Wp = g->nww * p->ww;
  g->n = p->wd * p->wh * p->ww;
  g->scale = p->scale;
  if (p->bias_table) {
    MISEG_REQUIRE(p->tw > 0 && p->tw * p->tw * p->tw >= g->n, MISEG_E_BADARG, "winattn: table window %d^3 smaller than the %d-token window", p->tw, g->n);
  }
  MISEG_REQUIRE(g->hd % 4 == 0 && g->hd <= 64, MISEG_E_UNSUPPORTED, "winattn: head_dim %d (need multiple of 4, <= 64)", g->hd);
  return MISEG_OK;
}

#define HD_SWITCH(hd4, ...)                                     \
  switch (hd4) {                                                \
    case 1: { constexpr int HD4 = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int HD4 = 2; __VA_ARGS__; } break;      \
    case 3: { constexpr int HD4 = 3; __VA_ARGS__; } break;      \
    case 4: { constexpr int HD4 = 4; __VA_ARGS__; } break;      \
    case 8: { constexpr int HD4 = 8; __VA_ARGS__; } break;      \
    case 16: { constexpr int HD4 = 16; __VA_ARGS__; } break;    \
    default: return set_error(MISEG_E_UNSUPPORTED, "winattn: head_dim %d not instantiated (4,8,12,16,32,64)", 4 * (hd4)); \
  }

extern "C" int miseg_winattn_fwd(const miseg_winattn_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p, MISEG_E_BADARG, "winattn_fwd: null params");
  WinGeom g;
  int rc = make_geom(p, &g);
  if (rc) return rc;
  const int tb = 2 * g.tw - 1, tsize = p->bias_table ? tb * tb * tb : 0;
  const size_t sh = attn_smem_bytes(g.n, g.hd, tsize, false);
  MISEG_REQUIRE(sh <= 160 * 1024, MISEG_E_UNSUPPORTED, "winattn_fwd: %zu bytes of LDS needed", sh);
  dim3 grid(g.B * g.nwd * g.nwh * g.nww, g.heads);
  const int threads = cdiv(g.n, 64) * 64;
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    HD_SWITCH(g.hd / 4, {
      hipFuncSetAttribute((const void*)winattn_fwd_kernel<T, HD4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      winattn_fwd_kernel<T, HD4><<<grid, threads, sh, s>>>((const T*)p->qkv, p->ldq, (T*)p->out, p->ldo, p->qkv_bias, p->bias_table, p->lse, g, tsize);
    });
    MISEG_LAUNCH_CHECK("winattn_fwd");
    return MISEG_OK;
  });
}

extern "C" int miseg_winattn_bwd(const miseg_winattn_bwd_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->dout && p->dqkv, MISEG_E_BADARG, "winattn_bwd: null pointer");
  WinGeom g;
  int rc = make_geom(&p->f, &g);
  if (rc) return rc;
  const int tb = 2 * g.tw - 1, tsize = p->f.bias_table ? tb * tb * tb : 0;
  const size_t sh = attn_smem_bytes(g.n, g.hd, tsize, true);
  MISEG_REQUIRE(sh <= 160 * 1024, MISEG_E_UNSUPPORTED, "winattn_bwd: %zu bytes of LDS needed", sh);
  dim3 grid(g.B * g.nwd * g.nwh * g.nww, g.heads);
  const int threads = cdiv(g.n, 64) * 64;
  return dispatch_dtype(p->f.dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    HD_SWITCH(g.hd / 4, {
      hipFuncSetAttribute((const void*)winattn_bwd_kernel<T, HD4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      winattn_bwd_kernel<T, HD4><<<grid, threads, sh, s>>>((const T*)p->f.qkv, p->f.ldq, (const T*)p->f.out, p->f.ldo, (const T*)p->dout, p->lddo, (T*)p->dqkv, p->lddq,
                                                           p->f.qkv_bias, p->f.bias_table, p->f.lse, p->dqkv_bias, p->dbias_table ? p->dbias_table : nullptr, g,
                                                           tsize);
    });
    MISEG_LAUNCH_CHECK("winattn_bwd");
    return MISEG_OK;
  });
}
