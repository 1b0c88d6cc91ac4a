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
  float* padb;   // [3*HD] bias gradient of zero-padded tokens (bwd)
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
  s.row = reinterpret_cast<int*>(f); f += n;
  s.padb = f;
  return s;
}

static size_t attn_smem_bytes(int n, int hd, int tsize, bool bwd) {
  return ((size_t)2 * n * hd + (size_t)tsize * (bwd ? 2 : 1) + (size_t)5 * n + (size_t)3 * hd) * sizeof(float);
}

// dropout on the attention probabilities (attn_drop): struct AttnDrop, common.h

// fp32 parity mode: the library expf / logf (torch's CPU softmax evaluates these); bf16 keeps the single-instruction forms
template <class T> __device__ __forceinline__ float exp_t(float x) {
  if constexpr (sizeof(T) == 4) return expf(x);
  else return __expf(x);
}
template <class T> __device__ __forceinline__ float log_t(float x) {
  if constexpr (sizeof(T) == 4) return logf(x);
  else return __logf(x);
}

template <class T, int HD4>
__global__ void __launch_bounds__(384) winattn_fwd_kernel(const T* __restrict__ qkv, int64_t ldq, T* __restrict__ out, int64_t ldo, const float* __restrict__ qkv_bias,
                                                          const float* __restrict__ bias_table, float* __restrict__ lse_out, WinGeom g, int tsize, AttnDrop dr) {
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
  // attn_drop: out = (P * mask / (1 - p)) V with P the full softmax (the denominator l sums every key, dropped or not)
  const uint64_t dk0 = dr.thresh ? dropout_step_key(dr.key, dr.step_dev) : 0ull;
  const int64_t drow = ((int64_t)win * g.heads + head) * n + t;
  const int dcg = (n + 3) / 4;
  uint64_t dh = 0;
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
    const float alpha = exp_t<T>(m - mn);
    float p = exp_t<T>(sc - mn);
    l = l * alpha + p;
    if (dr.thresh) {
      if ((j & 3) == 0) dh = dropout_group_hash(dk0, drow, dcg, j);
      p = dropout_keeps(dh, j, dr.thresh) ? p * dr.scale : 0.f;
    }
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
  lse_out[((int64_t)win * g.heads + head) * n + t] = m + log_t<T>(l);
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
                                                          float* __restrict__ dbias_table, WinGeom g, int tsize, AttnDrop dr) {
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
  for (int i = t; i < 3 * HD; i += blockDim.x) s.padb[i] = 0.f;
  __syncthreads();
  // ---------------- phase A: lane = query i  ->  dq_i, dbias
  if (t < n) {
    float dq[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
    // attn_drop: O = (P * M) V with M = mask / (1 - p), so dP = (dO V^T) * M and delta = rowsum(dO * O) as without dropout
    const uint64_t dk0 = dr.thresh ? dropout_step_key(dr.key, dr.step_dev) : 0ull;
    const int64_t drow = ((int64_t)win * g.heads + head) * n + t;
    const int dcg = (n + 3) / 4;
    uint64_t dh = 0;
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
      const float p = exp_t<T>(sc - lse);
      if (dr.thresh) {
        if ((j & 3) == 0) dh = dropout_group_hash(dk0, drow, dcg, j);
        dp = dropout_keeps(dh, j, dr.thresh) ? dp * dr.scale : 0.f;
      }
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
      for (int d = 0; d < HD; ++d) atomicAdd(&s.padb[d], dq[d] * g.scale);
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
    const uint64_t dk0 = dr.thresh ? dropout_step_key(dr.key, dr.step_dev) : 0ull;
    const int dcg = (n + 3) / 4;
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
      const float p = exp_t<T>(sc - s.lse[i]);
      float pm = p;      // the probability that multiplied V: masked and rescaled under attn_drop
      if (dr.thresh) {
        const float keep = dropout_keeps(dropout_group_hash(dk0, ((int64_t)win * g.heads + head) * n + i, dcg, t), t, dr.thresh) ? dr.scale : 0.f;
        dp *= keep;
        pm = p * keep;
      }
      const float ds = p * (dp - s.delta[i]);
#pragma unroll
      for (int d4 = 0; d4 < HD4; ++d4)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dk[4 * d4 + e] = fmaf(ds, qq[d4][e], dk[4 * d4 + e]);   // q already carries the scale
          dv[4 * d4 + e] = fmaf(pm, gg[d4][e], dv[4 * d4 + e]);
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
        atomicAdd(&s.padb[HD + d], dk[d]);
        atomicAdd(&s.padb[2 * HD + d], dv[d]);
      }
    }
  }
  __syncthreads();
  if (dbias_table) {
    for (int i = t; i < tsize; i += blockDim.x) {
      const float v = s.dtable[i];
      if (v != 0.f) atomicAdd(dbias_table + (int64_t)i * g.heads + head, v);
    }
  }
  if (dqkv_bias)
    for (int i = t; i < 3 * HD; i += blockDim.x) {
      const float v = s.padb[i];
      if (v != 0.f) atomicAdd(dqkv_bias + (i / HD) * C + head * HD + i % HD, v);
    }
}

// =====================================================================================================================
// MFMA path (bf16, head_dim 16): S^T = K Q^T on v_mfma_f32_32x32x16_bf16 with the QUERY on the lane, so the softmax
// statistics are lane-local (+ one exchange with lane^32) and the exponentiated tile is directly the B operand of
// O^T += V^T P^T (accumulator-as-operand, no LDS round trip).  One workgroup = (window, head), 4 waves, each wave owns
// query tiles of 32; keys are walked in tiles of 32.
// =====================================================================================================================
typedef __attribute__((ext_vector_type(16))) float f32x16;

static constexpr int ATT_TMAX = 2200;               // floats reserved for the bias table of a head ((2 * 7 - 1)^3 = 2197 entries)
static constexpr int ATT_MAXT = 11;                 // ceil(343 / 32) tiles
static constexpr int ATT_NP = ATT_MAXT * 32;        // 352 padded tokens
static constexpr int ATT_VT_LD = ATT_NP + 8;        // row stride (elements) of the dim-major copies
static constexpr float ATT_LOG2E = 1.4426950408889634f;   // scores are kept in the log2 domain (v_exp_f32 is exp2)

__device__ __forceinline__ int att_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// key/query info word: code (12 bits) | label << 12 (5 bits) | valid << 20
__device__ __forceinline__ int att_pack(int code, int label, bool valid) { return code | (label << 12) | ((int)valid << 20); }

__device__ __forceinline__ bf16x8 cvt8(const float* p) {
  return bf16x8{(bf16)p[0], (bf16)p[1], (bf16)p[2], (bf16)p[3], (bf16)p[4], (bf16)p[5], (bf16)p[6], (bf16)p[7]};
}

// dim-major operand (V^T, dO^T, Q^T, K^T) fragment for the accumulator-as-B products: lane (r = dim, h) element j is
// tok = t0 + 16*s + 8*(j>>2) + 4*h + (j&3); dims >= 16 are zero rows.
__device__ __forceinline__ bf16x8 att_dimmajor_frag(const bf16* base /*[16][ATT_VT_LD]*/, int r, int h, int t0, int s) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16)0.f;
  if (r < 16) {
    const bf16* p = base + r * ATT_VT_LD + t0 + 16 * s + 4 * h;
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(p + 8);
    f = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  return f;
}

__device__ __forceinline__ void att_load16v(const bf16* p, bool vec, float* dst) {
  if (vec) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p), b = *reinterpret_cast<const bf16x8*>(p + 8);
#pragma unroll
    for (int d = 0; d < 8; ++d) { dst[d] = to_f32(a[d]); dst[8 + d] = to_f32(b[d]); }
  } else {
#pragma unroll
    for (int d = 0; d < 16; ++d) dst[d] = to_f32(p[d]);
  }
}

// workgroup -> (window, head) of the matrix-core kernels (1-D grid of windows * heads).  The hardware deals consecutive workgroups to the 8
// XCDs round-robin, each with its own L2; a head reads its 32-byte slices of the window's q / k / v / dO / O rows, i.e. the same 128-byte
// lines as the window's other heads.  As grid (window, head) those ran a whole grid.x apart on other XCDs and every line came from HBM once
// per head (round 5, stage 1 backward: the operand staging alone 37 of 142 us = 174 MB at 4.7 TB/s for 59 MB of operands).  XCD k takes the
// k-th contiguous eighth of the units, heads of a window adjacent: they are dispatched back to back on one XCD and share the lines in its L2.
__device__ __forceinline__ void att_unit(int heads, int& win, int& head) {
  const int nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, r = nb & 7;
  const int unit = xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3);
  win = unit / heads;
  head = unit - win * heads;
}

// NW = waves per workgroup; a wave owns whole 32-query tiles (11 of them for a 7^3 window).  8 for the large grids (measured: 4 -> 8
// +0.7 % on the step, 11 no better), 11 - one tile per wave - when the launch has fewer workgroups than CUs (stages 3 and 4) and lasts
// exactly as long as one workgroup.
// DROP: attn_drop on the probabilities (round 3: on the matrix-core path too).  The mask is miseg_dropout's over [windows * heads * n][n]:
// one 64-bit hash per (query row, group of 4 consecutive keys) - a lane of the 32x32 accumulator holds exactly such groups (keys 8g + 4h ..
// + 3 of its query), so the cost is 4 hashes per 16 probabilities, what the query-lane kernels pay.
template <bool MASK, int NW = 4, bool DROP = false>
__global__ void __launch_bounds__(NW * 64) winattn_fwd_mfma_kernel(const bf16* __restrict__ qkv, int64_t ldq, bf16* __restrict__ out, int64_t ldo,
                                                               const float* __restrict__ qkv_bias, const float* __restrict__ bias_table,
                                                               float* __restrict__ lse_out, WinGeom g, int tsize, bool vec, AttnDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Qs = reinterpret_cast<bf16*>(smem);                 // [NP][16]  (pre-scaled, log2 domain)
  bf16* Ks = Qs + ATT_NP * 16;                              // [NP][16]
  bf16* Vt = Ks + ATT_NP * 16;                              // [16][ATT_VT_LD]
  int* kofs = reinterpret_cast<int*>(Vt + 16 * ATT_VT_LD);  // [NP] rel-pos code * 4 (byte offset into the table)
  int* klab = kofs + ATT_NP;                                // [NP] region label of the rolled grid
  int* rowq = klab + ATT_NP;                                // [NP]
  float* table = reinterpret_cast<float*>(rowq + ATT_NP);   // [tsize] * log2e
  int win, head;
  att_unit(g.heads, win, head);
  const int tid = threadIdx.x;
  const int n = g.n, C = g.C, ntiles = (n + 31) / 32;
  for (int t = tid; t < ntiles * 32; t += NW * 64) {
    int row = -1, label = 0, code = 0;
    float q[16], k[16], v[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) q[d] = k[d] = v[d] = 0.f;
    if (t < n) {
      token_info(g, win, t, row, label, code);
      if (row >= 0) {
        const bf16* p = qkv + (int64_t)row * ldq + head * 16;
        att_load16v(p, vec, q);
        att_load16v(p + C, vec, k);
        att_load16v(p + 2 * C, vec, v);
      } else if (qkv_bias) {
#pragma unroll
        for (int d = 0; d < 16; ++d) { q[d] = qkv_bias[head * 16 + d]; k[d] = qkv_bias[C + head * 16 + d]; v[d] = qkv_bias[2 * C + head * 16 + d]; }
      }
    }
    const float qs = g.scale * ATT_LOG2E;
#pragma unroll
    for (int d = 0; d < 16; ++d) q[d] *= qs;
    *reinterpret_cast<bf16x8*>(Qs + t * 16) = cvt8(q);
    *reinterpret_cast<bf16x8*>(Qs + t * 16 + 8) = cvt8(q + 8);
    *reinterpret_cast<bf16x8*>(Ks + t * 16) = cvt8(k);
    *reinterpret_cast<bf16x8*>(Ks + t * 16 + 8) = cvt8(k + 8);
#pragma unroll
    for (int d = 0; d < 16; ++d) Vt[d * ATT_VT_LD + t] = (bf16)v[d];
    kofs[t] = code * 4;
    klab[t] = label;
    rowq[t] = row;
  }
  for (int i = tid; i < tsize; i += NW * 64) table[i] = bias_table[(int64_t)i * g.heads + head] * ATT_LOG2E;
  __syncthreads();
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int tb = 2 * g.tw - 1;
  const int centre = ((g.tw - 1) * tb + (g.tw - 1)) * tb + (g.tw - 1);
  const bool ragged = (n & 31) != 0;
  constexpr float MASKV = -100.f * ATT_LOG2E;
  const uint64_t dk0 = DROP ? dropout_step_key(dr.key, dr.step_dev) : 0ull;
  const int dcg = (n + 3) / 4;
  for (int qt = wave; qt < ntiles; qt += NW) {
    const int qi = qt * 32 + r;
    const int64_t drow = ((int64_t)win * g.heads + head) * n + qi;
    const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + qi * 16 + 8 * h);
    const char* tq = reinterpret_cast<const char*>(table) + kofs[qi] + 4 * centre;   // table[code_q + centre - code_k]
    const int lq = klab[qi];
    float m = -INFINITY, l = 0.f;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // the key tile as a lambda instantiated twice: the test for keys beyond the window belongs to the LAST tile of a ragged window only - as a
    // condition inside one loop body the compiler turned it into selects executed for every tile (15 compares + 30 v_cndmask of ~175 VALU)
    auto key_tile = [&](int kt, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + r) * 16 + 8 * h);
      f32x16 sacc;
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, sacc, 0, 0, 0);   // rows = keys, col = this lane's query
      float sv[16];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int k0 = kt * 32 + 8 * gq + 4 * h;
        const int4 ko = *reinterpret_cast<const int4*>(kofs + k0);
        const int kov[4] = {ko.x, ko.y, ko.z, ko.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[4 * gq + e] = sacc[4 * gq + e] + *reinterpret_cast<const float*>(tq - kov[e]);
        if (MASK) {
          const int4 kl = *reinterpret_cast<const int4*>(klab + k0);
          const int klv[4] = {kl.x, kl.y, kl.z, kl.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) sv[4 * gq + e] += (klv[e] != lq) ? MASKV : 0.f;
        }
      }
      if constexpr (LAST) {      // keys beyond the window (only in the last tile of a ragged window)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kt * 32 + att_row(i, h) >= n) sv[i] = -INFINITY;
      }
      float mx = sv[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      float ls = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_exp2f(sv[i] - mn); ls += sv[i]; }
      ls += __shfl_xor(ls, 32, 64);
      l = l * alpha + ls;
      m = mn;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] *= alpha;
      if (DROP) {      // out = (P * mask / (1 - p)) V; the denominator l sums every key, dropped or not
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const uint64_t hh = dropout_group_hash(dk0, drow, dcg, kt * 32 + 8 * gq + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) sv[4 * gq + e] = dropout_keeps(hh, e, dr.thresh) ? sv[4 * gq + e] * dr.scale : 0.f;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = cvt8(sv + 8 * s2);
        const bf16x8 vf = att_dimmajor_frag(Vt, r, h, kt * 32, s2);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, acc, 0, 0, 0);     // O^T[dim][query] += V^T P^T
      }
    };
    const int nfull = ragged ? ntiles - 1 : ntiles;
    for (int kt = 0; kt < nfull; ++kt) key_tile(kt, std::false_type{});
    if (ragged) key_tile(ntiles - 1, std::true_type{});
    const float inv = 1.f / l;
    const int row = rowq[qi];
    if (qi < n) {
      if (h == 0) lse_out[((int64_t)win * g.heads + head) * n + qi] = (m + __log2f(l)) * 0.6931471805599453f;   // natural-log units
      if (row >= 0) {
        bf16* op = out + (int64_t)row * ldo + head * 16;
        *reinterpret_cast<bf16x4*>(op + 4 * h) = bf16x4{(bf16)(acc[0] * inv), (bf16)(acc[1] * inv), (bf16)(acc[2] * inv), (bf16)(acc[3] * inv)};
        *reinterpret_cast<bf16x4*>(op + 8 + 4 * h) = bf16x4{(bf16)(acc[4] * inv), (bf16)(acc[5] * inv), (bf16)(acc[6] * inv), (bf16)(acc[7] * inv)};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// MFMA backward (bf16, head_dim 16), single pass over 16x16 (query x key) units on v_mfma_f32_16x16x16_bf16.
//   * a wave owns key tiles kt = wave, wave+4, ... (<= 6 of 16 keys); their K/V operands and the dK^T/dV^T accumulators
//     (4 registers each) stay in registers for the whole workgroup; all waves walk the query tiles together.
//   * S' = Q'K^T - lse' and dP - delta leave the MFMA ready: -lse' and -delta are the initial accumulators
//     (log2 domain: Q' = q*scale*log2e, table' = table*log2e, p = exp2(S' + table' [+ mask'])).
//   * key on the lane: the P / dS accumulators are directly the B operands of dV^T += dO^T P and dK^T += Q'^T dS;
//     dS crosses LDS once (wave-private 16x16 tile, read back transposed) for dQ^T += K^T dS^T, which is summed over a
//     wave's key tiles in registers and over the 4 waves through a double-buffered LDS tile per query tile.
//   * A operands that need the token on the k index come from the row-major images by ds_read_b64_tr_b16 (no
//     dim-major copies); the images swap the 16-byte halves of rows with bit 3 set (conflict-free 8-byte reads).
//   * rel-pos bias gradients are LDS atomics, flushed once per workgroup.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;

static constexpr int ATT_NI = 6;                        // key tiles per wave (22 tiles of 16 / 4 waves)
static constexpr int ATT_DS_LD = 48;                    // byte stride of the wave-private dS tile rows (conflict-free)

__device__ __forceinline__ int att_chunk(int t, int c) { return t * 32 + ((c ^ ((t >> 2) & 2)) << 3); }   // byte offset of dims 4c..4c+3

__device__ __forceinline__ s16x4 att_ld4(const bf16* img, int t, int c) {
  return *reinterpret_cast<const s16x4*>(reinterpret_cast<const char*>(img) + att_chunk(t, c));
}

__device__ __forceinline__ s16x4 att_tr4(const char* p) {
  const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
  return __builtin_bit_cast(s16x4, v);
}

// lane (i = lane & 15, kg = lane >> 4) -> element [dim i][token t0 + 4kg .. +3] of a row-major image (transposed read)
__device__ __forceinline__ s16x4 att_tr_rows(const bf16* img, int t0, int fi, int kg) {
  return att_tr4(reinterpret_cast<const char*>(img) + att_chunk(t0 + 4 * kg + (fi >> 2), fi & 3));
}

__device__ __forceinline__ void att_store_row(bf16* img, int t, const float* v) {
  char* p = reinterpret_cast<char*>(img) + t * 32;
  const int sw = ((t >> 2) & 2) * 8;
  *reinterpret_cast<bf16x8*>(p + sw) = cvt8(v);
  *reinterpret_cast<bf16x8*>(p + (16 - sw)) = cvt8(v + 8);
}

__device__ __forceinline__ void att_load16(const bf16* p, bool vec, float* dst) {
  if (vec) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p), b = *reinterpret_cast<const bf16x8*>(p + 8);
#pragma unroll
    for (int d = 0; d < 8; ++d) { dst[d] = to_f32(a[d]); dst[8 + d] = to_f32(b[d]); }
  } else {
#pragma unroll
    for (int d = 0; d < 16; ++d) dst[d] = to_f32(p[d]);
  }
}

typedef unsigned pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned att_pk_bf16(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{a, b}, bf16x2_));
}

// NW = waves per workgroup (key tiles are dealt to the waves: 24 / NW each).  8 everywhere: two such workgroups still fit a CU (2 x 81 KB
// LDS, 118 VGPRs); at the deep stages the launch is ONE round of < 256 workgroups and lasts as long as one workgroup, which the extra
// waves halve (+1.2 % on the step), the large grids gain another 0.7 %.
// DROP: dS = P o (M s o dP - delta) with the forward's mask M (re-created from the key), dV += (P o M s)^T dO.  A lane of the 16x16 accumulator
// holds ONE key of four queries, i.e. four mask rows: four hashes per tile where the forward needs one per four probabilities.
template <bool MASK, bool DT /* rel-pos table gradient wanted: no per-element branch around its atomics */, int NW = 4, bool DROP = false>
__global__ void __launch_bounds__(NW * 64, 2) winattn_bwd_mfma_kernel(const bf16* __restrict__ qkv, int64_t ldq, const bf16* __restrict__ out, int64_t ldo,
                                                                  const bf16* __restrict__ dout, int64_t lddo, bf16* __restrict__ dqkv, int64_t lddq,
                                                                  const float* __restrict__ qkv_bias, const float* __restrict__ bias_table,
                                                                  const float* __restrict__ lse_in, float* __restrict__ dqkv_bias,
                                                                  float* __restrict__ dbias_table, WinGeom g, int tsize, bool vec, AttnDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS layout: the two tables FIRST - the bias gather and the bin update of a (query, key) pair share one address register (the byte offset
  // code_q - code_k) and both bases are ds_* immediates (< 64 KB); behind the big tiles the bin update paid an address add per element
  constexpr int NI = 24 / NW, NTHR = NW * 64;
  float* table = reinterpret_cast<float*>(smem);            // [ATT_TMAX] bias of this head * log2e
  int* dtable = reinterpret_cast<int*>(table + ATT_TMAX);   // [ATT_TMAX] fixed point, see `fscale`
  float* padb = reinterpret_cast<float*>(dtable + ATT_TMAX);   // [48] + 3 bound words (|dO|^2, |V|^2, |delta| maxima), padded to 64 words
  unsigned* bound = reinterpret_cast<unsigned*>(padb + 48);
  char* dstile = reinterpret_cast<char*>(padb + 64);        // [NW waves][16 keys][ATT_DS_LD]
  float* dqpart = reinterpret_cast<float*>(dstile + NW * 16 * ATT_DS_LD);   // [2][NW waves][16 queries][16 dims]
  int* qcode = reinterpret_cast<int*>(dqpart + 2 * NW * 256);
  int* qlabel = qcode + ATT_NP;
  int* rowq = qlabel + ATT_NP;
  float* nlse = reinterpret_cast<float*>(rowq + ATT_NP);   // -lse * log2e  (-inf beyond the window)
  float* ndelta = nlse + ATT_NP;                            // -rowsum(dO * O)
  bf16* Qs = reinterpret_cast<bf16*>(ndelta + ATT_NP);      // [NP][16] q * scale * log2e
  bf16* Gs = Qs + ATT_NP * 16;                              // dO
  bf16* Ks = Gs + ATT_NP * 16;
  int win, head;
  att_unit(g.heads, win, head);
  const int tid = threadIdx.x;
  const int n = g.n, C = g.C, nt16 = (n + 15) / 16;
#if defined(ATT_EXP_STOP) && ATT_EXP_STOP == 0
  return;
#endif
  if (tid < 52) padb[tid] = 0.f;
  __syncthreads();
  float mg2 = 0.f, mv2 = 0.f, mda = 0.f;
  for (int t = tid; t < nt16 * 16; t += NTHR) {
    int row = -1, label = 0, code = 0;
    float q[16], k[16], go[16];
    float dl = 0.f, ls = INFINITY, g2 = 0.f, v2 = 0.f;
#pragma unroll
    for (int d = 0; d < 16; ++d) q[d] = k[d] = go[d] = 0.f;
    if (t < n) {
      token_info(g, win, t, row, label, code);
      ls = lse_in[((int64_t)win * g.heads + head) * n + t];
      float v[16];
      if (row >= 0) {
        float o[16];
        att_load16(qkv + (int64_t)row * ldq + head * 16, vec, q);
        att_load16(qkv + (int64_t)row * ldq + C + head * 16, vec, k);
        att_load16(qkv + (int64_t)row * ldq + 2 * C + head * 16, vec, v);
        att_load16(dout + (int64_t)row * lddo + head * 16, vec, go);
        att_load16(out + (int64_t)row * ldo + head * 16, vec, o);
#pragma unroll
        for (int d = 0; d < 16; ++d) { dl = fmaf(go[d], o[d], dl); g2 = fmaf(go[d], go[d], g2); }
      } else {
#pragma unroll
        for (int d = 0; d < 16; ++d) {
          q[d] = qkv_bias ? qkv_bias[head * 16 + d] : 0.f;
          k[d] = qkv_bias ? qkv_bias[C + head * 16 + d] : 0.f;
          v[d] = qkv_bias ? qkv_bias[2 * C + head * 16 + d] : 0.f;
        }
      }
#pragma unroll
      for (int d = 0; d < 16; ++d) v2 = fmaf(v[d], v[d], v2);
    }
    mg2 = fmaxf(mg2, g2); mv2 = fmaxf(mv2, v2); mda = fmaxf(mda, fabsf(dl));
    const float qs = g.scale * ATT_LOG2E;
#pragma unroll
    for (int d = 0; d < 16; ++d) q[d] *= qs;
    att_store_row(Qs, t, q);
    att_store_row(Ks, t, k);
    att_store_row(Gs, t, go);
    qcode[t] = code * 4;            // byte offsets into the table
    qlabel[t] = label;
    rowq[t] = row;
    nlse[t] = -ls * ATT_LOG2E;
    ndelta[t] = -dl;
  }
  // workgroup maxima for the fixed-point scale of the bias-gradient bins (non-negative floats order as unsigned)
#pragma unroll
  for (int o2 = 1; o2 < 64; o2 <<= 1) { mg2 = fmaxf(mg2, __shfl_xor(mg2, o2, 64)); mv2 = fmaxf(mv2, __shfl_xor(mv2, o2, 64)); mda = fmaxf(mda, __shfl_xor(mda, o2, 64)); }
  if ((tid & 63) == 0) { atomicMax(&bound[0], __float_as_uint(mg2)); atomicMax(&bound[1], __float_as_uint(mv2)); atomicMax(&bound[2], __float_as_uint(mda)); }
  for (int i = tid; i < tsize; i += NTHR) { table[i] = bias_table[(int64_t)i * g.heads + head] * ATT_LOG2E; dtable[i] = 0; }
  __syncthreads();
  // LDS float atomics cost ~190 cycles per wave-instruction on gfx950 (integer ones 4-8): the rel-pos bias gradient is
  // binned in fixed point.  |dS_qk| <= p_qk (|dO_q||V_k| + |delta_q|) and a bin receives at most one key per query, so
  // |bin| <= n * bmax; the scale keeps two bits of headroom for the bf16 rounding of the operands.
#if defined(ATT_EXP_STOP) && ATT_EXP_STOP == 1
  if (tsize != -1) return;
#endif
  const float bmax = sqrtf(__uint_as_float(bound[0]) * __uint_as_float(bound[1])) + __uint_as_float(bound[2]);
  // (a contribution stays below 2^21 so that fma(x, fscale, 1.5 * 2^23) rounds it to an integer in the low mantissa bits: one fma + one
  // integer subtract where mul + rndne + cvt were three instructions)
  const float fscale = bmax > 0.f ? 536870912.f / ((float)(n > 256 ? n : 256) * bmax) : 0.f;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, kg = lane >> 4;   // wave: scalar, so the per-tile guards are scalar branches
  const int tb = 2 * g.tw - 1;
  const int centre = ((g.tw - 1) * tb + (g.tw - 1)) * tb + (g.tw - 1);
  constexpr float MASKV = -100.f * ATT_LOG2E;

  // ---- per-wave key tiles
  s16x4 kfB[NI], vfB[NI], kA[NI];
  f32x4 dvt[NI], dkt[NI];
  int ck[NI], lk[NI];
  const char* tkb[NI];        // table - code_k (bytes)
  bool kval = true;
  int itail = -1;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int kt = wave + NW * i;
    dvt[i] = dkt[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    kfB[i] = vfB[i] = kA[i] = s16x4{0, 0, 0, 0};
    ck[i] = lk[i] = 0;
    tkb[i] = reinterpret_cast<const char*>(table);
    if (kt < nt16) {
      const int ki = kt * 16 + fi;
      kfB[i] = att_ld4(Ks, ki, kg);                           // B operand of S:  [k = dim 4kg..][col = key]
      kA[i] = att_tr_rows(Ks, kt * 16, fi, kg);               // A operand of dQ^T: [row = dim][k = key 4kg..]
      ck[i] = qcode[ki] - 4 * centre;
      tkb[i] = reinterpret_cast<const char*>(table) - ck[i];
      lk[i] = qlabel[ki];
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (ki < n) {
        const int row = rowq[ki];
        if (row >= 0) {
          const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(qkv + (int64_t)row * ldq + 2 * C + head * 16 + 4 * kg);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = to_f32(t4[e]);
        } else if (qkv_bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = qkv_bias[2 * C + head * 16 + 4 * kg + e];
        }
      }
      vfB[i] = __builtin_bit_cast(s16x4, bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]});   // B operand of dP
      if (kt == nt16 - 1 && (n & 15)) { itail = i; kval = ki < n; }
    }
  }
#if defined(ATT_EXP_STOP) && ATT_EXP_STOP == 2
  if (tsize != -1) return;
#endif
  const uint64_t dk0 = DROP ? dropout_step_key(dr.key, dr.step_dev) : 0ull;
  const int dcg = (n + 3) / 4;
  const int64_t dbase = ((int64_t)win * g.heads + head) * n;
  char* mytile = dstile + wave * 16 * ATT_DS_LD;
  char* const ds_w = mytile + fi * ATT_DS_LD + kg * 8;                         // dS[key fi][queries 4kg..]
  const char* const ds_r = mytile + (4 * kg + (fi >> 2)) * ATT_DS_LD + (fi & 3) * 8;   // transposed read: lane (query fi, keys 4kg..)

  f32x4 padq = f32x4{0.f, 0.f, 0.f, 0.f};      // dQ of zero-padded queries (lane: dims 4 (lane & 3) ..), summed over this wave's query tiles
#ifdef ATT_EXP_QDIV
  for (int qt = 0; qt < nt16 / ATT_EXP_QDIV; ++qt) {
#else
  for (int qt = 0; qt < nt16; ++qt) {
#endif
    const int q0 = qt * 16;
    const int4 qc4 = *reinterpret_cast<const int4*>(qcode + q0 + 4 * kg);
    const int qc[4] = {qc4.x, qc4.y, qc4.z, qc4.w};
    int ql[4] = {0, 0, 0, 0};
    if (MASK) {
      const int4 ql4 = *reinterpret_cast<const int4*>(qlabel + q0 + 4 * kg);
      ql[0] = ql4.x; ql[1] = ql4.y; ql[2] = ql4.z; ql[3] = ql4.w;
    }
    const f32x4 nl4 = *reinterpret_cast<const f32x4*>(nlse + q0 + 4 * kg);
    const f32x4 nd4 = *reinterpret_cast<const f32x4*>(ndelta + q0 + 4 * kg);
    const s16x4 qfA = att_ld4(Qs, q0 + fi, kg);               // A of S:    [row = query][k = dim 4kg..]
    const s16x4 gfA = att_ld4(Gs, q0 + fi, kg);               // A of dP
    const s16x4 qtA = att_tr_rows(Qs, q0, fi, kg);            // A of dK^T: [row = dim][k = query 4kg..]
    const s16x4 gtA = att_tr_rows(Gs, q0, fi, kg);            // A of dV^T
    f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (wave + NW * i < nt16) {
        const f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qfA, kfB[i], nl4, 0, 0, 0);   // rows = queries 4kg+e, col = key fi
        const f32x4 pacc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gfA, vfB[i], DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : nd4, 0, 0, 0);   // dP - delta (DROP: dP)
        float pv[4], dsv[4], tbv[4];
        int bidx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {      // all four bias gathers first: an LDS read issued behind one of the bin atomics would wait for it
          bidx[e] = qc[e] - ck[i];
#ifdef ATT_EXP_NOBIAS
          tbv[e] = 0.f;
#else
          tbv[e] = *reinterpret_cast<const float*>(tkb[i] + qc[e]);      // = table + bidx: the key's share of the address is formed once per wave
#endif
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sc = sacc[e] + tbv[e];
          if (MASK) sc += (ql[e] != lk[i]) ? MASKV : 0.f;
          pv[e] = __builtin_amdgcn_exp2f(sc);
        }
        if (i == itail) {                  // keys beyond the window: a scalar branch taken by one tile of one wave
#pragma unroll
          for (int e = 0; e < 4; ++e) pv[e] = kval ? pv[e] : 0.f;
        }
        float pm[4];       // the probabilities that multiplied V in the forward pass
        if (DROP) {
          const int ki = (wave + NW * i) * 16 + fi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint64_t hh = dropout_group_hash(dk0, dbase + q0 + 4 * kg + e, dcg, ki);
            pm[e] = dropout_keeps(hh, ki, dr.thresh) ? pv[e] * dr.scale : 0.f;
            dsv[e] = fmaf(pm[e], pacc[e], pv[e] * nd4[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { pm[e] = pv[e]; dsv[e] = pv[e] * pacc[e]; }
        }
        // pairs through v_cvt_pk_bf16_f32 (two values per instruction, already in operand order): element-wise casts compiled to one
        // conversion per value plus v_perm packing, and a second set of conversions for the LDS copy - 18 instructions where 4 do
        const pk2 pbp = pk2{att_pk_bf16(pm[0], pm[1]), att_pk_bf16(pm[2], pm[3])};
        const pk2 dbp = pk2{att_pk_bf16(dsv[0], dsv[1]), att_pk_bf16(dsv[2], dsv[3])};
        const s16x4 pb = __builtin_bit_cast(s16x4, pbp), db = __builtin_bit_cast(s16x4, dbp);
        const bf16x4 db4 = __builtin_bit_cast(bf16x4, dbp);
        dvt[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gtA, pb, dvt[i], 0, 0, 0);   // dV^T[dim][key] += dO^T P
        dkt[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qtA, db, dkt[i], 0, 0, 0);   // dK^T[dim][key] += Q'^T dS
        *reinterpret_cast<bf16x4*>(ds_w) = db4;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const s16x4 dsT = att_tr4(ds_r);                                                 // [k = key 4kg..][col = query fi]
        dq = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kA[i], dsT, dq, 0, 0, 0);        // dQ^T[dim][query] += K^T dS^T
        __builtin_amdgcn_wave_barrier();
#ifdef ATT_EXP_NOATOM
        if (false) {
#else
        if (DT) {
#endif
#pragma unroll
          for (int e = 0; e < 4; ++e)
            atomicAdd(reinterpret_cast<int*>(reinterpret_cast<char*>(dtable) + bidx[e]), __float_as_int(fmaf(dsv[e], fscale, 12582912.f)) - 0x4B400000);
        }
      }
    }
    // partial dQ^T of this wave: lane (query fi, dims 4kg..4kg+3)
    float* part = dqpart + (qt & 1) * (NW * 256);
    *reinterpret_cast<f32x4*>(part + (wave * 16 + fi) * 16 + 4 * kg) = dq;
    __syncthreads();
    if (wave == (qt & (NW - 1))) {
      const int ql_ = lane >> 2, d4 = lane & 3, t = q0 + ql_;
      f32x4 sum = *reinterpret_cast<const f32x4*>(part + ql_ * 16 + 4 * d4);
#pragma unroll
      for (int w = 1; w < NW; ++w) sum += *reinterpret_cast<const f32x4*>(part + (w * 16 + ql_) * 16 + 4 * d4);
      if (t < n) {
        const int row = rowq[t];
        if (row >= 0) {
          *reinterpret_cast<bf16x4*>(dqkv + (int64_t)row * lddq + head * 16 + 4 * d4) =
              bf16x4{(bf16)(sum[0] * g.scale), (bf16)(sum[1] * g.scale), (bf16)(sum[2] * g.scale), (bf16)(sum[3] * g.scale)};
        } else if (dqkv_bias) {
          padq += sum;
        }
      }
    }
  }
#if defined(ATT_EXP_STOP) && ATT_EXP_STOP == 3
  if (tsize != -1) return;
#endif
  // bias gradient of the zero-padded tokens (their q / k / v rows are the bias itself): per-lane sums, folded over the lanes that hold the same
  // dims, one LDS atomic per (dim, wave).  Round 5: as one LDS float atomic per padded token and dim - 16 lanes of a wave on ONE address, ~190
  // cycles each - they were 30 of the 86 us of the 24^3 stage (37 % of its padded 28^3 grid is padding) and 5 us per query tile of a border window.
  if (dqkv_bias) {
#pragma unroll
    for (int o2 = 4; o2 < 64; o2 <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) padq[e] += __shfl_xor(padq[e], o2, 64);
    if (lane < 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (padq[e] != 0.f) atomicAdd(&padb[4 * lane + e], padq[e] * g.scale);
    }
  }
  // ---- dK, dV of the wave's keys: lane (key fi, dims 4kg..4kg+3); dK carries 1/log2e from Q'
  constexpr float LN2 = 0.6931471805599453f;
  f32x4 padk = f32x4{0.f, 0.f, 0.f, 0.f}, padv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ki = (wave + NW * i) * 16 + fi;
    if (wave + NW * i < nt16 && ki < n) {
      const int row = rowq[ki];
      if (row >= 0) {
        bf16* pk = dqkv + (int64_t)row * lddq + C + head * 16 + 4 * kg;
        *reinterpret_cast<bf16x4*>(pk) = bf16x4{(bf16)(dkt[i][0] * LN2), (bf16)(dkt[i][1] * LN2), (bf16)(dkt[i][2] * LN2), (bf16)(dkt[i][3] * LN2)};
        *reinterpret_cast<bf16x4*>(pk + C) = bf16x4{(bf16)dvt[i][0], (bf16)dvt[i][1], (bf16)dvt[i][2], (bf16)dvt[i][3]};
      } else if (dqkv_bias) {
        padk += dkt[i];
        padv += dvt[i];
      }
    }
  }
  if (dqkv_bias) {      // lanes fi = 0..15 of a k group hold the same dims: fold them, then one LDS atomic per (dim, wave)
#pragma unroll
    for (int o2 = 1; o2 < 16; o2 <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) { padk[e] += __shfl_xor(padk[e], o2, 64); padv[e] += __shfl_xor(padv[e], o2, 64); }
    if (fi == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (padk[e] != 0.f) atomicAdd(&padb[16 + 4 * kg + e], padk[e] * LN2);
        if (padv[e] != 0.f) atomicAdd(&padb[32 + 4 * kg + e], padv[e]);
      }
    }
  }
#if defined(ATT_EXP_STOP) && ATT_EXP_STOP == 4
  if (tsize != -1) return;
#endif
  __syncthreads();
#ifdef ATT_EXP_NOFLUSH
  if (false)
#else
  if (dbias_table)
#endif
    for (int i = tid; i < tsize; i += NTHR) {
      const int v = dtable[i];
      if (v != 0) atomicAdd(dbias_table + (int64_t)i * g.heads + head, (float)v / fscale);
    }
  if (dqkv_bias && tid < 48) {   // one global atomic per (q/k/v, dim) per workgroup instead of one per padded token
    const float v = padb[tid];
    if (v != 0.f) atomicAdd(dqkv_bias + (tid >> 4) * C + head * 16 + (tid & 15), v);
  }
}

static size_t attn_mfma_bwd_smem(int tsize, int nw = 4) {
  (void)tsize;
  return (size_t)3 * ATT_NP * 32 + (size_t)5 * ATT_NP * 4 + (size_t)2 * nw * 256 * 4 + (size_t)nw * 16 * ATT_DS_LD + 64 * 4 + (size_t)2 * ATT_TMAX * 4;
}

static size_t attn_mfma_fwd_smem(int tsize) {
  return (size_t)(2 * ATT_NP * 16 + 16 * ATT_VT_LD) * 2 + (size_t)3 * ATT_NP * 4 + (size_t)tsize * 4;
}

// csrc/attention_global.hip: head_dim 64 global attention on the matrix cores; return 1 when they took the call (status in *rc)
int global_attn_fwd(const miseg_winattn_params* p, const AttnDrop& dr, hipStream_t s, int* rc);
int global_attn_bwd(const miseg_winattn_bwd_params* p, const AttnDrop& dr, hipStream_t s, int* rc);
bool global_attn_takes(const miseg_winattn_params* p);

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

static AttnDrop attn_drop_args(const miseg_winattn_params* p) {
  AttnDrop dr;
  dr.thresh = (unsigned)lrintf(p->drop_p * 65536.f);      // p quantised to 1 / 65536 like miseg_dropout
  dr.scale = 1.f / (1.f - (float)dr.thresh / 65536.f);
  dr.key = dropout_host_key(p->drop_seed, p->drop_stream);
  dr.step_dev = p->drop_step_dev;
  return dr;
}

static bool fwd_takes_mfma(const miseg_winattn_params* p, const WinGeom& g) {
  return p->dtype == MISEG_BF16 && g.hd == 16 && g.n <= ATT_NP && p->bias_table && ((uintptr_t)p->out % 8 == 0) && p->ldo % 4 == 0;
}

extern "C" int miseg_winattn_on_matrix_cores(const miseg_winattn_params* p) {
  WinGeom g;
  if (!p || make_geom(p, &g)) return 0;
  return (fwd_takes_mfma(p, g) || global_attn_takes(p)) ? 1 : 0;
}

extern "C" int miseg_winattn_fwd(const miseg_winattn_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p, MISEG_E_BADARG, "winattn_fwd: null params");
  WinGeom g;
  int rc = make_geom(p, &g);
  if (rc) return rc;
  MISEG_REQUIRE(p->drop_p >= 0.f && p->drop_p < 1.f, MISEG_E_BADARG, "winattn_fwd: drop_p = %f must lie in [0, 1)", (double)p->drop_p);
  const AttnDrop dr = attn_drop_args(p);
  if (global_attn_fwd(p, dr, s, &rc)) return rc;       // one window = the whole token grid, head_dim 64, bf16 (the ViT of C-UNETR; round 4: with attention dropout too)
  const int tb = 2 * g.tw - 1, tsize = p->bias_table ? tb * tb * tb : 0;
  const size_t sh = attn_smem_bytes(g.n, g.hd, tsize, false);
  MISEG_REQUIRE(sh <= 160 * 1024, MISEG_E_UNSUPPORTED, "winattn_fwd: %zu bytes of LDS needed", sh);
  dim3 grid(g.B * g.nwd * g.nwh * g.nww, g.heads);
  const int threads = cdiv(g.n, 64) * 64;
  if (fwd_takes_mfma(p, g)) {
    const size_t shm = attn_mfma_fwd_smem(tsize);
    const bool vec = p->ldq % 8 == 0 && (uintptr_t)p->qkv % 16 == 0;
    const bool wide = (int64_t)grid.x * grid.y < 256;       // fewer workgroups than CUs: one query tile per wave
#define FWD_MFMA_D(M, NWV, D)                                                                                                                  \
  MISEG_SET_SMEM((winattn_fwd_mfma_kernel<M, NWV, D>), shm);                    \
  winattn_fwd_mfma_kernel<M, NWV, D><<<grid.x * grid.y, NWV * 64, shm, s>>>((const bf16*)p->qkv, p->ldq, (bf16*)p->out, p->ldo, p->qkv_bias, p->bias_table, \
                                                                 p->lse, g, tsize, vec, dr)
#define FWD_MFMA(M, NWV) do { if (dr.thresh) { FWD_MFMA_D(M, NWV, true); } else { FWD_MFMA_D(M, NWV, false); } } while (0)
    if ((g.sd | g.sh | g.sw) != 0) { if (wide) { FWD_MFMA(true, 11); } else { FWD_MFMA(true, 8); } }
    else { if (wide) { FWD_MFMA(false, 11); } else { FWD_MFMA(false, 8); } }
#undef FWD_MFMA
#undef FWD_MFMA_D
    MISEG_LAUNCH_CHECK("winattn_fwd_mfma");
    return MISEG_OK;
  }
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    HD_SWITCH(g.hd / 4, {
      MISEG_SET_SMEM((winattn_fwd_kernel<T, HD4>), sh);
      winattn_fwd_kernel<T, HD4><<<grid, threads, sh, s>>>((const T*)p->qkv, p->ldq, (T*)p->out, p->ldo, p->qkv_bias, p->bias_table, p->lse, g, tsize, dr);
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
  MISEG_REQUIRE(p->f.drop_p >= 0.f && p->f.drop_p < 1.f, MISEG_E_BADARG, "winattn_bwd: drop_p = %f must lie in [0, 1)", (double)p->f.drop_p);
  const AttnDrop dr = attn_drop_args(&p->f);
  if (global_attn_bwd(p, dr, s, &rc)) return rc;       // a call it declines (stricter alignment than the forward's) falls through: every forward kernel stores the log-sum-exp in natural-log units
  const int tb = 2 * g.tw - 1, tsize = p->f.bias_table ? tb * tb * tb : 0;
  const size_t sh = attn_smem_bytes(g.n, g.hd, tsize, true);
  MISEG_REQUIRE(sh <= 160 * 1024, MISEG_E_UNSUPPORTED, "winattn_bwd: %zu bytes of LDS needed", sh);
  dim3 grid(g.B * g.nwd * g.nwh * g.nww, g.heads);
  const int threads = cdiv(g.n, 64) * 64;
  if (p->f.dtype == MISEG_BF16 && g.hd == 16 && g.n <= ATT_NP && p->f.bias_table && tsize <= ATT_TMAX && ((uintptr_t)p->dqkv % 8 == 0) && p->lddq % 4 == 0 &&
      ((uintptr_t)p->f.qkv % 8 == 0) && p->f.ldq % 4 == 0) {
    const bool vec = p->f.ldq % 8 == 0 && p->f.ldo % 8 == 0 && p->lddo % 8 == 0 && (uintptr_t)p->f.qkv % 16 == 0 && (uintptr_t)p->f.out % 16 == 0 &&
                     (uintptr_t)p->dout % 16 == 0;
    // fewer workgroups than CUs (stages 3 and 4): the 8-wave form
    const bool wide = true;       // 8 waves everywhere: two such workgroups still fit a CU (2 x 81 KB LDS, 118 VGPRs) and every stage gains
    const size_t shm = attn_mfma_bwd_smem(tsize, wide ? 8 : 4);
#define BWD_MFMA_X(M, D, NWV, X)                                                                                                                            \
  MISEG_SET_SMEM((winattn_bwd_mfma_kernel<M, D, NWV, X>), shm);                              \
  winattn_bwd_mfma_kernel<M, D, NWV, X><<<grid.x * grid.y, NWV * 64, shm, s>>>((const bf16*)p->f.qkv, p->f.ldq, (const bf16*)p->f.out, p->f.ldo, (const bf16*)p->dout,    \
                                                                    p->lddo, (bf16*)p->dqkv, p->lddq, p->f.qkv_bias, p->f.bias_table, p->f.lse, p->dqkv_bias, \
                                                                    p->dbias_table, g, tsize, vec, dr)
#define BWD_MFMA(M, D, NWV) do { if (dr.thresh) { BWD_MFMA_X(M, D, NWV, true); } else { BWD_MFMA_X(M, D, NWV, false); } } while (0)
#define BWD_MFMA_W(M, D) do { if (wide) { BWD_MFMA(M, D, 8); } else { BWD_MFMA(M, D, 4); } } while (0)
    const bool masked = (g.sd | g.sh | g.sw) != 0;
    if (p->dbias_table) { if (masked) BWD_MFMA_W(true, true); else BWD_MFMA_W(false, true); }
    else { if (masked) BWD_MFMA_W(true, false); else BWD_MFMA_W(false, false); }
#undef BWD_MFMA_W
#undef BWD_MFMA
#undef BWD_MFMA_X
    MISEG_LAUNCH_CHECK("winattn_bwd_mfma");
    return MISEG_OK;
  }
  return dispatch_dtype(p->f.dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    HD_SWITCH(g.hd / 4, {
      MISEG_SET_SMEM((winattn_bwd_kernel<T, HD4>), sh);
      winattn_bwd_kernel<T, HD4><<<grid, threads, sh, s>>>((const T*)p->f.qkv, p->f.ldq, (const T*)p->f.out, p->f.ldo, (const T*)p->dout, p->lddo, (T*)p->dqkv, p->lddq,
                                                           p->f.qkv_bias, p->f.bias_table, p->f.lse, p->dqkv_bias, p->dbias_table ? p->dbias_table : nullptr, g,
                                                           tsize, dr);
    });
    MISEG_LAUNCH_CHECK("winattn_bwd");
    return MISEG_OK;
  });
}
