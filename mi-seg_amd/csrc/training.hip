// What runs right after the hot path every training / validation step, all HBM-bound:
//   * DiceFocal / DiceCE loss: one pass for the per-(b,c) sums + loss, one pass for d(loss)/d(logits)      (lightning_monai.py:46-67)
//   * Dice metric after argmax / one-hot                                                                     (lightning_monai.py:68-79,190-195)
//   * AdamW / Adam / SGD-nesterov step of EVERY parameter in one launch over the gradient arena              (lightning_monai.py:255-278)
//   * sliding-window stitching as one gather over resident window logits                                     (lightning_monai.py:86-93,187)
// MONAI 1.1.0 arithmetic restated from its public API (parity unpinned by any reference test).
#include "common.h"
#include "opt_math.h"
#include "../../include/miseg_hip_debug.h"
#include <math.h>

namespace miseg {

constexpr int LOSS_MAXC = 16;          // channels kept in registers per voxel
#ifndef MISEG_LOSS_VEC
#define MISEG_LOSS_VEC 4               // voxels per thread of the vector instantiations (4: one 16-byte load per channel)
#endif
constexpr int LOSS_VEC = MISEG_LOSS_VEC;
constexpr int LOSS_VPB = 256 * LOSS_VEC;         // voxels per workgroup (256 threads x 1 float4 group; round 5: 2048 left 1.7 workgroups per CU - a chain of
                                       // transcendentals per voxel with nobody to hide it: 50 / 59 us forward / backward on the 96^3 x 6 logits)

template <class L> __device__ __forceinline__ int label_at(const L* lab, int64_t i) { return (int)lab[i]; }

struct LossGeom {
  int B, C, kind, c0, sq;
  int64_t S;
  float gamma;
};

// value of the per-element focal term and its derivative w.r.t. x (MONAI 1.1.0 FocalLoss, sigmoid form on raw logits):
//   ce = x - x t + softplus(-x);   w = exp(gamma * logsigmoid(-x z)), z = 2 t - 1;   f = w ce
// The target is one-hot (t is 0 or 1), so with u = x z, e = exp(-|x|) = exp(-|u|) and L = log1p(e) everything comes from ONE exponential and
// ONE logarithm:  softplus(+-u) = max(+-u, 0) + L;   ce = softplus(-u);   logsigmoid(-u) = -softplus(u);   sigmoid(u) = u >= 0 ? r : e r and
// sigmoid(-u) = u >= 0 ? e r : r with r = 1 / (1 + e);   sigmoid(x) - t = -z sigmoid(-u).  (Round 5: the literal form evaluated two
// softplus and two sigmoids per channel - 5 exp, 2 log1p, 2 divisions - a dependent chain of ~250 instructions per element and the reason
// the two loss kernels took 48 / 56 us on 21 MB of logits; it also lost ce to cancellation for t = 0, x << 0, where this form is exact.)
template <bool GRAD>
__device__ __forceinline__ void focal_term(float x, float t, float gamma, float& f, float& df) {
  const bool pos = t != 0.f;
  const float u = pos ? x : -x;
  const float e = expf(-fabsf(x));
  const float L = log1pf(e);
  const float ce = fmaxf(-u, 0.f) + L;                  // softplus(-u)
  const float w = expf(-gamma * (fmaxf(u, 0.f) + L));   // exp(gamma logsigmoid(-u))
  f = w * ce;
  if (GRAD) {
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    const float su = u >= 0.f ? r : e * r, snu = u >= 0.f ? e * r : r;      // sigmoid(u), sigmoid(-u)
    const float d = w * (gamma * su * ce + snu);        // df/du
    df = pos ? -d : d;                                   // du/dx = z, and the bracket carries -z
  }
}

// softmax over channels [c0s, C) of x -> p (p[c < c0s] = 0); returns log-sum-exp
template <int MAXC> __device__ __forceinline__ float softmax_from(const float (&x)[MAXC], float (&p)[MAXC], int c0s, int C) {
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) if (c >= c0s && c < C) mx = fmaxf(mx, x[c]);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    p[c] = (c >= c0s && c < C) ? expf(x[c] - mx) : 0.f;
    sum += p[c];
  }
  const float inv = 1.f / sum;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) p[c] *= inv;
  return mx + logf(sum);
}

// grid (blocks per sample, B).  part: double [B][nblk][3 C + 1]
template <class L, int MAXC, int VEC>
__global__ void __launch_bounds__(256) seg_loss_fwd_kernel(const float* __restrict__ logits, const L* __restrict__ label, LossGeom g, double* __restrict__ part) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const int c0d = g.c0, c0s = g.kind == MISEG_LOSS_DICE_CE ? 0 : g.c0;
  const float* xb = logits + (int64_t)b * g.C * g.S;
  const L* lb = label + (int64_t)b * g.S;
  float aI[MAXC], aP[MAXC], aT[MAXC], aO = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) aI[c] = aP[c] = aT[c] = 0.f;
  const int64_t base = (int64_t)blockIdx.x * LOSS_VPB;
  for (int it = 0; it < LOSS_VPB / (256 * VEC); ++it) {
    const int64_t s0 = base + ((int64_t)it * 256 + tid) * VEC;
    if (s0 >= g.S) break;
    float xv[MAXC][VEC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < g.C) {
        if constexpr (VEC == 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(xb + (int64_t)c * g.S + s0);
#pragma unroll
          for (int v = 0; v < 4; ++v) xv[c][v] = t[v];
        } else if constexpr (VEC == 2) {
          const float2 t = *reinterpret_cast<const float2*>(xb + (int64_t)c * g.S + s0);
          xv[c][0] = t.x; xv[c][1] = t.y;
        } else xv[c][0] = xb[(int64_t)c * g.S + s0];
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) xv[c][v] = 0.f;
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (s0 + v >= g.S) break;
      const int lab = label_at(lb, s0 + v);
      float x[MAXC], p[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; ++c) x[c] = xv[c][v];
      const float lse = softmax_from<MAXC>(x, p, c0s, g.C);
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        if (c >= c0d && c < g.C) {
          const float t = (lab == c) ? 1.f : 0.f;
          aI[c] += p[c] * t;
          aP[c] += g.sq ? p[c] * p[c] : p[c];
          aT[c] += t;
          if (g.kind == MISEG_LOSS_DICE_FOCAL) {
            float f, df;
            focal_term<false>(x[c], t, g.gamma, f, df);
            aO += f;
          }
        }
      }
      if (g.kind == MISEG_LOSS_DICE_CE) {
        float xl = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) if (c == lab) xl = x[c];
        aO += lse - xl;
      }
    }
  }
  __shared__ float red[4][3 * MAXC + 1];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    if (c < g.C) {       // uniform branch
      const float i = wave_sum(aI[c]), pp = wave_sum(aP[c]), t = wave_sum(aT[c]);
      if (lane == 0) { red[wave][3 * c] = i; red[wave][3 * c + 1] = pp; red[wave][3 * c + 2] = t; }
    }
  }
  const float o = wave_sum(aO);
  if (lane == 0) red[wave][3 * MAXC] = o;
  __syncthreads();
  const int nv = 3 * g.C + 1;
  if (tid < nv) {
    const int src = tid < 3 * g.C ? tid : 3 * MAXC;
    const double v = ((double)red[0][src] + (double)red[1][src]) + ((double)red[2][src] + (double)red[3][src]);
    part[((int64_t)b * gridDim.x + blockIdx.x) * nv + tid] = v;
  }
}

// one workgroup: sums[b][c][k] = sum over blocks (fixed order), sums[3 B C] = total of the focal / CE term, loss scalar.
// One WAVE per (sample, value): a lane adds every 64th block partial, the 64 lane sums meet in a shuffle tree (round 5: the 256-thread LDS
// tree took 8 barriers per value, 19 values per sample one after the other - 23 us for what is 7 additions per lane).
static __global__ void __launch_bounds__(1024) seg_loss_finalize_kernel(const double* __restrict__ part, int nblk, LossGeom g, float nr, float dr, float ld, float lo,
                                                                     double* __restrict__ sums, float* __restrict__ loss) {
  const int nv = 3 * g.C + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  __shared__ double other[64];
  for (int job = wave; job < g.B * nv; job += nwave) {
    const int b = job / nv, v = job - b * nv;
    double a = 0.0;
    for (int k = lane; k < nblk; k += 64) a += part[((int64_t)b * nblk + k) * nv + v];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) {
      if (v < 3 * g.C) sums[(int64_t)b * 3 * g.C + v] = a;
      else other[b] = a;
    }
  }
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    double tot_o = 0.0, dice = 0.0;
    for (int b = 0; b < g.B; ++b) {
      tot_o += other[b];
      for (int c = g.c0; c < g.C; ++c) {
        const double* q = sums + ((int64_t)b * g.C + c) * 3;
        dice += 1.0 - (2.0 * q[0] + nr) / (q[2] + q[1] + dr);
      }
    }
    const int cn = g.C - g.c0;
    sums[(int64_t)3 * g.B * g.C] = tot_o;
    const double o_mean = g.kind == MISEG_LOSS_DICE_FOCAL ? tot_o / ((double)g.B * cn * g.S) : tot_o / ((double)g.B * g.S);
    *loss = (float)(ld * dice / ((double)g.B * cn) + lo * o_mean);
  }
}

template <class L, int MAXC, int VEC>
__global__ void __launch_bounds__(256) seg_loss_bwd_kernel(const float* __restrict__ logits, const L* __restrict__ label, LossGeom g, float nr, float dr, float ld, float lo,
                                                           const double* __restrict__ sums, const float* __restrict__ gscale, float* __restrict__ dlogits) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const int c0d = g.c0, c0s = g.kind == MISEG_LOSS_DICE_CE ? 0 : g.c0;
  const int cn = g.C - g.c0;
  const float gs = gscale ? *gscale : 1.f;
  // per-channel Dice coefficients: dD/dp = ca t + cb p (squared) | ca t + cb (plain)
  float ca[MAXC], cb[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    ca[c] = cb[c] = 0.f;
    if (c >= c0d && c < g.C) {
      const double* q = sums + ((int64_t)b * g.C + c) * 3;
      const double den = q[2] + q[1] + dr, num = 2.0 * q[0] + nr, k = (double)ld * gs / ((double)g.B * cn);
      ca[c] = (float)(-2.0 * k / den);
      cb[c] = (float)((g.sq ? 2.0 : 1.0) * k * num / (den * den));
    }
  }
  const float ko = g.kind == MISEG_LOSS_DICE_FOCAL ? lo * gs / ((float)g.B * cn * (float)g.S) : lo * gs / ((float)g.B * (float)g.S);
  const float* xb = logits + (int64_t)b * g.C * g.S;
  float* db = dlogits + (int64_t)b * g.C * g.S;
  const L* lb = label + (int64_t)b * g.S;
  const int64_t base = (int64_t)blockIdx.x * LOSS_VPB;
  for (int it = 0; it < LOSS_VPB / (256 * VEC); ++it) {
    const int64_t s0 = base + ((int64_t)it * 256 + tid) * VEC;
    if (s0 >= g.S) break;
    float xv[MAXC][VEC], dv[MAXC][VEC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < g.C) {
        if constexpr (VEC == 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(xb + (int64_t)c * g.S + s0);
#pragma unroll
          for (int v = 0; v < 4; ++v) xv[c][v] = t[v];
        } else if constexpr (VEC == 2) {
          const float2 t = *reinterpret_cast<const float2*>(xb + (int64_t)c * g.S + s0);
          xv[c][0] = t.x; xv[c][1] = t.y;
        } else xv[c][0] = xb[(int64_t)c * g.S + s0];
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) xv[c][v] = 0.f;
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int lab = (s0 + v < g.S) ? label_at(lb, s0 + v) : 0;
      float x[MAXC], p[MAXC], G[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; ++c) x[c] = xv[c][v];
      softmax_from<MAXC>(x, p, c0s, g.C);
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const float t = (lab == c) ? 1.f : 0.f;
        G[c] = ca[c] * t + (g.sq ? cb[c] * p[c] : cb[c]);       // zero outside the Dice channels (ca = cb = 0)
        dot += G[c] * p[c];
      }
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const float t = (lab == c) ? 1.f : 0.f;
        float d = p[c] * (G[c] - dot);                            // softmax Jacobian; p = 0 outside the softmax support
        if (g.kind == MISEG_LOSS_DICE_FOCAL) {
          if (c >= c0d && c < g.C) {
            float f, df;
            focal_term<true>(x[c], t, g.gamma, f, df);
            d += ko * df;
          }
        } else if (c < g.C) d += ko * (p[c] - t);                 // CE: softmax over all channels is p itself (c0s = 0)
        dv[c][v] = d;
      }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < g.C) {
        if constexpr (VEC == 4) {
          f32x4 t;
#pragma unroll
          for (int v = 0; v < 4; ++v) t[v] = dv[c][v];
          *reinterpret_cast<f32x4*>(db + (int64_t)c * g.S + s0) = t;
        } else if constexpr (VEC == 2) {
          *reinterpret_cast<float2*>(db + (int64_t)c * g.S + s0) = float2{dv[c][0], dv[c][1]};
        } else db[(int64_t)c * g.S + s0] = dv[c][0];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ Dice metric
template <class L, int MAXC>
__global__ void __launch_bounds__(256) dice_count_kernel(const float* __restrict__ logits, const L* __restrict__ label, int C, int64_t S, unsigned long long* __restrict__ counts) {
  __shared__ int cnt[3 * MAXC];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < 3 * MAXC) cnt[tid] = 0;
  __syncthreads();
  const float* xb = logits + (int64_t)b * C * S;
  const L* lb = label + (int64_t)b * S;
  for (int64_t s = (int64_t)blockIdx.x * 256 + tid; s < S; s += (int64_t)gridDim.x * 256) {
    int arg = 0;
    float mx = xb[s];
    for (int c = 1; c < C; ++c) {
      const float v = xb[(int64_t)c * S + s];
      if (v > mx) { mx = v; arg = c; }            // strict: the FIRST maximum wins (torch.argmax)
    }
    const int lab = label_at(lb, s);
    atomicAdd(&cnt[3 * arg + 1], 1);
    if (lab >= 0 && lab < C) {
      atomicAdd(&cnt[3 * lab + 2], 1);
      if (lab == arg) atomicAdd(&cnt[3 * arg], 1);
    }
  }
  __syncthreads();
  if (tid < 3 * C && cnt[tid]) atomicAdd(&counts[(int64_t)b * 3 * C + tid], (unsigned long long)cnt[tid]);
}

static __global__ void dice_finalize_kernel(const unsigned long long* __restrict__ counts, int n, float* __restrict__ dice) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double inter = (double)counts[3 * i], np = (double)counts[3 * i + 1], nl = (double)counts[3 * i + 2];
  dice[i] = nl > 0 ? (float)(2.0 * inter / (np + nl)) : __int_as_float(0x7fc00000);
}

// ------------------------------------------------------------------------------------------------ optimiser
constexpr int OPT_BLOCK = 4096;     // elements per workgroup (256 threads x 4 float4)

static __global__ void __launch_bounds__(256) opt_step_kernel(const miseg_opt_desc* __restrict__ descs, int ndesc, int kind, const float* __restrict__ grad, float* __restrict__ s1,
                                                            float* __restrict__ s2, const int32_t* __restrict__ used, int32_t* __restrict__ steps, float lr, float b1, float b2,
                                                            float eps, float wd, float mom, const float* __restrict__ lr_dev, const int32_t* __restrict__ index) {
  // descriptor of this workgroup: the last one with block0 <= blockIdx.x
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const miseg_opt_desc d = descs[lo];
  const int pi = index ? index[lo] : lo;             // the parameter's row in used / steps (a table may hold a subset of the parameters)
  if (used && !used[pi]) return;
  if (lr_dev) lr = *lr_dev;
  const int blk = blockIdx.x - d.block0;
  // (steps[pi]: read by every workgroup of the tensor; written back by a separate tiny launch)
  const OptHyper hy = opt_hyper(kind, steps[pi] + 1, lr, b1, b2, eps, wd, mom);
  float* p = d.param;
  const float* gp = grad + d.off;
  float* m = s1 + d.off;
  float* v = s2 ? s2 + d.off : nullptr;
  const int e0 = blk * OPT_BLOCK;
  const bool vec = (((uintptr_t)p) & 15) == 0;      // arena slots are 16-byte aligned; a parameter view may not be
  for (int it = 0; it < 4; ++it) {
    const int e = e0 + (it * 256 + threadIdx.x) * 4;
    if (e >= d.n) break;
    const int cnt = d.n - e < 4 ? d.n - e : 4;
    float pv[4], gv[4], mv[4], vv[4];
    if (vec && cnt == 4) {
      const f32x4 tp = *reinterpret_cast<const f32x4*>(p + e), tg = *reinterpret_cast<const f32x4*>(gp + e), tm = *reinterpret_cast<const f32x4*>(m + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) { pv[k] = tp[k]; gv[k] = tg[k]; mv[k] = tm[k]; }
      if (v) {
        const f32x4 tv = *reinterpret_cast<const f32x4*>(v + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) vv[k] = tv[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = k < cnt;
        pv[k] = ok ? p[e + k] : 0.f; gv[k] = ok ? gp[e + k] : 0.f; mv[k] = ok ? m[e + k] : 0.f; vv[k] = (ok && v) ? v[e + k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) opt_update(hy, pv[k], gv[k], mv[k], vv[k]);
    if (vec && cnt == 4) {
      f32x4 tp, tm, tv;
#pragma unroll
      for (int k = 0; k < 4; ++k) { tp[k] = pv[k]; tm[k] = mv[k]; tv[k] = vv[k]; }
      *reinterpret_cast<f32x4*>(p + e) = tp;
      *reinterpret_cast<f32x4*>(m + e) = tm;
      if (v) *reinterpret_cast<f32x4*>(v + e) = tv;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < cnt) { p[e + k] = pv[k]; m[e + k] = mv[k]; if (v) v[e + k] = vv[k]; }
    }
  }
}

static __global__ void opt_count_kernel(const int32_t* __restrict__ used, int32_t* __restrict__ steps, int n, int64_t* __restrict__ params_version,
                                        int64_t* __restrict__ pack_state) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (!used || used[i])) steps[i] += 1;
  if (i == 0 && params_version) {
    *params_version += 1;      // the parameters changed: the versioned refresh kernels re-lay-out their copies
    if (pack_state) pack_state[0] = *params_version;      // ... except the packs the fused launch in front of this one has just written (opt_math.h)
  }
}

// ------------------------------------------------------------------------------------------------ stitching
struct StitchArgs {
  int C, D, H, W, rd, rh, rw, nd, nh, nw, d0;
  int sd[MISEG_STITCH_MAX_WINDOWS], sh[MISEG_STITCH_MAX_WINDOWS], sw[MISEG_STITCH_MAX_WINDOWS];
};

// windows of one axis covering coordinate x: starts are non-decreasing, so they form a contiguous index range [lo, hi]
__device__ __forceinline__ void cover(const int* st, int n, int r, int x, int& lo, int& hi) {
  lo = n; hi = -1;
  for (int i = 0; i < n; ++i)
    if (st[i] <= x && x < st[i] + r) { if (lo == n) lo = i; hi = i; }
}

// grid (ceil(W/64 / 4)..., H, D): thread = one voxel; channels looped (C small).  Reads of a window row are contiguous along w.
static __global__ void __launch_bounds__(256) stitch_kernel(const float* __restrict__ win, float* __restrict__ out, uint16_t* __restrict__ count, StitchArgs a) {
  const int w = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, d = a.d0 + blockIdx.z;
  if (w >= a.W) return;
  int dl, dh_, hl, hh, wl, wh;
  cover(a.sd, a.nd, a.rd, d, dl, dh_);
  cover(a.sh, a.nh, a.rh, h, hl, hh);
  cover(a.sw, a.nw, a.rw, w, wl, wh);
  const int n = (dh_ - dl + 1) * (hh - hl + 1) * (wh - wl + 1);
  const int64_t rvol = (int64_t)a.rd * a.rh * a.rw, vox = (int64_t)a.D * a.H * a.W;
  const int64_t o = ((int64_t)d * a.H + h) * a.W + w;
  if (count) count[o] = (uint16_t)n;
  for (int c = 0; c < a.C; ++c) {
    float acc = 0.f;
    for (int id = dl; id <= dh_; ++id)
      for (int ih = hl; ih <= hh; ++ih)
        for (int iw = wl; iw <= wh; ++iw) {
          const int64_t wi = ((int64_t)id * a.nh + ih) * a.nw + iw;
          acc += win[(wi * a.C + c) * rvol + ((int64_t)(d - a.sd[id]) * a.rh + (h - a.sh[ih])) * a.rw + (w - a.sw[iw])];
        }
    out[(int64_t)c * vox + o] = acc / (float)n;         // MONAI: output_image / count_map (a true division: bit-identical)
  }
}

// ------------------------------------------------------------------------------------------------ augmentation
struct AugArgs {
  int C, D, H, W, rd, rh, rw, n, lb;
  miseg_aug_sample s[MISEG_AUG_MAX_SAMPLES];
};

// patch coordinate (after rot90 + flips) -> coordinate inside the un-augmented crop.  The forward chain is crop -> flip0 -> flip1 -> flip2 ->
// rot90^k in the (0, 1) plane; the gather inverts it back to front.
__device__ __forceinline__ void aug_source(const miseg_aug_sample& a, int rd, int rh, int rw, int i, int j, int k, int& si, int& sj, int& sk) {
  // torch.rot90(x, k, (0, 1)): k = 1: out[i][j] = in[j][n1 - 1 - i] (n1 = in.size(1) = out.size(0))
  int pi = i, pj = j;
  const int kk = a.rot_k & 3;
  if (kk == 1) { pi = j; pj = rd - 1 - i; }
  else if (kk == 2) { pi = rd - 1 - i; pj = rh - 1 - j; }
  else if (kk == 3) { pi = rh - 1 - j; pj = i; }
  // (rd == rh whenever kk is odd: checked on the host, so the pre-rotation patch has the same extents)
  si = a.flip[0] ? rd - 1 - pi : pi;
  sj = a.flip[1] ? rh - 1 - pj : pj;
  sk = a.flip[2] ? rw - 1 - k : k;
}

template <class LB>
__global__ void __launch_bounds__(256) augment_kernel(const float* __restrict__ image, const LB* __restrict__ label, float* __restrict__ oimg, LB* __restrict__ olab, AugArgs a) {
  const int n = blockIdx.z;
  const miseg_aug_sample& sm = a.s[n];
  const int64_t pvol = (int64_t)a.rd * a.rh * a.rw, vvol = (int64_t)a.D * a.H * a.W;
  const float mul = 1.f + sm.scale;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < pvol; o += (int64_t)gridDim.x * 256) {
    const int k = (int)(o % a.rw), j = (int)((o / a.rw) % a.rh), i = (int)(o / ((int64_t)a.rw * a.rh));
    int si, sj, sk;
    aug_source(sm, a.rd, a.rh, a.rw, i, j, k, si, sj, sk);
    const int64_t src = ((int64_t)(sm.origin[0] + si) * a.H + (sm.origin[1] + sj)) * a.W + (sm.origin[2] + sk);
    for (int c = 0; c < a.C; ++c) oimg[((int64_t)n * a.C + c) * pvol + o] = image[(int64_t)c * vvol + src] * mul + sm.shift;
    if (label) olab[(int64_t)n * pvol + o] = label[src];
  }
}

// ------------------------------------------------------------------------------------------------ dropout / drop-path
// (mix64 and the key / group-hash helpers live in common.h: the attention kernels draw the same masks)

// one 64-bit hash per group of 4 consecutive channels: four 16-bit draws against a 16-bit threshold (p is quantised to 1 / 65536)
template <class T>
__global__ void __launch_bounds__(256) dropout_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int64_t rows, int C, int64_t rps,
                                                      unsigned thresh, float scale, uint64_t key, const uint64_t* __restrict__ step_dev) {
  const uint64_t k = dropout_step_key(key, step_dev);
  const int cg = (C + 3) / 4;
  const int64_t total = rows * cg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cg;
    const int c0 = (int)(i - r * cg) * 4;
    const uint64_t h = rps > 0 ? mix64(k + (uint64_t)(r / rps) * 0xD1342543DE82EF95ull) : mix64(k + (uint64_t)i * 0xD1342543DE82EF95ull);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c0 + e < C) {
        const unsigned draw = rps > 0 ? (unsigned)(h & 0xffff) : (unsigned)((h >> (16 * e)) & 0xffff);
        const float v = to_f32(x[r * ldx + c0 + e]);
        y[r * ldy + c0 + e] = from_f32<T>(draw >= thresh ? v * scale : 0.f);
      }
    }
  }
}

static __global__ void counter_add_kernel(uint64_t* c, uint64_t v) { *c += v; }
static __global__ void counter_copy_kernel(uint64_t* d, const uint64_t* s) { *d = *s; }

// ------------------------------------------------------------------------------------------------ resampling
struct ResampleArgs { int C, Di, Hi, Wi, Do, Ho, Wo; };

__device__ __forceinline__ float rs_src(int dst, int in, int out) { return ((float)dst + 0.5f) * ((float)in / (float)out) - 0.5f; }

static __global__ void __launch_bounds__(256) resample_linear_kernel(const float* __restrict__ in, float* __restrict__ out, ResampleArgs a) {
  const int64_t ovol = (int64_t)a.Do * a.Ho * a.Wo, ivol = (int64_t)a.Di * a.Hi * a.Wi;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < ovol; o += (int64_t)gridDim.x * 256) {
    const int w = (int)(o % a.Wo), h = (int)((o / a.Wo) % a.Ho), d = (int)(o / ((int64_t)a.Wo * a.Ho));
    const float fd = fminf(fmaxf(rs_src(d, a.Di, a.Do), 0.f), (float)(a.Di - 1));
    const float fh = fminf(fmaxf(rs_src(h, a.Hi, a.Ho), 0.f), (float)(a.Hi - 1));
    const float fw = fminf(fmaxf(rs_src(w, a.Wi, a.Wo), 0.f), (float)(a.Wi - 1));
    const int d0 = (int)fd, h0 = (int)fh, w0 = (int)fw;
    const int d1 = min(d0 + 1, a.Di - 1), h1 = min(h0 + 1, a.Hi - 1), w1 = min(w0 + 1, a.Wi - 1);
    const float td = fd - d0, th = fh - h0, tw = fw - w0;
    for (int c = 0; c < a.C; ++c) {
      const float* p = in + (int64_t)c * ivol;
      auto at = [&](int z, int y, int x) { return p[((int64_t)z * a.Hi + y) * a.Wi + x]; };
      const float c00 = at(d0, h0, w0) * (1.f - tw) + at(d0, h0, w1) * tw, c01 = at(d0, h1, w0) * (1.f - tw) + at(d0, h1, w1) * tw;
      const float c10 = at(d1, h0, w0) * (1.f - tw) + at(d1, h0, w1) * tw, c11 = at(d1, h1, w0) * (1.f - tw) + at(d1, h1, w1) * tw;
      out[(int64_t)c * ovol + o] = (c00 * (1.f - th) + c01 * th) * (1.f - td) + (c10 * (1.f - th) + c11 * th) * td;
    }
  }
}

template <class E>
__global__ void __launch_bounds__(256) resample_nearest_kernel(const E* __restrict__ in, E* __restrict__ out, ResampleArgs a) {
  const int64_t ovol = (int64_t)a.Do * a.Ho * a.Wo, ivol = (int64_t)a.Di * a.Hi * a.Wi;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < ovol; o += (int64_t)gridDim.x * 256) {
    const int w = (int)(o % a.Wo), h = (int)((o / a.Wo) % a.Ho), d = (int)(o / ((int64_t)a.Wo * a.Ho));
    const int sd = min(max((int)floorf(rs_src(d, a.Di, a.Do) + 0.5f), 0), a.Di - 1);
    const int sh = min(max((int)floorf(rs_src(h, a.Hi, a.Ho) + 0.5f), 0), a.Hi - 1);
    const int sw = min(max((int)floorf(rs_src(w, a.Wi, a.Wo) + 0.5f), 0), a.Wi - 1);
    for (int c = 0; c < a.C; ++c) out[(int64_t)c * ovol + o] = in[(int64_t)c * ivol + ((int64_t)sd * a.Hi + sh) * a.Wi + sw];
  }
}

template <class F> static int dispatch_label(int dt, F&& f) {
  switch (dt) {
    case MISEG_LABEL_F32: return f((const float*)nullptr);
    case MISEG_LABEL_I32: return f((const int32_t*)nullptr);
    case MISEG_LABEL_I64: return f((const int64_t*)nullptr);
    case MISEG_LABEL_U8: return f((const uint8_t*)nullptr);
  }
  return set_error(MISEG_E_BADARG, "unknown label dtype %d", dt);
}

static int loss_check(const miseg_seg_loss_params* p, const char* what) {
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_seg_loss_params), MISEG_E_BADARG, "%s: struct_size %u != %zu (header / binding drift)", what,
                p ? p->struct_size : 0u, sizeof(miseg_seg_loss_params));
  MISEG_REQUIRE(p->logits && p->label && p->sums && p->workspace, MISEG_E_BADARG, "%s: null pointer", what);
  MISEG_REQUIRE(p->B > 0 && p->B <= 64 && p->C >= 2 && p->C <= LOSS_MAXC && p->S > 0, MISEG_E_UNSUPPORTED, "%s: B %d (<= 64), C %d (2..%d)", what, p->B, p->C, LOSS_MAXC);
  MISEG_REQUIRE(p->kind == MISEG_LOSS_DICE_FOCAL || p->kind == MISEG_LOSS_DICE_CE, MISEG_E_BADARG, "%s: kind %d", what, p->kind);
  return MISEG_OK;
}

static LossGeom loss_geom(const miseg_seg_loss_params* p) {
  LossGeom g;
  g.B = p->B; g.C = p->C; g.kind = p->kind; g.c0 = p->include_background ? 0 : 1; g.sq = p->squared_pred ? 1 : 0; g.S = p->S; g.gamma = p->gamma;
  return g;
}

}  // namespace miseg

using namespace miseg;

extern "C" size_t miseg_seg_loss_workspace_bytes(int B, int C, int64_t S) {
  return (size_t)B * (size_t)cdiv(S, LOSS_VPB) * (3 * C + 1) * sizeof(double);
}

extern "C" int miseg_seg_loss_fwd(const miseg_seg_loss_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  if (int rc = loss_check(p, "seg_loss_fwd")) return rc;
  MISEG_REQUIRE(p->loss, MISEG_E_BADARG, "seg_loss_fwd: null loss pointer");
  const LossGeom g = loss_geom(p);
  const int nblk = cdiv(p->S, LOSS_VPB);
  const bool vec = p->S % 4 == 0 && ((uintptr_t)p->logits & 15) == 0;
  return dispatch_label(p->label_dtype, [&](auto* tag) -> int {
    typedef typename std::remove_const<typename std::remove_pointer<decltype(tag)>::type>::type L;
    dim3 grid(nblk, p->B);
    if (p->C <= 8) {
      if (vec) seg_loss_fwd_kernel<L, 8, LOSS_VEC><<<grid, 256, 0, s>>>(p->logits, (const L*)p->label, g, (double*)p->workspace);
      else seg_loss_fwd_kernel<L, 8, 1><<<grid, 256, 0, s>>>(p->logits, (const L*)p->label, g, (double*)p->workspace);
    } else {
      seg_loss_fwd_kernel<L, LOSS_MAXC, 1><<<grid, 256, 0, s>>>(p->logits, (const L*)p->label, g, (double*)p->workspace);
    }
    MISEG_LAUNCH_CHECK("seg_loss_fwd");
    seg_loss_finalize_kernel<<<1, 1024, 0, s>>>((const double*)p->workspace, nblk, g, p->smooth_nr, p->smooth_dr, p->lambda_dice, p->lambda_other, p->sums, p->loss);
    MISEG_LAUNCH_CHECK("seg_loss_finalize");
    return MISEG_OK;
  });
}

extern "C" int miseg_seg_loss_bwd(const miseg_seg_loss_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  if (int rc = loss_check(p, "seg_loss_bwd")) return rc;
  MISEG_REQUIRE(p->dlogits, MISEG_E_BADARG, "seg_loss_bwd: null dlogits pointer");
  const LossGeom g = loss_geom(p);
  const int nblk = cdiv(p->S, LOSS_VPB);
  const bool vec = p->S % 4 == 0 && ((uintptr_t)p->logits & 15) == 0 && ((uintptr_t)p->dlogits & 15) == 0;
  return dispatch_label(p->label_dtype, [&](auto* tag) -> int {
    typedef typename std::remove_const<typename std::remove_pointer<decltype(tag)>::type>::type L;
    dim3 grid(nblk, p->B);
#define MISEG_LB(MAXC, VEC) seg_loss_bwd_kernel<L, MAXC, VEC><<<grid, 256, 0, s>>>(p->logits, (const L*)p->label, g, p->smooth_nr, p->smooth_dr, p->lambda_dice, \
                                                                                    p->lambda_other, p->sums, p->gscale, p->dlogits)
    if (p->C <= 8) {
      if (vec) MISEG_LB(8, LOSS_VEC); else MISEG_LB(8, 1);
    } else MISEG_LB(LOSS_MAXC, 1);
#undef MISEG_LB
    MISEG_LAUNCH_CHECK("seg_loss_bwd");
    return MISEG_OK;
  });
}

extern "C" int miseg_dice_metric(const miseg_dice_metric_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_dice_metric_params), MISEG_E_BADARG, "dice_metric: struct_size %u != %zu", p ? p->struct_size : 0u,
                sizeof(miseg_dice_metric_params));
  MISEG_REQUIRE(p->logits && p->label && p->counts && p->dice, MISEG_E_BADARG, "dice_metric: null pointer");
  MISEG_REQUIRE(p->B > 0 && p->C >= 1 && p->C <= 64 && p->S > 0, MISEG_E_UNSUPPORTED, "dice_metric: C %d (1..64)", p->C);
  if (fill_words_async(p->counts, 0, (size_t)p->B * p->C * 3 * 2, s) != hipSuccess) return set_error(MISEG_E_LAUNCH, "dice_metric: fill");
  return dispatch_label(p->label_dtype, [&](auto* tag) -> int {
    typedef typename std::remove_const<typename std::remove_pointer<decltype(tag)>::type>::type L;
    int gx = cdiv(p->S, 256 * 8);
    if (gx > 2048) gx = 2048;
    dice_count_kernel<L, 64><<<dim3(gx, p->B), 256, 0, s>>>(p->logits, (const L*)p->label, p->C, p->S, (unsigned long long*)p->counts);
    MISEG_LAUNCH_CHECK("dice_count");
    dice_finalize_kernel<<<cdiv(p->B * p->C, 64), 64, 0, s>>>((const unsigned long long*)p->counts, p->B * p->C, p->dice);
    MISEG_LAUNCH_CHECK("dice_finalize");
    return MISEG_OK;
  });
}

extern "C" int miseg_opt_step(const miseg_opt_step_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_opt_step_params), MISEG_E_BADARG, "opt_step: struct_size %u != %zu", p ? p->struct_size : 0u,
                sizeof(miseg_opt_step_params));
  MISEG_REQUIRE(p->descs_dev && p->grad && p->state1 && p->steps && p->ndesc > 0 && p->total_blocks > 0, MISEG_E_BADARG, "opt_step: null pointer / empty table");
  MISEG_REQUIRE(p->kind == MISEG_OPT_ADAMW || p->kind == MISEG_OPT_ADAM || p->kind == MISEG_OPT_SGD_NESTEROV, MISEG_E_BADARG, "opt_step: kind %d", p->kind);
  MISEG_REQUIRE(p->kind == MISEG_OPT_SGD_NESTEROV || p->state2, MISEG_E_BADARG, "opt_step: Adam needs state2");
  opt_step_kernel<<<p->total_blocks, 256, 0, s>>>(p->descs_dev, p->ndesc, p->kind, p->grad, p->state1, p->kind == MISEG_OPT_SGD_NESTEROV ? nullptr : p->state2, p->used,
                                                 p->steps, p->lr, p->beta1, p->beta2, p->eps, p->weight_decay, p->momentum, p->lr_dev, p->index);
  MISEG_LAUNCH_CHECK("opt_step");
  const int cn = p->count_n < 0 ? p->ndesc : p->count_n;
  MISEG_REQUIRE(p->index || p->count_n < 0 || p->count_n == p->ndesc, MISEG_E_BADARG, "opt_step: count_n %d without an index table", p->count_n);
  if (cn > 0) {
    const int rc = opt_count_launch(p->used, p->steps, cn, p->params_version, nullptr, s);
    if (rc != MISEG_OK) return rc;
  }
  return MISEG_OK;
}

int miseg::opt_count_launch(const int32_t* used, int32_t* steps, int n, int64_t* params_version, int64_t* pack_state, hipStream_t s) {
  opt_count_kernel<<<cdiv(n, 256), 256, 0, s>>>(used, steps, n, params_version, pack_state);
  MISEG_LAUNCH_CHECK("opt_count");
  return MISEG_OK;
}

extern "C" int miseg_stitch_windows(const miseg_stitch_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_stitch_params), MISEG_E_BADARG, "stitch_windows: struct_size %u != %zu", p ? p->struct_size : 0u,
                sizeof(miseg_stitch_params));
  MISEG_REQUIRE(p->win && p->out && p->start_d && p->start_h && p->start_w, MISEG_E_BADARG, "stitch_windows: null pointer");
  MISEG_REQUIRE(p->nd > 0 && p->nh > 0 && p->nw > 0 && p->nd <= MISEG_STITCH_MAX_WINDOWS && p->nh <= MISEG_STITCH_MAX_WINDOWS && p->nw <= MISEG_STITCH_MAX_WINDOWS,
                MISEG_E_UNSUPPORTED, "stitch_windows: %d x %d x %d windows (max %d per axis)", p->nd, p->nh, p->nw, MISEG_STITCH_MAX_WINDOWS);
  MISEG_REQUIRE(p->C > 0 && p->D > 0 && p->H > 0 && p->W > 0 && p->H <= 65535 && p->D <= 65535, MISEG_E_BADARG, "stitch_windows: bad volume");
  StitchArgs a;
  a.C = p->C; a.D = p->D; a.H = p->H; a.W = p->W; a.rd = p->rd; a.rh = p->rh; a.rw = p->rw; a.nd = p->nd; a.nh = p->nh; a.nw = p->nw;
  const int* src[3] = {p->start_d, p->start_h, p->start_w};
  int* dst[3] = {a.sd, a.sh, a.sw};
  const int n[3] = {p->nd, p->nh, p->nw}, r[3] = {p->rd, p->rh, p->rw}, size[3] = {p->D, p->H, p->W};
  const bool slab = p->d_count > 0;
  MISEG_REQUIRE(p->d_count >= 0 && (!slab || (p->d_begin >= 0 && p->d_begin + p->d_count <= p->D)), MISEG_E_BADARG, "stitch_windows: slab [%d, %d + %d) of depth %d",
                p->d_begin, p->d_begin, p->d_count, p->D);
  a.d0 = slab ? p->d_begin : 0;
  for (int ax = 0; ax < 3; ++ax) {
    for (int i = 0; i < MISEG_STITCH_MAX_WINDOWS; ++i) dst[ax][i] = i < n[ax] ? src[ax][i] : 0;
    // every coordinate must be covered, windows inside the volume, starts non-decreasing: checked here, on the host, before any launch
    int reach = 0;
    for (int i = 0; i < n[ax]; ++i) {
      if (ax == 0 && slab && i == 0) {      // resident layers of a slab: the first one only has to reach back to the slab's first depth
        MISEG_REQUIRE(src[0][0] >= 0 && src[0][0] <= p->d_begin, MISEG_E_BADARG, "stitch_windows: the first resident layer starts at %d, behind the slab's first depth %d",
                      src[0][0], p->d_begin);
        reach = src[0][0];
      }
      MISEG_REQUIRE(src[ax][i] >= 0 && src[ax][i] + r[ax] <= size[ax] && (i == 0 || src[ax][i] >= src[ax][i - 1]) && src[ax][i] <= reach, MISEG_E_BADARG,
                    "stitch_windows: axis %d window %d start %d (roi %d, size %d) leaves a gap or leaves the volume", ax, i, src[ax][i], r[ax], size[ax]);
      reach = src[ax][i] + r[ax];
    }
    if (ax == 0 && slab) {
      MISEG_REQUIRE(reach >= p->d_begin + p->d_count, MISEG_E_BADARG, "stitch_windows: the resident layers cover depths up to %d, the slab ends at %d", reach,
                    p->d_begin + p->d_count);
      continue;
    }
    MISEG_REQUIRE(reach == size[ax], MISEG_E_BADARG, "stitch_windows: axis %d is covered up to %d of %d", ax, reach, size[ax]);
  }
  stitch_kernel<<<dim3(cdiv(p->W, 256), p->H, slab ? p->d_count : p->D), 256, 0, s>>>(p->win, p->out, p->count, a);
  MISEG_LAUNCH_CHECK("stitch_windows");
  return MISEG_OK;
}

extern "C" int miseg_augment_crop(const miseg_augment_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_augment_params), MISEG_E_BADARG, "augment_crop: struct_size %u != %zu", p ? p->struct_size : 0u,
                sizeof(miseg_augment_params));
  MISEG_REQUIRE(p->image && p->out_image && p->samples_host && (!p->label || p->out_label), MISEG_E_BADARG, "augment_crop: null pointer");
  MISEG_REQUIRE(p->n > 0 && p->n <= MISEG_AUG_MAX_SAMPLES && p->C > 0, MISEG_E_UNSUPPORTED, "augment_crop: %d samples (max %d)", p->n, MISEG_AUG_MAX_SAMPLES);
  MISEG_REQUIRE(p->label_bytes == 1 || p->label_bytes == 4 || p->label_bytes == 8 || !p->label, MISEG_E_UNSUPPORTED, "augment_crop: label element of %d bytes", p->label_bytes);
  AugArgs a;
  a.C = p->C; a.D = p->D; a.H = p->H; a.W = p->W; a.rd = p->rd; a.rh = p->rh; a.rw = p->rw; a.n = p->n; a.lb = p->label_bytes;
  for (int i = 0; i < p->n; ++i) {
    const miseg_aug_sample& q = p->samples_host[i];
    // every gathered coordinate stays inside the volume: checked here, on the host, before the launch
    MISEG_REQUIRE(q.origin[0] >= 0 && q.origin[1] >= 0 && q.origin[2] >= 0 && q.origin[0] + p->rd <= p->D && q.origin[1] + p->rh <= p->H && q.origin[2] + p->rw <= p->W,
                  MISEG_E_BADARG, "augment_crop: sample %d crop (%d,%d,%d)+(%d,%d,%d) leaves the %dx%dx%d volume", i, q.origin[0], q.origin[1], q.origin[2], p->rd, p->rh,
                  p->rw, p->D, p->H, p->W);
    MISEG_REQUIRE(!(q.rot_k & 1) || p->rd == p->rh, MISEG_E_BADARG, "augment_crop: an odd number of quarter turns needs roi_d == roi_h");
    a.s[i] = q;
  }
  int gx = cdiv((int64_t)p->rd * p->rh * p->rw, 256 * 4);
  if (gx > 1024) gx = 1024;
  const dim3 grid(gx, 1, p->n);
  if (!p->label || p->label_bytes == 1) augment_kernel<uint8_t><<<grid, 256, 0, s>>>(p->image, (const uint8_t*)p->label, p->out_image, (uint8_t*)p->out_label, a);
  else if (p->label_bytes == 4) augment_kernel<uint32_t><<<grid, 256, 0, s>>>(p->image, (const uint32_t*)p->label, p->out_image, (uint32_t*)p->out_label, a);
  else augment_kernel<uint64_t><<<grid, 256, 0, s>>>(p->image, (const uint64_t*)p->label, p->out_image, (uint64_t*)p->out_label, a);
  MISEG_LAUNCH_CHECK("augment_crop");
  return MISEG_OK;
}

extern "C" int miseg_resample3d(const miseg_resample3d_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_resample3d_params), MISEG_E_BADARG, "resample3d: struct_size %u != %zu", p ? p->struct_size : 0u,
                sizeof(miseg_resample3d_params));
  MISEG_REQUIRE(p->in && p->out && p->C > 0 && p->Di > 0 && p->Hi > 0 && p->Wi > 0 && p->Do > 0 && p->Ho > 0 && p->Wo > 0, MISEG_E_BADARG, "resample3d: bad arguments");
  ResampleArgs a{p->C, p->Di, p->Hi, p->Wi, p->Do, p->Ho, p->Wo};
  int grid = cdiv((int64_t)p->Do * p->Ho * p->Wo, 256 * 4);
  if (grid > 4096) grid = 4096;
  if (p->mode == 0) {
    MISEG_REQUIRE(p->elem_bytes == 4, MISEG_E_UNSUPPORTED, "resample3d: trilinear needs fp32");
    resample_linear_kernel<<<grid, 256, 0, s>>>((const float*)p->in, (float*)p->out, a);
  } else if (p->mode == 1) {
    if (p->elem_bytes == 1) resample_nearest_kernel<uint8_t><<<grid, 256, 0, s>>>((const uint8_t*)p->in, (uint8_t*)p->out, a);
    else if (p->elem_bytes == 4) resample_nearest_kernel<uint32_t><<<grid, 256, 0, s>>>((const uint32_t*)p->in, (uint32_t*)p->out, a);
    else if (p->elem_bytes == 8) resample_nearest_kernel<uint64_t><<<grid, 256, 0, s>>>((const uint64_t*)p->in, (uint64_t*)p->out, a);
    else return set_error(MISEG_E_UNSUPPORTED, "resample3d: element of %d bytes", p->elem_bytes);
  } else return set_error(MISEG_E_BADARG, "resample3d: mode %d", p->mode);
  MISEG_LAUNCH_CHECK("resample3d");
  return MISEG_OK;
}

extern "C" int miseg_dropout(const miseg_dropout_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_dropout_params), MISEG_E_BADARG, "dropout: struct_size %u != %zu", p ? p->struct_size : 0u,
                sizeof(miseg_dropout_params));
  MISEG_REQUIRE(p->x && p->y && p->rows > 0 && p->C > 0 && p->rows_per_sample >= 0, MISEG_E_BADARG, "dropout: bad arguments");
  MISEG_REQUIRE(p->p >= 0.f && p->p < 1.f, MISEG_E_BADARG, "dropout: p = %f must lie in [0, 1)", (double)p->p);
  const unsigned thresh = (unsigned)lrintf(p->p * 65536.f);
  const float scale = 1.f / (1.f - (float)thresh / 65536.f);
  const uint64_t key = dropout_host_key(p->seed, p->stream_id);
  int grid = cdiv(p->rows * ((p->C + 3) / 4), 256 * 4);
  if (grid > 4096) grid = 4096;
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    dropout_kernel<T><<<grid, 256, 0, s>>>((const T*)p->x, p->ldx, (T*)p->y, p->ldy, p->rows, p->C, p->rows_per_sample, thresh, scale, key, p->step_dev);
    MISEG_LAUNCH_CHECK("dropout");
    return MISEG_OK;
  });
}

extern "C" int miseg_counter_add(uint64_t* c, uint64_t v, miseg_stream_t s_) {
  MISEG_REQUIRE(c, MISEG_E_BADARG, "counter_add: null pointer");
  counter_add_kernel<<<1, 1, 0, (hipStream_t)s_>>>(c, v);
  MISEG_LAUNCH_CHECK("counter_add");
  return MISEG_OK;
}

// one-thread spin on a device flag (round 4): returns once *flag >= *want - set by miseg_counter_copy from another stream / another graph launch -
// or after `timeout_ticks` of the 100 MHz wall clock, in which case *timed_out is incremented and the caller's results are not to be trusted
// (every wave reaches an exit: a flag that never comes costs the timeout, not the card).  Agent-scope acquire loads: the flag is written by
// a kernel that may run on another XCD.
__global__ void flag_wait_kernel(const uint64_t* flag, const uint64_t* want, uint64_t timeout_ticks, uint32_t* timed_out) {
  const uint64_t w = *want, t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < w) {
    if (wall_clock64() - t0 > timeout_ticks) {
      if (timed_out) atomicAdd(timed_out, 1u);
      return;
    }
    __builtin_amdgcn_s_sleep(32);
  }
}

extern "C" int miseg_flag_wait(const uint64_t* flag, const uint64_t* want, uint64_t timeout_us, uint32_t* timed_out, miseg_stream_t s_) {
  MISEG_REQUIRE(flag && want, MISEG_E_BADARG, "flag_wait: null pointer");
  MISEG_REQUIRE(timeout_us > 0 && timeout_us <= 2000000, MISEG_E_BADARG, "flag_wait: timeout %llu us (1 .. 2,000,000)", (unsigned long long)timeout_us);
  flag_wait_kernel<<<1, 1, 0, (hipStream_t)s_>>>(flag, want, timeout_us * 100, timed_out);
  MISEG_LAUNCH_CHECK("flag_wait");
  return MISEG_OK;
}

__global__ void stamp_kernel(uint64_t* slot) { *slot = wall_clock64(); }

extern "C" int miseg_debug_stamp(uint64_t* slot, miseg_stream_t s_) {
  MISEG_REQUIRE(slot, MISEG_E_BADARG, "debug_stamp: null pointer");
  stamp_kernel<<<1, 1, 0, (hipStream_t)s_>>>(slot);
  MISEG_LAUNCH_CHECK("debug_stamp");
  return MISEG_OK;
}

extern "C" int miseg_counter_copy(uint64_t* d, const uint64_t* src, miseg_stream_t s_) {
  MISEG_REQUIRE(d && src, MISEG_E_BADARG, "counter_copy: null pointer");
  counter_copy_kernel<<<1, 1, 0, (hipStream_t)s_>>>(d, src);
  MISEG_LAUNCH_CHECK("counter_copy");
  return MISEG_OK;
}
