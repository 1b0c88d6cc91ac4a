// 3x3x3 / stride 1 / pad 1 convolution over channels-last volumes as an implicit GEMM on the matrix cores.
//
// forward (and data gradient, with the mirrored pack):
//   one workgroup = one 4x8x8 output brick x up to 96 output channels.  The (4+2)x(8+2)x(8+2) input halo of a
//   channel chunk (<= 128 bytes per voxel) is staged once in LDS; the im2col matrix is never materialised: a
//   lane's MFMA operand is the 16-byte channel group (tap, cg) of "its" voxel, read straight from the halo at
//   row (voxel + tap offset).  K = 27 * Cin is walked in groups of 16 bytes, four groups per MFMA k-step, so
//   channel counts that are not a multiple of the MFMA depth (48!) waste nothing.  Weights come from the
//   [Cout][27][CinP] pack (L2 resident, 16-byte loads).
// weight gradient:
//   one workgroup = (48 out-ch) x (48 in-ch) x 27 taps, accumulated in registers over a strided set of bricks
//   (k = voxels; operands are "transposed": ds_read_b64_tr_b16 for bf16, 4-byte reads for fp32), written as one
//   fp32 slab per workgroup and reduced by a second kernel (deterministic, no atomics).
#include <type_traits>
#include "common.h"
#include "opt_math.h"

namespace miseg {

template <class T> struct MmaC;
template <> struct MmaC<bf16> {
  static constexpr int KPC = 8;
  __device__ static __forceinline__ void run(f32x4& acc, const bf16x8& a, const bf16x8& b) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); }
};
template <> struct MmaC<float> {
  static constexpr int KPC = 4;
  __device__ static __forceinline__ void run(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
  }
};

static constexpr int BD = 4, BH = 8, BW = 8;             // output brick
static constexpr int HH = BH + 2, HW = BW + 2;            // halo extents (depth = bd + 2)

struct ConvGeom {
  int B, D, H, W;
  int nbd, nbh, nbw;
};

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <class T, int NT>
__global__ void __launch_bounds__(256) conv3_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, const T* __restrict__ wpk, ConvGeom g,
                                                        int Cin, int CinP, int Cout, int chunk_elems, int rowb, bool vec_x) {
  typedef typename Vec16<T>::type VT;
  constexpr int KPC = MmaC<T>::KPC;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  __shared__ int tapoff[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int bw = bid % g.nbw; bid /= g.nbw;
  const int bh = bid % g.nbh; bid /= g.nbh;
  const int bd = bid % g.nbd;
  const int b = bid / g.nbd;
  const int d0 = bd * BD, h0 = bh * BH, w0 = bw * BW;
  const int n0 = blockIdx.y * (16 * NT);
  if (tid < 27) tapoff[tid] = ((tid / 9) * HH + (tid / 3) % 3) * HW + tid % 3;

  const int fi = lane & 15, fq = lane >> 4;
  // halo row of this lane's voxel in M-tile mt at tap (0,0,0): d = wave, h = 2mt + (fi>>3), w = fi&7
  int vbase[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) vbase[mt] = (wave * HH + 2 * mt + (fi >> 3)) * HW + (fi & 7);

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* wrow[NT];
  bool wvalid[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = n0 + nt * 16 + fi;
    wvalid[nt] = co < Cout;
    wrow[nt] = wpk + (int64_t)(wvalid[nt] ? co : 0) * 27 * CinP;
  }

  const int HROWS = (BD + 2) * HH * HW;
  for (int c0 = 0; c0 < CinP; c0 += chunk_elems) {
    const int cl = min(chunk_elems, CinP - c0);  // elements in this chunk
    const int gpt = cl / KPC;                    // 16-byte groups per tap
    const int G = 27 * gpt;
    const unsigned inv = (65536u + gpt - 1) / gpt;
    __syncthreads();
    // ---- stage halo chunk: rows x gpt 16-byte groups
    for (int idx = tid; idx < HROWS * gpt; idx += 256) {
      const int row = idx / gpt, cg = idx - row * gpt;
      const int hd = row / (HH * HW), rem = row - hd * (HH * HW);
      const int hh = rem / HW, hw = rem - hh * HW;
      const int d = d0 - 1 + hd, h = h0 - 1 + hh, w = w0 - 1 + hw;
      VT v;
#pragma unroll
      for (int e = 0; e < KPC; ++e) v[e] = from_f32<T>(0.f);
      if (d >= 0 && d < g.D && h >= 0 && h < g.H && w >= 0 && w < g.W) {
        const int c = c0 + cg * KPC;
        const T* p = x + ((((int64_t)b * g.D + d) * g.H + h) * g.W + w) * ldx + c;
        if (vec_x && c + KPC <= Cin) v = *reinterpret_cast<const VT*>(p);
        else {
#pragma unroll
          for (int e = 0; e < KPC; ++e)
            if (c + e < Cin) v[e] = p[e];
        }
      }
      *reinterpret_cast<VT*>(lds + row * rowb + cg * 16) = v;
    }
    __syncthreads();
    // ---- K loop over (tap, group) in steps of 4 groups
    const int nsteps = (G + 3) / 4;
    // fp32 parity mode: blocked summation (every 4 k-steps = 64 products per output element join the running sum once), see conv3_fwd96_kernel
    constexpr bool BLOCKED = std::is_same<T, float>::value;
    f32x4 pacc[BLOCKED ? 4 : 1][BLOCKED ? NT : 1];
    if constexpr (BLOCKED) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pacc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int s = 0; s < nsteps; ++s) {
      int gi = 4 * s + fq;
      const bool gv = gi < G;
      gi = gv ? gi : 0;
      const int tap = (int)(((unsigned)gi * inv) >> 16);
      const int cg = gi - tap * gpt;
      const int aoff = tapoff[tap] * rowb + cg * 16;
      VT bfr[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        VT v;
#pragma unroll
        for (int e = 0; e < KPC; ++e) v[e] = from_f32<T>(0.f);
        if (gv && wvalid[nt]) v = *reinterpret_cast<const VT*>(wrow[nt] + tap * CinP + c0 + cg * KPC);
        bfr[nt] = v;
      }
      VT af[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const VT*>(lds + vbase[mt] * rowb + aoff);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (BLOCKED) MmaC<T>::run(pacc[mt][nt], bfr[nt], af[mt]);
          else MmaC<T>::run(acc[mt][nt], bfr[nt], af[mt]);
        }
      if constexpr (BLOCKED) {
        if ((s & 3) == 3 || s + 1 == nsteps) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { acc[mt][nt] += pacc[mt][nt]; pacc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
      }
    }
  }
  // ---- epilogue: lane holds channels n0 + 16nt + 4fq .. +3 of voxel (d0+wave, h0+2mt+(fi>>3), w0+(fi&7))
  const int d = d0 + wave;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int h = h0 + 2 * mt + (fi >> 3), w = w0 + (fi & 7);
    if (d < g.D && h < g.H && w < g.W) {
      T* yr = y + ((((int64_t)b * g.D + d) * g.H + h) * g.W + w) * ldy;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + nt * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < Cout) yr[co + r] = from_f32<T>(acc[mt][nt][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward, fast path: channel chunks of exactly 96 bytes (6 groups per tap: 48 bf16 / 24 fp32 channels).
//   brick 4(d) x 4(h) x 16(w): wave = d, M-tile = h, lane&15 = w  -> the 16 lanes of an operand read touch 16 consecutive
//   halo rows.  LDS images are PLANAR so that bank = row mod 16 whatever the channel group:
//     halo    [6 channel groups][656 rows (648 used)][16 B]   (plane stride = 0 mod 16 rows)
//     weights [12 k-groups of the phase][16*NT out-channels][16 B]
//   => same-(tap,group) operand reads are conflict-free (SQ_LDS_BANK_CONFLICT was 57 % of LDS cycles with row-major
//   112-byte rows).  Weights come from the planar pack [tap][group][CoutP16][16 B], staged per phase of 2 taps with a
//   linear copy, next phase prefetched into registers during the MFMAs.
// ---------------------------------------------------------------------------------------------------------
// K walk of the fast path.  A chunk (6 channel groups x 27 taps) is covered by 14 phases; a phase gives each of the four MFMA k-slots q
// one (kd, kw, channel group) "combo" and walks the three kh with it (12 weight groups = 3 MFMA k-steps):
//   phases 0..8  : kdw = kd * 3 + kw = phase, channel group q               (groups 0..3 of all nine (kd, kw))
//   phases 9..13 : kdw = 2 (phase - 9) + (q >> 1), channel group 4 + (q & 1)   (groups 4, 5; kdw == 9 in the last phase is a zero dummy)
// so a lane's halo address is (lane constant) + (compile-time displacement of the phase) - the first version decoded
// combo = 4 phase + q -> (kdw, cg) with per-lane divisions every phase, ~100 VALU cycles of address arithmetic per phase beside 576 of
// MFMA - and the weight pack is stored in exactly this order, [chunk][phase][kh * 4 + q][CoutP16][16 B], so a phase's 12 groups are one
// linear run: uniform base + lane constant (the [tap][group] pack cost three 64-bit multiply-add chains per thread and phase).
// Narrow rows (round 4).  A chunk need not be 6 groups wide: the two halves of the table above stand on their own -
//   GPT = 4 (64-byte chunks: 32 / 64 / 128 / 256 bf16 channels): phases 0..8 only, 9 phases, no dummy slot;
//   GPT = 2 (32-byte rows: 16 bf16 channels): the "groups 4, 5" half with groups 0, 1: phase = kdw >> 1, 5 phases (kdw == 9: zero dummy).
// Until then such rows were padded to 96 bytes (14 phases, a third to two thirds of the MFMAs multiplying zeros) or went to the generic kernel.
__host__ __device__ constexpr int fwd96_phases(int gpt) { return gpt == 6 ? 14 : gpt == 4 ? 9 : 5; }
__host__ __device__ constexpr int fwd96_phase_of(int gpt, int kdw, int cg) { return gpt == 2 ? (kdw >> 1) : (cg < 4 ? kdw : 9 + (kdw >> 1)); }
__host__ __device__ constexpr int fwd96_slot_of(int gpt, int kdw, int cg) { return gpt == 2 ? ((kdw & 1) << 1) + cg : (cg < 4 ? cg : ((kdw & 1) << 1) + (cg - 4)); }
static constexpr int FWD96_GROUPS = 12;

static constexpr int FBD = 4, FBH = 4, FBW = 16;
static constexpr int FHH = FBH + 2, FHW = FBW + 2;
static constexpr int FHROWS = (FBD + 2) * FHH * FHW;      // 648
static constexpr int FPS = 656;                            // plane stride in rows (multiple of 16)

// WD = how many phases ahead the weights are requested (3: a phase's 36 MFMAs per wave last ~0.3 us, an L2 round trip under load ~1 us;
// measured on 48->48 @ 96^3, same box: WD 1 / 2 / 3 = 135 / 128 / 123 us once the loads were branch-free - see wload below)
// EPI: the optional epilogue pieces (residual add, norm statistics) are compiled in; the plain instantiation carries none of it
template <class T, int NT, int WD = 1, bool EPI = false, int GPT = 6>
// (workgroups per CU the LDS images leave room for: 6 groups 81 KB -> 2; 4 groups 48 / 54 / 60 KB -> 3 / 2 / 2; 2 groups 27 / 33 / 39 KB -> 4 / 3 / 3 asked for)
__global__ void __launch_bounds__(256, GPT == 6 ? 2 : GPT == 4 ? (NT == 1 ? 3 : 2) : (NT == 1 ? 4 : 3)) conv3_fwd96_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, const T* __restrict__ wpk,
                                                             ConvGeom g, int Cin, int CinP, int Cout, int CoP, bool vec_x, bool vec_y,
                                                             float* __restrict__ scratch, int chunks_per_split, const T* __restrict__ res, int64_t ldres,
                                                             double* __restrict__ stat, int ny, const T* __restrict__ scx = nullptr, int64_t ldscx = 0,
                                                             const T* __restrict__ scw = nullptr, int Csc = 0, T* __restrict__ s2c_out = nullptr, int s2c_C = 0,
                                                             const T* __restrict__ fsw = nullptr, T* __restrict__ fsy = nullptr, int64_t ldfsy = 0,
                                                             double* __restrict__ fsstat = nullptr) {
  typedef typename Vec16<T>::type VT;
  constexpr int KPC = MmaC<T>::KPC;
  constexpr int CHUNK = GPT * KPC, NPH = fwd96_phases(GPT);
  constexpr int NROWS = 16 * NT;
  constexpr int WITEMS = 12 * NROWS;                       // 16-byte items of one weight phase
  constexpr int WLOADS = (WITEMS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* lh = lds;                                          // [GPT][FPS][16]
  char* lw = lds + GPT * FPS * 16;                         // 2 x [12][NROWS][16] (double buffer)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> brick.  The hardware deals consecutive workgroups to the 8 XCDs round-robin, each with its own L2: with bid = blockIdx.x
  // the spatial neighbours of a brick, which share a third of its halo, run on seven OTHER XCDs and every halo plane is fetched from HBM
  // by each of them (measured 2.2 x the algorithmic bytes).  XCD k takes the k-th contiguous eighth of the bricks instead.
  // The output-channel groups of a brick (ny of them: 2 for the data gradient 48 -> 96 at 96^3) are consecutive units of one XCD, so they
  // run side by side and share the halo, the residual rows and the output rows' cache lines in that L2: as grid.y they were dispatched a
  // whole grid.x apart, each group fetched the halo from HBM again and read / wrote its 96-byte halves of the 192-byte rows as partial
  // lines long after the other half had left the cache (48 -> 96 with the fused residual: 317 us against 215 without it).
  int bid, ygrp;
  {
    const int nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, r = nb & 7;
    const int unit = xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3);
    bid = unit / ny;
    ygrp = unit - bid * ny;
  }
  // inside an XCD's range the bricks advance d-fastest, then h, then w: a brick shares 2 of its 6 halo planes with its d- and with its
  // h-neighbour and only 2 of 18 columns with its w-neighbour, and the L2 (4 MB) has to hold the bricks between two neighbours for the
  // overlap to hit: one brick for d, one column of nbd bricks (1.5 MB at 96^3) for h; w-fastest needed nbh * nbw bricks (9 MB) for d
  const int bd = bid % g.nbd; bid /= g.nbd;
  const int bh = bid % g.nbh; bid /= g.nbh;
  const int bw = bid % g.nbw;
  const int b = bid / g.nbw;
  const int d0 = bd * FBD, h0 = bh * FBH, w0 = bw * FBW;
  const int n0 = ygrp * NROWS;
  const int fi = lane & 15, fq = lane >> 4;
  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // K walk.  The 27 taps x 6 channel groups of a chunk are enumerated as 54 "combos" c = (kd * 3 + kw) * 6 + cg; a phase
  // covers combos 4*phase .. +3 (one per MFMA k-group fq) for ALL THREE kh, i.e. 12 (tap, cg) weight groups = 3 MFMA steps.
  // A lane's combo is fixed for the phase, so its operand for (output row mt, tap row kh) is the halo row mt + kh of ONE
  // plane: 6 LDS reads serve the 12 (mt, kh) pairs (the tap-major walk needed 12) -- the kernel is LDS-bandwidth bound
  // (21 ds_read_b128 per 36 MFMAs per wave = 85 % of the LDS peak at the MFMA-bound rate).  Phase 13 holds combos 52, 53 only.
  VT wreg[WD][WLOADS];
  // element offset of this thread's items inside a phase of the pack.  The LOADS are unconditional straight-line code (surplus lanes
  // re-read item 0, rows beyond CoutP16 are clamped to a valid row - their output columns are never stored) and the LDS stores are
  // guarded by a wave-uniform condition (WITEMS is a multiple of 64): behind any branch around a load the compiler can no longer count
  // the loads in flight and waits with vmcnt(0) - draining the prefetch of the LATER phases too, which is why requesting weights two or
  // three phases ahead never paid while the guards were per-lane.
  int wl[WLOADS];
  bool wok[WLOADS];
#pragma unroll
  for (int i = 0; i < WLOADS; ++i) {
    const int idx = tid + 256 * i;
    wok[i] = (wave * 64 + 256 * i) < WITEMS;
    const int cidx = idx < WITEMS ? idx : 0;
    const int gk = cidx / NROWS, row = cidx - gk * NROWS;
    wl[i] = (gk * CoP + min(n0 + row, CoP - 1)) * KPC;
  }
  const int64_t wphase_stride = (int64_t)FWD96_GROUPS * CoP * KPC;
  auto wload = [&](int phase, int chunk, int slot) {
    const T* wph = wpk + ((int64_t)chunk * NPH + phase) * wphase_stride;      // uniform
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) wreg[slot][i] = *reinterpret_cast<const VT*>(wph + wl[i]);     // unconditional (surplus lanes re-read item 0)
  };
  auto wstore = [&](int buf, int slot) {
#pragma unroll
    for (int i = 0; i < WLOADS; ++i)
      if (wok[i]) *reinterpret_cast<VT*>(lw + (buf * WITEMS + tid + 256 * i) * 16) = wreg[slot][i];
  };
  // halo addresses of this lane: plane = channel group of its k-slot (see the phase table above), row of (d = wave, h = 0, w = fi)
  const int laneA = (fq * FPS + (wave * FHH) * FHW + fi) * 16;
  const int laneB = (((GPT == 6 ? 4 : 0) + (fq & 1)) * FPS + (wave * FHH) * FHW + fi) * 16;
  const bool hiB = (fq >> 1) != 0;
  const int wfrag = fi * 16;

  int* rowoff = reinterpret_cast<int*>(lw + WITEMS * 16);   // aliases weight buffer 1 (rewritten by phase 1 => rebuilt per chunk)
  const int cbeg = blockIdx.z * chunks_per_split * CHUNK, cend = min(CinP, cbeg + chunks_per_split * CHUNK);
  for (int c0 = cbeg; c0 < cend; c0 += CHUNK) {
#pragma unroll
    for (int q = 0; q < WD; ++q) wload(q, c0 / CHUNK, q);
    // halo row -> voxel index (or -1 outside the volume = zero padding)
    for (int row = tid; row < FHROWS; row += 256) {
      const int hd = row / (FHH * FHW), rem = row - hd * (FHH * FHW);
      const int hh = rem / FHW, hw = rem - hh * FHW;
      const int d = d0 - 1 + hd, h = h0 - 1 + hh, w = w0 - 1 + hw;
      rowoff[row] = (d >= 0 && d < g.D && h >= 0 && h < g.H && w >= 0 && w < g.W) ? (((b * g.D + d) * g.H + h) * g.W + w) : -1;
    }
    __syncthreads();
    // halo: items ordered so that 8 consecutive lanes write 8 consecutive rows of one plane (conflict-free ds_write_b128)
    // while 48 consecutive lanes read 8 voxels x 96 contiguous bytes from HBM; all 16 loads of a lane are issued before
    // the first LDS write (row -> global offset comes from the table built once per workgroup)
    {
      constexpr int NIT = (FHROWS * GPT + 255) / 256;      // 16 (6 groups), 11, 6
      constexpr int IPB = 8 * GPT;                         // items of a block of 8 rows
      VT hv[NIT];
      int ro[NIT];
#pragma unroll
      for (int i = 0; i < NIT; ++i) {                      // the 16 table reads first (in-kernel timestamps: with the read inside the
        const int idx = tid + 256 * i;                     // load loop every load waited for its own LDS round trip, 4300 cycles)
        const int blk = idx / IPB, j = idx - blk * IPB;
        ro[i] = idx < FHROWS * GPT ? rowoff[blk * 8 + (j & 7)] : -1;
      }
      if (vec_x && c0 + CHUNK <= Cin) {                    // whole chunk present and 16-byte aligned: branch-free vector loads
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int idx = tid + 256 * i;
          const int j = idx % IPB;
          VT v;
#pragma unroll
          for (int e = 0; e < KPC; ++e) v[e] = from_f32<T>(0.f);
          if (ro[i] >= 0) v = *reinterpret_cast<const VT*>(x + (int64_t)ro[i] * ldx + c0 + (j >> 3) * KPC);
          hv[i] = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int idx = tid + 256 * i;
          const int j = idx % IPB;
          VT v;
#pragma unroll
          for (int e = 0; e < KPC; ++e) v[e] = from_f32<T>(0.f);
          if (ro[i] >= 0) {
            const int c = c0 + (j >> 3) * KPC;
            const T* p = x + (int64_t)ro[i] * ldx + c;
            if (vec_x && c + KPC <= Cin) v = *reinterpret_cast<const VT*>(p);
            else {
#pragma unroll
              for (int e = 0; e < KPC; ++e)
                if (c + e < Cin) v[e] = p[e];
            }
          }
          hv[i] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int idx = tid + 256 * i;
        const int blk = idx / IPB, j = idx - blk * IPB;
        if (idx < FHROWS * GPT) *reinterpret_cast<VT*>(lh + ((j >> 3) * FPS + blk * 8 + (j & 7)) * 16) = hv[i];
      }
    }
    wstore(0, 0);
    __syncthreads();
    auto run_phase = [&](int phase, int slot_load, int slot_store) {      // `phase` is a compile-time constant after unrolling
      if (phase + WD < NPH) wload(phase + WD, c0 / CHUNK, slot_load);
      const char* abase;
      if (GPT == 4 || (GPT == 6 && phase < 9)) {
        abase = lh + laneA + (((phase / 3) * FHH) * FHW + phase % 3) * 16;
      } else {
        const int pb = GPT == 6 ? phase - 9 : phase;
        const int k0 = 2 * pb, k1 = k0 + 1 > 8 ? 8 : k0 + 1;       // (kdw == 9: the dummy slots read a valid row against zero weights)
        const int d0 = (((k0 / 3) * FHH) * FHW + k0 % 3) * 16, d1 = (((k1 / 3) * FHH) * FHW + k1 % 3) * 16;
        abase = lh + laneB + (hiB ? d1 : d0);
      }
      const char* wb = lw + ((phase & 1) * WITEMS + fq * NROWS) * 16 + wfrag;
      VT af[6], bfr[2][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bfr[0][nt] = *reinterpret_cast<const VT*>(wb + nt * 256);
#pragma unroll
      for (int hh = 0; hh < 6; ++hh) af[hh] = *reinterpret_cast<const VT*>(abase + hh * FHW * 16);
      // fp32 parity mode: a phase's 48 products per output element are summed in an accumulator of their own and join the running sum
      // once - two-level (blocked) summation.  One fp32 accumulator walked all K = 27 Cin terms (1296 .. 20736) and its rounding random
      // walk (~0.3 ulp sqrt(K)) was THE forward error of the parity mode: C2 logits 1.30e-6 from the float64 run, 3.5e-7 with this
      // convolution alone evaluated in float64 (scripts/debug/f32_error_sources.py); the reference's own fp32 run sits at 7.6e-7.
      // The bf16 mode keeps one accumulator (its error is the operands' rounding, 2^-9).
      constexpr bool BLOCKED = std::is_same<T, float>::value;
      if constexpr (BLOCKED) {
        // output row by output row: its three kh steps (48 products per element) go to a 4 x NT-register block accumulator, then one add
        VT bf3[3][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf3[0][nt] = bfr[0][nt];
#pragma unroll
        for (int kh = 1; kh < 3; ++kh)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bf3[kh][nt] = *reinterpret_cast<const VT*>(wb + (kh * 4 * NROWS) * 16 + nt * 256);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          f32x4 pacc[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) pacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) MmaC<T>::run(pacc[nt], bf3[kh][nt], af[mt + kh]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] += pacc[nt];
            // the sum is pinned HERE: left alone, the optimiser sank all adds behind the last phase (the running sum has no user before
            // the epilogue), every phase's block accumulators stayed live and the kernel spilled 2.3 KB per lane
            asm volatile("" : "+v"(acc[mt][nt]));
          }
        }
      } else {
        // (s_setprio 1 / 3 around this block, so that the other workgroup's staging does not delay the MFMA issue: 96^3 48->48 122.8 / 122.5 /
        // 125.9 us - nothing; scripts/debug/build_conv_exp.sh builds such variants beside the library)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          if (kh + 1 < 3) {   // next step's weight fragments in flight during the MFMAs
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[(kh + 1) & 1][nt] = *reinterpret_cast<const VT*>(wb + ((kh + 1) * 4 * NROWS) * 16 + nt * 256);
          }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) MmaC<T>::run(acc[mt][nt], bfr[kh & 1][nt], af[mt + kh]);
        }
      }
      if (phase + 1 < NPH) {
        wstore((phase + 1) & 1, slot_store);     // buffer last read in phase - 1: every wave is past that phase's barrier
        __syncthreads();
      }
    };
#pragma unroll
    for (int phase = 0; phase < NPH; ++phase) run_phase(phase, phase % WD, (phase + 1) % WD);   // phase, ring slots: compile-time
    __syncthreads();                 // halo + weight buffers are free for the next chunk
  }
  // Round 5 - the 1x1x1 shortcut of a residual block inside the data-gradient launch of its first convolution (dynunet_block.py:100-126:
  // out = conv1(x) ..., residual = conv3(x); backward dx = dgrad3x3(g1) + g3 W3).  The second term is one more tap: g3's 48-channel chunks at
  // the CENTRE of the halo, two short MFMA steps per chunk (k groups 0..3, then 4..5) against W3's transposed bf16 copy [Cout][Csc] read
  // as it is.  Until now a streaming GEMM wrote g3 W3 (170 MB at 96^3, 75 us) and this kernel read it back through its residual
  // epilogue (+60 us).  bf16, 96-byte chunks, unsplit launches; EPI instantiations only.
  // acc += src(centre tap) * wsrc^T: src rows over the same voxels, Csrc channels (a multiple of the chunk), wsrc [Cout][Csrc]
  auto centre_term = [&](const T* __restrict__ src, int64_t ldsrc, const T* __restrict__ wsrc, int Csrc) {
    if constexpr (EPI && GPT == 6 && std::is_same<T, bf16>::value) {
      for (int c0 = 0; c0 < Csrc; c0 += CHUNK) {
        // the brick's OWN 256 voxels only (the centre tap reads nothing else of the halo image): 6 items of 16 bytes per thread, addressed
        // arithmetically - 8 consecutive lanes take 8 consecutive voxels of one w-row and one channel group (conflict-free LDS writes,
        // 48 lanes read 768 contiguous bytes)
        {
          constexpr int NIT = (FBD * FBH * FBW * GPT) / 256;      // 6
          constexpr int IPB = 8 * GPT;
          VT hv[NIT];
#pragma unroll
          for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 256 * i;
            const int blk = idx / IPB, j = idx - blk * IPB;
            const int v = blk * 8 + (j & 7);                       // voxel of the brick: (bd, bh, bw) = (v / 64, (v / 16) % 4, v % 16)
            const int d = d0 + (v >> 6), h = h0 + ((v >> 4) & 3), w = w0 + (v & 15);
            VT val;
#pragma unroll
            for (int e = 0; e < KPC; ++e) val[e] = from_f32<T>(0.f);
            if (d < g.D && h < g.H && w < g.W) val = *reinterpret_cast<const VT*>(src + (int64_t)(((b * g.D + d) * g.H + h) * g.W + w) * ldsrc + c0 + (j >> 3) * KPC);
            hv[i] = val;
          }
#pragma unroll
          for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 256 * i;
            const int blk = idx / IPB, j = idx - blk * IPB;
            const int v = blk * 8 + (j & 7);
            const int hrow = (((v >> 6) + 1) * FHH + ((v >> 4) & 3) + 1) * FHW + (v & 15) + 1;
            *reinterpret_cast<VT*>(lh + ((j >> 3) * FPS + hrow) * 16) = hv[i];
          }
        }
        // weight images [4 k-slots][NROWS][16 B]: image 0 = k groups 0..3 of this chunk, image 1 = groups 4, 5 and two slots of zeros
        for (int idx = tid; idx < 2 * 4 * NROWS; idx += 256) {
          const int img = idx / (4 * NROWS), it = idx - img * (4 * NROWS);
          const int slot = it / NROWS, row = it - slot * NROWS;
          VT v;
#pragma unroll
          for (int e = 0; e < KPC; ++e) v[e] = from_f32<T>(0.f);
          const int grp = img * 4 + slot;
          if (grp < GPT && n0 + row < Cout) v = *reinterpret_cast<const VT*>(wsrc + (int64_t)(n0 + row) * Csrc + c0 + grp * KPC);
          *reinterpret_cast<VT*>(lw + (img * WITEMS + it) * 16) = v;
        }
        __syncthreads();
        {
          constexpr int CENTRE = ((1 * FHH) * FHW + 1) * 16;       // tap (kd, kw) = (1, 1); kh = 1 -> halo row mt + 1
          const char* a0 = lh + laneA + CENTRE;
          const char* a1 = lh + laneB + CENTRE;
          VT af0[4], af1[4], b0[NT], b1[NT];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            af0[mt] = *reinterpret_cast<const VT*>(a0 + (mt + 1) * FHW * 16);
            af1[mt] = *reinterpret_cast<const VT*>(a1 + (mt + 1) * FHW * 16);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            b0[nt] = *reinterpret_cast<const VT*>(lw + (fq * NROWS) * 16 + wfrag + nt * 256);
            b1[nt] = *reinterpret_cast<const VT*>(lw + (WITEMS + fq * NROWS) * 16 + wfrag + nt * 256);
          }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              MmaC<T>::run(acc[mt][nt], b0[nt], af0[mt]);
              MmaC<T>::run(acc[mt][nt], b1[nt], af1[mt]);
            }
        }
        __syncthreads();              // the images are free for the next shortcut chunk / the statistics reduction of the epilogue
      }
    }
  };
  if constexpr (EPI) {
    if (scx) centre_term(scx, ldscx, scw, Csc);      // (kernel argument: uniform)
  }
  // epilogue: lane holds channels n0 + 16nt + 4fq .. +3 of voxel (d0 + wave, h0 + mt, w0 + fi)
  auto epilogue = [&](T* __restrict__ y, int64_t ldy, bool vec_y, const T* __restrict__ res, int64_t ldres, double* __restrict__ stat, T* __restrict__ s2c_out) {
  const int d = d0 + wave, w = w0 + fi;
  const bool vec_r = res && (reinterpret_cast<uintptr_t>(res) & 15) == 0 && ldres % KPC == 0;
  float ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }
  const int64_t nvox_total = (int64_t)g.B * g.D * g.H * g.W;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int h = h0 + mt;
    if (d < g.D && h < g.H && w < g.W) {
      const int64_t vox = (((int64_t)b * g.D + d) * g.H + h) * g.W + w;
      if (scratch) {   // split over channel chunks: one fp32 partial slab per split (plain stores), summed + converted by a second kernel
        float* sl = scratch + ((int64_t)blockIdx.z * nvox_total + vox) * Cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int co = n0 + nt * 16 + fq * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < Cout) sl[co + r] = acc[mt][nt][r];
        }
        continue;
      }
      T* yr = y + vox * ldy;
      if (EPI && s2c_out && n0 < s2c_C) {
        // Round 5: the first s2c_C output channels leave in space-to-channel order - row (coarse voxel (d/2, h/2, w/2), j = 4 (d&1) + 2 (h&1) + (w&1))
        // of a [.., 8][s2c_C] tensor: what the transposed convolution in front of a decoder block reads as the gradient of its output
        // (unetr_block.py:80-85).  Until now those channels were written with the others and a gather pass re-laid them out.
        const int64_t cv = ((((int64_t)b * (g.D >> 1) + (d >> 1)) * (g.H >> 1) + (h >> 1)) * (g.W >> 1) + (w >> 1));
        yr = s2c_out + (cv * 8 + (((d & 1) << 2) | ((h & 1) << 1) | (w & 1))) * s2c_C;
      }
      const T* rr = (EPI && res) ? res + vox * ldres : nullptr;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + nt * 16 + fq * 4;
        f32x4 v = acc[mt][nt];
        if (EPI && rr) {
          if (vec_r && co + 3 < Cout) {
            if constexpr (std::is_same<T, bf16>::value) {
              const bf16x4 q = *reinterpret_cast<const bf16x4*>(rr + co);
              v += f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
            } else {
              v += *reinterpret_cast<const f32x4*>(rr + co);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (co + r < Cout) v[r] += to_f32(rr[co + r]);
          }
        }
        T o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = from_f32<T>(v[r]);
          if (EPI && co + r < Cout) {               // statistics of the stored (rounded) values
            const float q = to_f32(o[r]);
            ssum[nt][r] += q;
            ssq[nt][r] += q * q;
          }
        }
        if (vec_y && co + 3 < Cout) {
          if constexpr (std::is_same<T, bf16>::value) {
            *reinterpret_cast<bf16x4*>(yr + co) = bf16x4{o[0], o[1], o[2], o[3]};
          } else {
            *reinterpret_cast<f32x4*>(yr + co) = f32x4{o[0], o[1], o[2], o[3]};
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < Cout) yr[co + r] = o[r];
        }
      }
    }
  }
  if (EPI && stat) {   // kernel argument: uniform
    // every lane's 8 NT partial sums go to LDS ([k][column][wave * 16 + fi], 65-float rows: conflict-free both ways), then one thread
    // per (column, k) adds the 64 of them: 24 writes + 64 reads instead of ~190 cross-lane shuffles per lane
    float* red = reinterpret_cast<float*>(lds);      // the staging images are dead (last barrier of the K loop)
    constexpr int RS = 65;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = nt * 16 + fq * 4 + r;
        red[(0 * NROWS + col) * RS + wave * 16 + fi] = ssum[nt][r];
        red[(1 * NROWS + col) * RS + wave * 16 + fi] = ssq[nt][r];
      }
    __syncthreads();
    if (tid < NROWS * 2) {
      const int k = tid / NROWS, col = tid - k * NROWS;
      const float* rp = red + (k * NROWS + col) * RS;
      float tot = 0.f;
#pragma unroll 16
      for (int i = 0; i < 64; ++i) tot += rp[i];
      if (n0 + col < Cout) atomicAdd(stat + (((int64_t)(blockIdx.x & 15) * g.B + b) * Cout + n0 + col) * 2 + k, (double)tot);
    }
  }
  };
  epilogue(y, ldy, vec_y, res, ldres, stat, s2c_out);
  // Round 5 - forward mirror of the shortcut term: the block's 1x1x1 shortcut convolution of the SAME input as a second output of this launch
  // (dynunet_block.py:87-97,118-122: residual = conv3(inp) beside out = conv1(inp)): the accumulators start again from zero, the centre tap of
  // the input's chunks against W3 [Cout][Cin], a second epilogue (own output, own statistics).  Until now a streaming GEMM read the input
  // again (170 MB at 96^3).
  if constexpr (EPI && GPT == 6 && std::is_same<T, bf16>::value) {
    if (fsw) {
      __syncthreads();                // the statistics reduction of the first epilogue is through with the LDS
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      centre_term(x, ldx, fsw, Cin);
      epilogue(fsy, ldfsy, (reinterpret_cast<uintptr_t>(fsy) & 15) == 0 && ldfsy % KPC == 0, nullptr, 0, fsstat, nullptr);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Round 5 - forward / data gradient on TINY volumes (3^3, 6^3: encoder10 / decoder5 of the headline net, 384 - 768 channels): weight streaming.
// The brick kernel above spends a 256-voxel brick on 27 or 216 voxels (one or two bricks per layer), walks 14 barrier-separated phases per
// chunk and keeps 27 KB of weights in flight per workgroup: 13 - 35 us per launch for 8 - 32 MB of weights.  Here a WAVE owns 16 output
// channels of one 48-channel chunk: its B operands are 16-byte loads straight from the phase-ordered pack (every byte read once, a whole
// phase - three k-steps - requested while the previous one feeds the matrix pipe, no LDS, no barrier in the loop), the A operands come from
// a zero-bordered copy of the chunk's volume in LDS (row stride 112 bytes: conflict-free for the 16 rows of a fragment), the result is one
// fp32 slab per chunk - the split-K slabs the brick kernel writes for these layers, summed by the same second launch / the consumer norm.
// ---------------------------------------------------------------------------------------------------------
static constexpr int TINY_ROWB = 112;
template <int MT /* 16-voxel M tiles per wave */, int MG /* waves that share an output tile and split the voxels: 1 (<= 32 voxels, MT = 2) or 4 (<= 256, MT = 4) */>
__global__ void __launch_bounds__(256) conv3_fwd_tiny_kernel(const bf16* __restrict__ x, int64_t ldx, const bf16* __restrict__ wpk, int D, int H, int W, int Cin,
                                                             int Cout, int CoP, float* __restrict__ scratch, int nvox_total) {
  constexpr int NPH = 14, KPC = 8;
  extern __shared__ __attribute__((aligned(16))) char lds[];      // [(D + 2)(H + 2)(W + 2) rows][112 B], zero border
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, fq = lane >> 4;
  const int chunk = blockIdx.y, b = blockIdx.z, nvox = D * H * W;
  const int HP = H + 2, WP = W + 2, nrows = (D + 2) * HP * WP;
  for (int i = tid; i < nrows * (TINY_ROWB / 16); i += 256) *reinterpret_cast<f32x4*>(lds + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int i = tid; i < nvox * 6; i += 256) {
    const int v = i / 6, cg = i - v * 6;
    const int d = v / (H * W), r = v - d * (H * W), h = r / W, w = r - h * W;
    const int c = chunk * 48 + cg * KPC;
    bf16x8 val;
#pragma unroll
    for (int e = 0; e < 8; ++e) val[e] = (bf16)0.f;
    if (c + KPC <= Cin) val = *reinterpret_cast<const bf16x8*>(x + ((int64_t)b * nvox + v) * ldx + c);
    *reinterpret_cast<bf16x8*>(lds + (((d + 1) * HP + h + 1) * WP + w + 1) * TINY_ROWB + cg * 16) = val;
  }
  __syncthreads();
  const int ct = blockIdx.x * (4 / MG) + wave / MG, mg = wave % MG;      // 16-channel output tile of this wave, its share of the voxels
  if (ct * 16 >= CoP) return;                // (no barrier behind this point)
  // halo row (tap (0, 0, 0)) of this lane's voxel in every M tile; voxels beyond the volume read the zero corner row 0
  int rbase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int v = (mg * MT + mt) * 16 + fi;
    const int d = v / (H * W), r = v - d * (H * W), h = r / W, w = r - h * W;
    rbase[mt] = v < nvox ? ((d * HP + h) * WP + w) * TINY_ROWB : -1;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16* wph = wpk + ((int64_t)chunk * NPH * FWD96_GROUPS * CoP + (int64_t)ct * 16 + fi) * KPC;      // + (phase * 12 + kh * 4 + fq) * CoP * KPC
  bf16x8 wf[2][3];
  auto wload = [&](int phase, int buf) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) wf[buf][kh] = *reinterpret_cast<const bf16x8*>(wph + (int64_t)((phase * FWD96_GROUPS + kh * 4 + fq) * CoP) * KPC);
  };
  wload(0, 0);
#pragma unroll
  for (int phase = 0; phase < NPH; ++phase) {
    if (phase + 1 < NPH) wload(phase + 1, (phase + 1) & 1);
    // this lane's (kd, kw, channel group) of the phase (the table at fwd96_phases): slot fq
    int kdw, cg;
    if (phase < 9) { kdw = phase; cg = fq; }
    else { kdw = 2 * (phase - 9) + (fq >> 1); cg = 4 + (fq & 1); if (kdw > 8) kdw = 8; }      // (kdw == 9: the dummy slots - zero weights, any row)
    const int koff = ((kdw / 3) * HP * WP + (kdw % 3)) * TINY_ROWB + cg * 16;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int off = koff + kh * WP * TINY_ROWB;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(lds + (rbase[mt] >= 0 ? rbase[mt] + off : 0));
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[phase & 1][kh], af, acc[mt], 0, 0, 0);
      }
    }
  }
  // lane: channels ct * 16 + 4 fq .. + 3 of voxel mt * 16 + fi -> the chunk's slab [split = chunk][voxel][Cout]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int v = (mg * MT + mt) * 16 + fi, co = ct * 16 + fq * 4;
    if (v < nvox) {
      float* sl = scratch + ((int64_t)chunk * nvox_total + (int64_t)b * nvox + v) * Cout + co;
      if (co + 3 < Cout) *reinterpret_cast<f32x4*>(sl) = acc[mt];
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < Cout) sl[r] = acc[mt][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight packing.  Row-major packs (generic kernel):   fwd[co][tap][ciP] = w[co][ci][tap] ; bwd[ci][tap][coP] = w[co][ci][26-tap]
// Planar packs (fast path, chosen when the K-side channel row is a multiple of 96 bytes):
//   fwd[tap][ci / KPC][coP16][ci % KPC] = w[co][ci][tap]        bwd[tap][co / KPC][ciP16][co % KPC] = w[co][ci][26 - tap]
// ---------------------------------------------------------------------------------------------------------
// One workgroup packs a 16 (co) x 16 (ci) x 27 tile: the torch-layout weight is read in contiguous runs of 16*27 floats
// into LDS, every pack is written as 16-byte vectors whose fastest index follows the pack's memory order.
static constexpr int PK_T = 16;
static constexpr int PK_LD = PK_T * 27 + 1;      // floats per co row of the LDS tile

// K extent (elements, a whole number of 96-byte chunks) the fast path walks for C channels on its K side, or 0 where the generic row-major
// kernel serves them.  Rows that are not a multiple of 96 bytes but at least `pad_min_bytes` wide are PADDED to the next chunk (zero weights
// in the pack; the staging reads nothing beyond C): C-UNETR's 128 / 256-channel layers at 12^3 - 24^3 get the split-K plan of the fast path
// instead of 12 - 54 workgroups of the generic kernel (85 - 160 us -> 20 - 30 us each), its 32 / 64-channel layers the planar LDS images and
// the fused residual / statistics epilogue although a third to a half of their MFMA work multiplies zeros (C-UNETR patches/s with the
// threshold at 0 / 256 / 128 / 64 / 32 bytes: 146.2 / 155.2 / 161.0 / 165.7 / 163.4 - 16 channels padded to 48 no longer pay).  ONE predicate for the pack, the launch, the
// workspace size and the host side (miseg_conv3_k96).
// Round 4: rows of 4 k groups (64 bytes: 32, 64, 128, 256 bf16 channels) and of 2 groups (16 bf16 channels) are walked in chunks of their own
// width (fwd96_phases: 9 / 5 phases per chunk, nothing padded); `pad_min_bytes` carries the switch in bit 20 (CONV3_NARROW; MISEG_CONV3_NARROW=0
// restores the padding).  conv3_gpt: groups per chunk of the K extent conv3_k96 returns (6 when padded).
static constexpr int CONV3_NARROW = 1 << 20;
__host__ __device__ inline int conv3_gpt(int C, int esz, int plan) {
  const int kpc = 16 / esz, g = (C + kpc - 1) / kpc;
  // fp32 is the PARITY mode: it keeps ONE summation plan, the 96-byte chunks (rows padded to whole chunks) of rounds 2 - 3.  The narrow
  // chunks add the same fp32 products in another order - 1.4e-7 per output either way, but through the sign of ONE PReLU / LeakyReLU
  // pre-activation near zero that moved the plain UNet's gradient medians from 5.7e-7 to 1.2e-3 (round 4, profiles/r04_vs_truth.txt)
  if (esz == 4) return 6;
  if (g % 6 == 0 || !(plan & CONV3_NARROW)) return 6;
  if (g % 4 == 0) return 4;
  if (g == 2) return 2;
  return 6;
}
__host__ __device__ inline int conv3_k96(int C, int esz, int plan) {
  const int kpc = 16 / esz, cp = (C + kpc - 1) / kpc * kpc, per = 96 / esz, pad_min_bytes = plan & (CONV3_NARROW - 1);
  if ((cp * esz) % 96 == 0) return cp;
  if (conv3_gpt(C, esz, plan) != 6) return cp;
  if (pad_min_bytes > 0 && cp * esz >= pad_min_bytes) return (C + per - 1) / per * per;
  return 0;
}

static int conv3_pad_min_bytes() {      // rows of >= 64 bytes are padded to whole 96-byte chunks where no narrow chunk fits; narrow chunks on (bf16)
  return 64 | CONV3_NARROW;             // (rounds 3 - 4 swept both through the environment; the library reads no environment any more)
}

// element offset of the 16-byte group (tap, K-side channel group kg, N-side row) in the phase-ordered pack of the fast path
// [chunk = kg / gpt][phase][kh * 4 + slot][N16][KPC]
__device__ __forceinline__ int64_t fwd96_pack_offset(int gpt, int tap, int kg, int row, int N16, int KPC) {
  const int chunk = kg / gpt, cg = kg - chunk * gpt;
  const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3, kdw = kd * 3 + kw;
  const int phase = fwd96_phase_of(gpt, kdw, cg), slot = fwd96_slot_of(gpt, kdw, cg);
  return ((((int64_t)chunk * fwd96_phases(gpt) + phase) * FWD96_GROUPS + kh * 4 + slot) * N16 + row) * KPC;
}

// the two k-slots of the last phase that no (tap, group) maps to (6 and 2 groups per chunk) must read as zero: written by the tile that opens
// a chunk, for its 16 rows
template <class T>
__device__ __forceinline__ void fwd96_pack_dummies(int gpt, T* pack, int chunk, int row0, int N16) {
  typedef typename Vec16<T>::type VT;
  constexpr int KPC = Vec16<T>::N;
  if (gpt == 4) return;
  const int nph = fwd96_phases(gpt);
  VT z;
#pragma unroll
  for (int e = 0; e < KPC; ++e) z[e] = from_f32<T>(0.f);
  for (int i = threadIdx.x; i < 3 * 2 * PK_T; i += 256) {
    const int r = i % PK_T, sl = 2 + (i / PK_T) % 2, kh = i / (2 * PK_T);
    if (row0 + r < N16)
      *reinterpret_cast<VT*>(pack + ((((int64_t)chunk * nph + nph - 1) * FWD96_GROUPS + kh * 4 + sl) * N16 + row0 + r) * KPC) = z;
  }
}

// OPT (round 5, miseg_opt_step_pack_conv3): the tile is first UPDATED - one optimiser step on its weights, read with their gradient and
// state slots at the same element offsets of the flat arena - and the packs are written from the new values: the separate refresh pass of
// the next step (a second read of the 55 M conv weights, 440 MB moved) is gone.
struct OptTile { const float* g; float* m; float* v; OptHyper h; };      // g / m / v: already offset to this parameter's slot
template <class T, bool OPT = false>
__device__ __forceinline__ void pack_conv3_tile(typename std::conditional<OPT, float, const float>::type* __restrict__ w, T* __restrict__ fwd, T* __restrict__ bwd, int Cin,
                                                int Cout, int CinP, int CoutP, int Cin16, int Cout16, int fwd_gpt, int bwd_gpt, int bx, int by, const OptTile* ot = nullptr) {
  const bool fwd_planar = fwd_gpt != 0, bwd_planar = bwd_gpt != 0;      // groups per chunk of the planar pack's K side, 0 = row-major pack
  typedef typename Vec16<T>::type VT;
  constexpr int KPC = Vec16<T>::N, NG = PK_T / KPC;
  __shared__ float tile[PK_T * PK_LD];
  const int ci0 = bx * PK_T, co0 = by * PK_T;
  const int nci = min(PK_T, Cin - ci0);
  if (nci == PK_T && (Cin & 3) == 0) {
    // whole tile, 16-byte aligned rows (432 floats per out-channel): seven float4 loads per thread, ALL in flight before the first LDS write.
    // Round 4: the scalar loop below issued its 27 loads a few at a time - 27 memory round trips per tile, 265 us for the 55 M weights of
    // C-Swin-UNETR (1.7 TB/s of its 440 MB); what a live weight refresh costs every optimisation step
    constexpr int R4 = PK_T * 27 / 4, NV = (PK_T * R4 + 255) / 256;      // 108 float4 per row, 7 per thread
    // Every load is unconditional straight-line code at a clamped (valid, aligned) offset and the selects come after: behind a per-lane guard
    // the loads of the optimiser's three other operands went out one k at a time - seven dependent memory round trips per tile with 12 KB
    // in flight per workgroup (round 5: 379 us for the 1.6 GB the step moves, 4.2 TB/s)
    f32x4 v[NV];
    int64_t off[NV];
    bool ok[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = threadIdx.x + k * 256, co = idx / R4, r4 = idx - co * R4;
      ok[k] = idx < PK_T * R4 && co0 + co < Cout;
      const int coc = min(co0 + min(co, PK_T - 1), Cout - 1);
      off[k] = ((int64_t)coc * Cin + ci0) * 27 + 4 * r4;
      v[k] = *reinterpret_cast<const f32x4*>(w + off[k]);
    }
    if constexpr (OPT) {
      const bool adam = ot->h.kind != MISEG_OPT_SGD_NESTEROV;
      f32x4 g4[NV], m4[NV], v4[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        g4[k] = *reinterpret_cast<const f32x4*>(ot->g + off[k]);
        m4[k] = *reinterpret_cast<const f32x4*>(ot->m + off[k]);
      }
      if (adam) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v4[k] = *reinterpret_cast<const f32x4*>(ot->v + off[k]);
      } else {
#pragma unroll
        for (int k = 0; k < NV; ++k) v4[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float we = v[k][e], me = m4[k][e], ve = v4[k][e];
          opt_update(ot->h, we, g4[k][e], me, ve);
          v[k][e] = we; m4[k][e] = me; v4[k][e] = ve;
        }
        if (ok[k]) {
          *reinterpret_cast<f32x4*>(w + off[k]) = v[k];
          *reinterpret_cast<f32x4*>(ot->m + off[k]) = m4[k];
          if (adam) *reinterpret_cast<f32x4*>(ot->v + off[k]) = v4[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (!ok[k]) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = threadIdx.x + k * 256, co = idx / R4, r4 = idx - co * R4;
      if (idx < PK_T * R4) {
        float* t = tile + co * PK_LD + 4 * r4;
        t[0] = v[k][0]; t[1] = v[k][1]; t[2] = v[k][2]; t[3] = v[k][3];
      }
    }
  } else {
#pragma unroll 1
    for (int i = threadIdx.x; i < PK_T * PK_T * 27; i += 256) {
      const int co = i / (PK_T * 27), r = i - co * (PK_T * 27);     // r = ci_local * 27 + tap
      float wv = 0.f;
      if (co0 + co < Cout && r < nci * 27) {
        const int64_t off = ((int64_t)(co0 + co) * Cin + ci0) * 27 + r;
        wv = w[off];
        if constexpr (OPT) {
          float me = ot->m[off], ve = ot->h.kind != MISEG_OPT_SGD_NESTEROV ? ot->v[off] : 0.f;
          opt_update(ot->h, wv, ot->g[off], me, ve);
          w[off] = wv;
          ot->m[off] = me;
          if (ot->h.kind != MISEG_OPT_SGD_NESTEROV) ot->v[off] = ve;
        }
      }
      tile[co * PK_LD + r] = wv;
    }
  }
  __syncthreads();
  if (fwd) {   // K side = ci: vector = KPC consecutive ci of (co, tap)
#pragma unroll 1
    for (int i = threadIdx.x; i < PK_T * 27 * NG; i += 256) {
      int co, tap, cg;
      if (fwd_planar) { co = i % PK_T; cg = (i / PK_T) % NG; tap = i / (PK_T * NG); }
      else { cg = i % NG; tap = (i / NG) % 27; co = i / (NG * 27); }
      const int cib = ci0 + cg * KPC;
      if (cib >= CinP || (fwd_planar ? co0 + co >= Cout16 : co0 + co >= Cout)) continue;
      VT v;
#pragma unroll
      for (int e = 0; e < KPC; ++e) v[e] = from_f32<T>(tile[co * PK_LD + (cg * KPC + e) * 27 + tap]);
      const int64_t off = fwd_planar ? fwd96_pack_offset(fwd_gpt, tap, cib / KPC, co0 + co, Cout16, KPC) : ((int64_t)(co0 + co) * 27 + tap) * CinP + cib;
      *reinterpret_cast<VT*>(fwd + off) = v;
    }
    if (fwd_planar)      // every chunk that starts inside this tile's K range
      for (int ch = (ci0 + fwd_gpt * KPC - 1) / (fwd_gpt * KPC); ch * fwd_gpt * KPC < ci0 + PK_T && ch * fwd_gpt * KPC < CinP; ++ch)
        fwd96_pack_dummies<T>(fwd_gpt, fwd, ch, co0, Cout16);
  }
  if (bwd) {   // K side = co, taps mirrored: vector = KPC consecutive co of (ci, tap)
#pragma unroll 1
    for (int i = threadIdx.x; i < PK_T * 27 * NG; i += 256) {
      int ci, tap, cg;
      if (bwd_planar) { ci = i % PK_T; cg = (i / PK_T) % NG; tap = i / (PK_T * NG); }
      else { cg = i % NG; tap = (i / NG) % 27; ci = i / (NG * 27); }
      const int cob = co0 + cg * KPC;
      if (cob >= CoutP || (bwd_planar ? ci0 + ci >= Cin16 : ci0 + ci >= Cin)) continue;
      VT v;
#pragma unroll
      for (int e = 0; e < KPC; ++e) v[e] = from_f32<T>(tile[(cg * KPC + e) * PK_LD + ci * 27 + (26 - tap)]);
      const int64_t off = bwd_planar ? fwd96_pack_offset(bwd_gpt, tap, cob / KPC, ci0 + ci, Cin16, KPC) : ((int64_t)(ci0 + ci) * 27 + tap) * CoutP + cob;
      *reinterpret_cast<VT*>(bwd + off) = v;
    }
    if (bwd_planar)
      for (int ch = (co0 + bwd_gpt * KPC - 1) / (bwd_gpt * KPC); ch * bwd_gpt * KPC < co0 + PK_T && ch * bwd_gpt * KPC < CoutP; ++ch)
        fwd96_pack_dummies<T>(bwd_gpt, bwd, ch, ci0, Cin16);
  }
}

template <class T>
__global__ void __launch_bounds__(256) pack_conv3_kernel(const float* __restrict__ w, T* __restrict__ fwd, T* __restrict__ bwd, int Cin, int Cout, int CinP, int CoutP,
                                                         int Cin16, int Cout16, int fwd_gpt, int bwd_gpt) {
  pack_conv3_tile<T>(w, fwd, bwd, Cin, Cout, CinP, CoutP, Cin16, Cout16, fwd_gpt, bwd_gpt, blockIdx.x, blockIdx.y);
}

// (round 4: four workgroups per CU - the kernel is a latency chain per tile (search, load, two LDS passes, stores) and 230 VGPRs left room for
// two; the descriptor search runs on an LDS copy of the tile offsets instead of six dependent global loads per tile)
template <class T>
__global__ void __launch_bounds__(256) pack_conv3_batch_kernel(const miseg_pack_conv3_desc* __restrict__ descs, int n, int total_tiles,
                                                                  const int64_t* __restrict__ params_version, int64_t* __restrict__ state, int pad_min) {
  constexpr int KPC = Vec16<T>::N;
  constexpr int NS = 128;
  __shared__ int s_tile0[NS];
  const int64_t pv = params_version ? *params_version : 0;
  if (params_version && state[0] == pv) return;      // versioned refresh (miseg_hip.h): the packs were made from the current parameters
  const bool in_lds = n <= NS;
  if (in_lds) {
    for (int i = threadIdx.x; i < n; i += 256) s_tile0[i] = descs[i].tile0;
    __syncthreads();
  }
  for (int tl = blockIdx.x; tl < total_tiles; tl += gridDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {   // last descriptor with tile0 <= tl
      const int mid = (lo + hi + 1) >> 1;
      if ((in_lds ? s_tile0[mid] : descs[mid].tile0) <= tl) lo = mid; else hi = mid - 1;
    }
    const miseg_pack_conv3_desc d = descs[lo];
    const int kf = conv3_k96(d.Cin, (int)sizeof(T), pad_min), kb = conv3_k96(d.Cout, (int)sizeof(T), pad_min);
    const int t = tl - d.tile0, tci = ((kf > d.Cin ? kf : d.Cin) + PK_T - 1) / PK_T;     // tile grid of a layer: pack_tiles()
    const int CinP = kf ? kf : (d.Cin + KPC - 1) / KPC * KPC, CoutP = kb ? kb : (d.Cout + KPC - 1) / KPC * KPC;
    __syncthreads();      // pack_conv3_tile's LDS tile of the previous iteration has been read
    pack_conv3_tile<T>(d.w, (T*)d.fwd_pack, (T*)d.bwd_pack, d.Cin, d.Cout, CinP, CoutP, (d.Cin + 15) / 16 * 16, (d.Cout + 15) / 16 * 16,
                       kf ? conv3_gpt(d.Cin, (int)sizeof(T), pad_min) : 0, kb ? conv3_gpt(d.Cout, (int)sizeof(T), pad_min) : 0, t % tci, t / tci);
  }
  refresh_done(params_version, state, pv);
}

// one optimiser step on every 3x3x3 weight of the table + its packs (miseg_opt_step_pack_conv3).  Same tile walk as the refresh kernel above;
// a tensor without a gradient this step (used[] == 0) is left alone - weights, state and packs.
template <class T>
__global__ void __launch_bounds__(256) opt_pack_conv3_batch_kernel(const miseg_pack_conv3_desc* __restrict__ descs, const miseg_opt_pack_map* __restrict__ map, int n,
                                                                      int total_tiles, int pad_min, int kind, const float* __restrict__ grad, float* __restrict__ s1,
                                                                      float* __restrict__ s2, const int32_t* __restrict__ used, const int32_t* __restrict__ steps, float lr,
                                                                      float b1, float b2, float eps, float wd, float mom, const float* __restrict__ lr_dev) {
  constexpr int KPC = Vec16<T>::N;
  constexpr int NS = 128;
  __shared__ int s_tile0[NS];
  __shared__ OptTile s_ot;
  const bool in_lds = n <= NS;
  if (in_lds) {
    for (int i = threadIdx.x; i < n; i += 256) s_tile0[i] = descs[i].tile0;
    __syncthreads();
  }
  if (lr_dev) lr = *lr_dev;
  for (int tl = blockIdx.x; tl < total_tiles; tl += gridDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {   // last descriptor with tile0 <= tl
      const int mid = (lo + hi + 1) >> 1;
      if ((in_lds ? s_tile0[mid] : descs[mid].tile0) <= tl) lo = mid; else hi = mid - 1;
    }
    const miseg_opt_pack_map mp = map[lo];
    if (used && !used[mp.param_index]) continue;      // (uniform per workgroup)
    const miseg_pack_conv3_desc d = descs[lo];
    const int kf = conv3_k96(d.Cin, (int)sizeof(T), pad_min), kb = conv3_k96(d.Cout, (int)sizeof(T), pad_min);
    const int t = tl - d.tile0, tci = ((kf > d.Cin ? kf : d.Cin) + PK_T - 1) / PK_T;
    const int CinP = kf ? kf : (d.Cin + KPC - 1) / KPC * KPC, CoutP = kb ? kb : (d.Cout + KPC - 1) / KPC * KPC;
    __syncthreads();      // the LDS tile (and s_ot) of the previous iteration has been read
    if (threadIdx.x == 0) s_ot = OptTile{grad + mp.off, s1 + mp.off, s2 ? s2 + mp.off : nullptr, opt_hyper(kind, steps[mp.param_index] + 1, lr, b1, b2, eps, wd, mom)};
    __syncthreads();
    pack_conv3_tile<T, true>(const_cast<float*>(d.w), (T*)d.fwd_pack, (T*)d.bwd_pack, d.Cin, d.Cout, CinP, CoutP, (d.Cin + 15) / 16 * 16, (d.Cout + 15) / 16 * 16,
                             kf ? conv3_gpt(d.Cin, (int)sizeof(T), pad_min) : 0, kb ? conv3_gpt(d.Cout, (int)sizeof(T), pad_min) : 0, t % tci, t / tci, &s_ot);
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------
static constexpr int WG_THREADS = 256;   // 4 waves, one per SIMD: up to 512 registers each (the 27x9 accumulator tiles live in AGPRs), which
                                         // leaves the VGPRs to hold the NEXT brick's global loads across the MFMA loop
static constexpr int WG_WAVES = WG_THREADS / 64;
static constexpr int WG_TPW = 7;    // taps per wave: wave, wave + 4, ... (27 taps over 4 waves: 7, 7, 7, 6)
static constexpr int WG_CB = 48;   // channel block (3 MFMA tiles) on both the out- and in-channel side

#ifdef MISEG_WGRAD_STAMPS
__device__ unsigned long long miseg_wg_stamps[8];   // debug build only (scripts/debug_wgrad_stamps.py): summed s_memrealtime ticks
#define WG_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define WG_STAMP_ADD(i, a, b) do { if (tid == 0) atomicAdd(&miseg_wg_stamps[i], (b) - (a)); } while (0)
#else
#define WG_STAMP(var)
#define WG_STAMP_ADD(i, a, b)
#endif

// One workgroup = one (48 out-channel, 48 in-channel) pair x one share `split` of the bricks, all 27 taps.
// direct = 0: the 27x48x48 partial goes to its slab (summed by the reduce kernel);
// direct = 1 (only with nsplit == 1: every dw element has exactly one producer): transposed through LDS and added to
//             (2: stored into) the torch-layout gradient in contiguous runs - no slab round trip, no second launch.
// PIPED (bf16, whole 48-channel blocks, 16-byte aligned rows, at least one whole brick along every axis - the last one may be partial,
// its rows beyond the volume are staged as zeros): the software-pipelined MFMA loop.  A separate instantiation, not a branch: with both loops in one
// body the 252 accumulator registers of the two paths met in PHIs and the kernel spilled 350 VGPRs.
template <class T, int WBD /*brick depth*/, bool PIPED>
__device__ __forceinline__ void conv3_wgrad_body(const T* __restrict__ x, int64_t ldx, const T* __restrict__ dy, int64_t lddy, float* __restrict__ slabs,
                                                 float* __restrict__ dw, const ConvGeom& g, int Cin, int Cout, int ncib, int nsplit, int rowb, bool vec_x,
                                                 bool vec_dy, int split, int pair, int direct, bool xcd_map) {
  typedef typename Vec16<T>::type VT;
  constexpr int KPC = MmaC<T>::KPC;
  constexpr int GPR = WG_CB / KPC;        // 16-byte groups per staged row
  constexpr int NVOX = WBD * BH * BW;     // voxels (k) per brick
  constexpr int HROWS = (WBD + 2) * HH * HW;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* lx = lds;                          // halo of x : [HROWS][rowb]
  char* ld = lds + HROWS * rowb;           // dy brick  : [NVOX][rowb]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the tap guards around the MFMAs are real branches
  const int cob = pair / ncib, cib = pair - cob * ncib;
  const int co0 = cob * WG_CB, ci0 = cib * WG_CB;
  const int fi = lane & 15, fq = lane >> 4;

  // taps owned by this wave: wave, wave + 4, ..., wave + 24
  f32x4 acc[WG_TPW][3][3];
#pragma unroll
  for (int t = 0; t < WG_TPW; ++t)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int toff[WG_TPW];
#pragma unroll
  for (int t = 0; t < WG_TPW; ++t) {
    const int tap = wave + WG_WAVES * t;
    toff[t] = tap < 27 ? (((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3) : 0;
  }

  const int nbricks = g.B * g.nbd * g.nbh * g.nbw;
  // staging.  Fast path (whole 48-channel blocks present, 16-byte aligned rows): the 21 16-byte items a lane stages per brick are
  // loaded for the NEXT brick right before the MFMA loop of the current one and written to LDS after it (in-kernel timestamps of
  // the un-prefetched version: 54 % of a workgroup's time was the wait for these loads, 44 % the MFMA loop).
  constexpr int NXI = (HROWS * GPR + WG_THREADS - 1) / WG_THREADS, NDI = (NVOX * GPR + WG_THREADS - 1) / WG_THREADS;
  const bool fast = PIPED || (vec_x && vec_dy && ci0 + WG_CB <= Cin && co0 + WG_CB <= Cout);
  VT rx[NXI], rd[NDI];
  // One staged item = 16 bytes of one halo row (j < NXI) or of one dy row.  Loads are unconditional (an item outside the volume
  // reads the tensor's first bytes instead and is zeroed through `vmask` when it is written to LDS): no branch per item, so
  // the items can sit between the MFMAs of the k-loop.
  uint32_t vmask = 0;
  int nb = 0, nd0 = 0, nh0 = 0, nw0 = 0;
  bool nok = false;
  int64_t nmask = 0;
  int tid_s = tid;      // made opaque once per brick: otherwise the row/column split of all 21 items is hoisted out of the brick loop
                        // as loop invariants (~100 live VGPRs, 470 spills)
  auto set_next = [&](int brick, bool ok) {
    asm volatile("" : "+v"(tid_s));
    int bid = brick;
    const int bw = bid % g.nbw; bid /= g.nbw;
    const int bh = bid % g.nbh; bid /= g.nbh;
    nd0 = (bid % g.nbd) * WBD; nb = bid / g.nbd; nh0 = bh * BH; nw0 = bw * BW;
    nok = ok;
    nmask = ok ? (int64_t)-1 : 0;     // no next brick: every item re-reads the tensor's first bytes
    vmask = 0;
  };
  auto gitem = [&](int j) {
    if (j < NXI) {
      const int idx = tid_s + j * WG_THREADS;
      const int row = idx / GPR, cg = idx - row * GPR;
      const int hd = row / (HH * HW), rem = row - hd * (HH * HW);
      const int hh = rem / HW, hw = rem - hh * HW;
      const int d = nd0 - 1 + hd, h = nh0 - 1 + hh, w = nw0 - 1 + hw;
      const bool ok = nok && idx < HROWS * GPR && d >= 0 && d < g.D && h >= 0 && h < g.H && w >= 0 && w < g.W;
      // clamped coordinates: always a valid address, no select / branch on the 64-bit offset
      const int dc = min(max(d, 0), g.D - 1), hc = min(max(h, 0), g.H - 1), wc = min(max(w, 0), g.W - 1);
      const int64_t off = (((((int64_t)nb * g.D + dc) * g.H + hc) * g.W + wc) * ldx + ci0 + cg * KPC) & nmask;
      rx[j] = *reinterpret_cast<const VT*>(x + off);
      vmask |= (ok ? 1u : 0u) << j;
    } else {
      const int i = j - NXI;
      const int idx = tid_s + i * WG_THREADS;
      const int row = idx / GPR, cg = idx - row * GPR;
      const int vd = row / (BH * BW), rem = row - vd * (BH * BW);
      const int vh = rem / BW, vw = rem - vh * BW;
      const int d = nd0 + vd, h = nh0 + vh, w = nw0 + vw;
      const bool ok = nok && idx < NVOX * GPR && d < g.D && h < g.H && w < g.W;
      const int dc = min(d, g.D - 1), hc = min(h, g.H - 1), wc = min(w, g.W - 1);
      const int64_t off = (((((int64_t)nb * g.D + dc) * g.H + hc) * g.W + wc) * lddy + co0 + cg * KPC) & nmask;
      rd[i] = *reinterpret_cast<const VT*>(dy + off);
      vmask |= (ok ? 1u : 0u) << j;
    }
  };
  // PIPED: what an item needs per brick is reduced to a handful of VALU operations (one wave per SIMD: index arithmetic between
  // the MFMAs is not hidden by anybody; the generic form above costs ~70 instructions per item, a quarter of them 64-bit
  // multiplies).  Per lane and item, constant over the bricks: a 32-bit byte offset from the brick's halo (dy: brick) origin
  // and 8 "face" bits - which faces of the halo the item sits on (bit 7: lane beyond the item count); per brick, scalar: the two
  // origins and which faces lie outside the volume.  An item on such a face reads the halo's first interior voxel instead and
  // is zeroed by vmask.
  // Round 5: volumes that are no whole number of bricks (12^3: H = W = 8 + 4).  The last brick along an axis is PARTIAL: three more item
  // bits - the item lies beyond the remainder r = extent % brick on that axis - against three more brick bits - this brick is the partial one
  // there (16 bits per item now).  Until then such layers walked the generic loop (17 us per brick against 8.5: the 12^3 layers were the
  // long workgroups of the grouped launch, 238 us for a group whose other workgroups need 130).
  uint32_t ioff[PIPED ? NXI + NDI : 1];
  uint32_t iface[PIPED ? (NXI + NDI + 1) / 2 : 1];
  const char* xorg = nullptr;
  const char* dorg = nullptr;
  uint32_t bfaces = 0xffff, ctr = 0;
  if constexpr (PIPED) {
    const int rd_ = g.D % WBD, rh_ = g.H % BH, rw_ = g.W % BW;
#pragma unroll
    for (int q = 0; q < (NXI + NDI + 1) / 2; ++q) iface[q] = 0;
#pragma unroll
    for (int j = 0; j < NXI + NDI; ++j) {
      uint32_t f;
      if (j < NXI) {
        const int idx = tid + j * WG_THREADS;
        const int row = idx / GPR, cg = idx - row * GPR;
        const int hd = row / (HH * HW), rem = row - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        ioff[j] = (uint32_t)((((int64_t)hd * g.H + hh) * g.W + hw) * ldx + cg * KPC) * (uint32_t)sizeof(T);
        f = (hd == 0 ? 1u : 0u) | (hd == WBD + 1 ? 2u : 0u) | (hh == 0 ? 4u : 0u) | (hh == HH - 1 ? 8u : 0u) | (hw == 0 ? 16u : 0u) | (hw == HW - 1 ? 32u : 0u) |
            (idx >= HROWS * GPR ? 128u : 0u) | (hd > rd_ ? 256u : 0u) | (hh > rh_ ? 512u : 0u) | (hw > rw_ ? 1024u : 0u);
      } else {
        const int idx = tid + (j - NXI) * WG_THREADS;
        const int row = idx / GPR, cg = idx - row * GPR;
        const int vd = row / (BH * BW), rem = row - vd * (BH * BW);
        const int vh = rem / BW, vw = rem - vh * BW;
        ioff[j] = (uint32_t)((((int64_t)vd * g.H + vh) * g.W + vw) * lddy + cg * KPC) * (uint32_t)sizeof(T);
        f = (idx >= NVOX * GPR ? 128u : 0u) | (vd >= rd_ ? 256u : 0u) | (vh >= rh_ ? 512u : 0u) | (vw >= rw_ ? 1024u : 0u);
      }
      iface[j >> 1] |= f << ((j & 1) * 16);
    }
    ctr = (uint32_t)((((int64_t)g.H + 1) * g.W + 1) * ldx) * (uint32_t)sizeof(T);
  }
  auto pset_next = [&](int brick, bool ok) {
    int bid = brick;
    const int bw = bid % g.nbw; bid /= g.nbw;
    const int bh = bid % g.nbh; bid /= g.nbh;
    const int d0 = (bid % g.nbd) * WBD, b = bid / g.nbd, h0 = bh * BH, w0 = bw * BW;
    xorg = reinterpret_cast<const char*>(x + ((((int64_t)b * g.D + d0 - 1) * g.H + h0 - 1) * g.W + w0 - 1) * ldx + ci0);
    dorg = reinterpret_cast<const char*>(dy + ((((int64_t)b * g.D + d0) * g.H + h0) * g.W + w0) * lddy + co0);
    bfaces = ok ? (128u | (d0 == 0 ? 1u : 0u) | (d0 + WBD >= g.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) | (h0 + BH >= g.H ? 8u : 0u) | (w0 == 0 ? 16u : 0u) |
                   (w0 + BW >= g.W ? 32u : 0u) | (d0 + WBD > g.D ? 256u : 0u) | (h0 + BH > g.H ? 512u : 0u) | (w0 + BW > g.W ? 1024u : 0u))
                : 0xffffu;
    vmask = 0;
  };
  auto pitem = [&](int j) {
    const bool ok = ((iface[j >> 1] >> ((j & 1) * 16)) & bfaces) == 0;
    if (j < NXI) rx[j] = *reinterpret_cast<const VT*>(xorg + (ok ? ioff[j] : ctr));
    else rd[j - NXI] = *reinterpret_cast<const VT*>(dorg + (ok ? ioff[j] : 0u));
    vmask |= (ok ? 1u : 0u) << j;
  };
  auto gload = [&]() {
#pragma unroll
    for (int j = 0; j < NXI + NDI; ++j) {
      if constexpr (PIPED) pitem(j); else gitem(j);
    }
  };
  auto lstore = [&]() {
    VT zero;
#pragma unroll
    for (int e = 0; e < KPC; ++e) zero[e] = from_f32<T>(0.f);
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int idx = tid + i * WG_THREADS;
      if (idx < HROWS * GPR) *reinterpret_cast<VT*>(lx + (idx / GPR) * rowb + (idx % GPR) * 16) = (vmask >> i) & 1 ? rx[i] : zero;
    }
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
      const int idx = tid + i * WG_THREADS;
      if (idx < NVOX * GPR) *reinterpret_cast<VT*>(ld + (idx / GPR) * rowb + (idx % GPR) * 16) = (vmask >> (NXI + i)) & 1 ? rd[i] : zero;
    }
  };
  static_assert(NXI + NDI <= 32, "validity mask is one 32-bit word");
  // which bricks this workgroup walks.  Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2: with
  // xcd_map (nsplit % 8 == 0, first workgroup of the layer on XCD 0) XCD k takes the k-th contiguous eighth of the bricks, so the
  // halo rows shared by h/w-neighbouring bricks are fetched into one L2 instead of eight (the staged loads were ~3.5 us in
  // flight with the interleaved order)
  int b_begin = split, b_end = nbricks, b_step = nsplit;
  if (xcd_map) {
    const int per = cdiv(nbricks, 8);
    b_begin = (split & 7) * per + (split >> 3);
    b_end = min(((split & 7) + 1) * per, nbricks);
    b_step = nsplit >> 3;
  }
  if (fast && b_begin < b_end) {
    if constexpr (PIPED) pset_next(b_begin, true); else set_next(b_begin, true);
    gload();
  }
  for (int brick = b_begin; brick < b_end; brick += b_step) {
    int bdh = brick / g.nbw;
    const int bh0 = (bdh % g.nbh) * BH;
    const int bd0 = ((bdh / g.nbh) % g.nbd) * WBD;
    // every k-step of the brick inside the volume (always, except on the 12^3-and-smaller grids): the pipelined MFMA loop, which
    // also issues the next brick's loads between its MFMAs
    constexpr bool piped = PIPED && std::is_same<T, bf16>::value;
    WG_STAMP(t_s0);
    if (PIPED || fast) {
      __syncthreads();         // every wave is done with the previous brick's LDS image
      WG_STAMP(t_s1);
#ifdef MISEG_WGRAD_STAMPS
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) alone: how long the staged items are still in flight after the MFMA loop
      WG_STAMP(t_s1b);
      WG_STAMP_ADD(6, t_s1, t_s1b);
#endif
      lstore();
      __syncthreads();
      WG_STAMP(t_s2);
      const bool more = brick + b_step < b_end;
      if constexpr (PIPED) pset_next(more ? brick + b_step : brick, more); else set_next(more ? brick + b_step : brick, more);
      if (!piped) gload();     // in flight during the MFMA loop below
      WG_STAMP_ADD(4, t_s0, t_s1);
      WG_STAMP_ADD(5, t_s1, t_s2);
    } else {
    __syncthreads();
    int bid = brick;
    const int bw = bid % g.nbw; bid /= g.nbw;
    const int bh = bid % g.nbh; bid /= g.nbh;
    const int bd = bid % g.nbd;
    const int b = bid / g.nbd;
    const int d0 = bd * WBD, h0 = bh * BH, w0 = bw * BW;
    for (int idx = tid; idx < HROWS * GPR; idx += WG_THREADS) {
      const int row = idx / GPR, cg = idx - row * GPR;
      const int hd = row / (HH * HW), rem = row - hd * (HH * HW);
      const int hh = rem / HW, hw = rem - hh * HW;
      const int d = d0 - 1 + hd, h = h0 - 1 + hh, w = w0 - 1 + hw;
      VT v;
#pragma unroll
      for (int e = 0; e < KPC; ++e) v[e] = from_f32<T>(0.f);
      if (d >= 0 && d < g.D && h >= 0 && h < g.H && w >= 0 && w < g.W) {
        const int c = ci0 + cg * KPC;
        const T* p = x + ((((int64_t)b * g.D + d) * g.H + h) * g.W + w) * ldx + c;
        if (vec_x && c + KPC <= Cin) v = *reinterpret_cast<const VT*>(p);
        else {
#pragma unroll
          for (int e = 0; e < KPC; ++e)
            if (c + e < Cin) v[e] = p[e];
        }
      }
      *reinterpret_cast<VT*>(lx + row * rowb + cg * 16) = v;
    }
    for (int idx = tid; idx < NVOX * GPR; idx += WG_THREADS) {
      const int row = idx / GPR, cg = idx - row * GPR;
      const int vd = row / (BH * BW), rem = row - vd * (BH * BW);
      const int vh = rem / BW, vw = rem - vh * BW;
      const int d = d0 + vd, h = h0 + vh, w = w0 + vw;
      VT v;
#pragma unroll
      for (int e = 0; e < KPC; ++e) v[e] = from_f32<T>(0.f);
      if (d < g.D && h < g.H && w < g.W) {
        const int c = co0 + cg * KPC;
        const T* p = dy + ((((int64_t)b * g.D + d) * g.H + h) * g.W + w) * lddy + c;
        if (vec_dy && c + KPC <= Cout) v = *reinterpret_cast<const VT*>(p);
        else {
#pragma unroll
          for (int e = 0; e < KPC; ++e)
            if (c + e < Cout) v[e] = p[e];
        }
      }
      *reinterpret_cast<VT*>(ld + row * rowb + cg * 16) = v;
    }
    __syncthreads();
    }
    WG_STAMP(t_k0);
    WG_STAMP_ADD(0, t_s0, t_k0);
    if constexpr (std::is_same<T, bf16>::value) {
      // k-step = 32 voxels = 4 h-rows x 8 w at one depth; MFMA k-group fq <-> h-row, element j <-> w
      const int qq = fi >> 2, p4 = (fi & 3) * 4;
      if constexpr (piped) {
        // fully unrolled: the fragments of slot (ks, t + 1) are requested before the MFMAs of slot (ks, t) issue (one wave per
        // SIMD: nobody else hides the LDS latency), and every other slot carries one of the next brick's global loads
        constexpr int NKS = NVOX / 32;
        // per-lane bases; everything that depends on (ks, nt) is a compile-time byte offset (the ds_read immediate field)
        const char* abase = ld + (fq * BW + qq) * rowb + p4 * 2;
        const char* bbase[WG_TPW];
#pragma unroll
        for (int t = 0; t < WG_TPW; ++t) bbase[t] = lx + (fq * HW + qq + toff[t]) * rowb + p4 * 2;
        constexpr int ROWB = WG_CB * (int)sizeof(T) + 16;       // == rowb (checked by the launcher)
        // One wave per SIMD: whatever is issued between two MFMA bursts leaves the matrix pipe idle (an MFMA occupies it for 16
        // cycles, the wave can issue ~3 other instructions meanwhile), so the LDS reads of slot + 2 and the next brick's global
        // loads are dealt out one piece after each MFMA, and sched_barrier pins that order.
        bf16x8 afr[2][3], bfr[3][3];
        constexpr int NSLOT = NKS * WG_TPW;
        auto frag_a = [&](int ks, int mt) {
          const char* a1 = abase + (((ks >> 1) * BH + (ks & 1) * 4) * BW) * ROWB + mt * 32;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1 + 4 * ROWB));
          afr[ks & 1][mt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto frag_b = [&](int sl, int nt) {
          const int ks = sl / WG_TPW, t = sl - ks * WG_TPW;
          const char* a1 = bbase[t] + (((ks >> 1) * HH + (ks & 1) * 4) * HW) * ROWB + nt * 32;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1 + 4 * ROWB));
          bfr[sl % 3][nt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
          for (int i = 0; i < 3; ++i) { if (sl == 0) frag_a(0, i); frag_b(sl, i); }
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
          for (int t = 0; t < WG_TPW; ++t) {
            const int slot = ks * WG_TPW + t, nx = slot + 2;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
              const int mt = i / 3, nt = i - 3 * mt;
              // the accumulators are pinned to AGPRs ("a"): left to the allocator, tiles shuttled between the two register
              // files (600 v_accvgpr moves per brick) and the staged items spilled
              asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[t][mt][nt]) : "v"(bfr[slot % 3][nt]), "v"(afr[ks & 1][mt]));
              if (nx < NSLOT) {
                if (i < 3) frag_b(nx, i);
                else if (i < 6 && nx % WG_TPW == 0) frag_a(nx / WG_TPW, i - 3);
              }
              if (i == 6 && slot < NXI + NDI) pitem(slot);          // one per slot from the start: the last item gets 35 slots (~3.7 us) of lead
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        static_assert(NKS * WG_TPW >= NXI + NDI, "not enough slots for the staged items");
      } else
#pragma unroll 1
      for (int ks = 0; ks < NVOX / 32; ++ks) {
        if (bd0 + (ks >> 1) >= g.D || bh0 + (ks & 1) * 4 >= g.H) continue;   // k-step entirely outside the volume (3^3, 6^3 grids): all zeros
        const int vd = ks >> 1, vh = (ks & 1) * 4 + fq;
        // rows supplied by this lane for the two transposed reads: w = qq and w = 4 + qq
        const int vrow = (vd * BH + vh) * BW + qq;
        const int hrow = (vd * HH + vh) * HW + qq;  // halo row at tap (0,0,0)
        bf16x8 af[3];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
          const char* a1 = ld + vrow * rowb + (mt * 16 + p4) * 2;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1 + 4 * rowb));
          af[mt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        // no guard around a tap slot: the last wave's 7th slot (tap 27) re-computes tap 0 into accumulators nobody stores, which
        // keeps the k-step one basic block (a guarded slot waited for its own LDS reads before every 9 MFMAs)
#pragma unroll
        for (int t = 0; t < WG_TPW; ++t) {
          bf16x8 bfr[3];
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) {
            const char* a1 = lx + (hrow + toff[t]) * rowb + (nt * 16 + p4) * 2;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1 + 4 * rowb));
            bfr[nt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[t][mt][nt], 0, 0, 0);
        }
      }
    } else {
      // fp32: k-step = 4 voxels (w = 4*(ks&1) .. +3 of one (d,h) row); lane k index = fq
#pragma unroll 1
      for (int ks = 0; ks < NVOX / 4; ++ks) {
        const int vrow = ks * 4 + fq;                  // voxel index in brick
        const int vd = vrow / (BH * BW), rem = vrow - vd * (BH * BW);
        const int vh = rem / BW, vw = rem - vh * BW;
        const int hrow = (vd * HH + vh) * HW + vw;
        float af[3];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) af[mt] = *reinterpret_cast<const float*>(ld + vrow * rowb + (mt * 16 + fi) * 4);
#pragma unroll
        for (int t = 0; t < WG_TPW; ++t) {
          float bfr[3];
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) bfr[nt] = *reinterpret_cast<const float*>(lx + (hrow + toff[t]) * rowb + (nt * 16 + fi) * 4);
#pragma unroll
          for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[nt], af[mt], acc[t][mt][nt], 0, 0, 0);
        }
      }
    }
    WG_STAMP(t_k1);
    WG_STAMP_ADD(1, t_k0, t_k1);
    WG_STAMP_ADD(3, t_k1 - 1, t_k1);
  }
  WG_STAMP(t_e0);
  // slab[pair][split][co 48][tap 27][ci 48] (a reducing workgroup reads one contiguous 5 KB run per split);
  // swapped operands => lane holds ci = 16nt + 4fq + r, co = 16mt + fi
  if (direct) {
    // 16 out-channels at a time: tile[co 16][ci 48][tap 27] fp32 (83 KB of the staging LDS), then one contiguous
    // nci*27-float run of dw per out-channel
    float* tile = reinterpret_cast<float*>(lds);
    const int nci = min(WG_CB, Cin - ci0);
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      __syncthreads();
#pragma unroll
      for (int t = 0; t < WG_TPW; ++t) {
        const int tap = wave + WG_WAVES * t;
        if (tap < 27) {
#pragma unroll
          for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(fi * WG_CB + nt * 16 + fq * 4 + r) * 27 + tap] = acc[t][mt][nt][r];
        }
      }
      __syncthreads();
      // this workgroup is the only producer of these elements (nsplit == 1): a plain read-modify-write, not atomics (the L2
      // atomic units sustained ~1 TB/s here: 60 us for the 64 MB of a 768->768 layer)
      const int run = nci * 27;
      if ((run & 3) == 0 && (Cin & 3) == 0) {
        const int run4 = run >> 2;
        for (int idx = tid; idx < 16 * run4; idx += WG_THREADS) {
          const int col = idx / run4, o = (idx - col * run4) * 4;
          const int co = co0 + mt * 16 + col;
          if (co < Cout) {
            f32x4* dst = reinterpret_cast<f32x4*>(dw + ((int64_t)co * Cin + ci0) * 27 + o);
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + col * (WG_CB * 27) + o);
            if (direct != 2) v += *dst;
            __builtin_nontemporal_store(v, dst);      // 64 MB per 768 -> 768 layer, written once and read by the optimiser much later
          }
        }
      } else {
        for (int idx = tid; idx < 16 * run; idx += WG_THREADS) {
          const int col = idx / run, o = idx - col * run;
          const int co = co0 + mt * 16 + col;
          if (co < Cout) {
            float* dst = dw + ((int64_t)co * Cin + ci0) * 27 + o;
            const float v = tile[col * (WG_CB * 27) + o];
            *dst = direct == 2 ? v : *dst + v;
          }
        }
      }
    }
    WG_STAMP(t_e1);
    WG_STAMP_ADD(2, t_e0, t_e1);
    return;
  }
  float* slab = slabs + ((int64_t)pair * nsplit + split) * 27 * WG_CB * WG_CB;
#pragma unroll
  for (int t = 0; t < WG_TPW; ++t) {
    const int tap = wave + WG_WAVES * t;
    if (tap < 27) {
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
          *reinterpret_cast<f32x4*>(slab + ((int64_t)(mt * 16 + fi) * 27 + tap) * WG_CB + nt * 16 + fq * 4) = acc[t][mt][nt];
    }
  }
  WG_STAMP(t_e2);
  WG_STAMP_ADD(2, t_e0, t_e2);
}

template <class T, int WBD, bool PIPED>
__global__ void __launch_bounds__(WG_THREADS) conv3_wgrad_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ dy, int64_t lddy, float* __restrict__ slabs,
                                                                 float* __restrict__ dw, ConvGeom g, int Cin, int Cout, int ncib, int nsplit, int rowb, bool vec_x,
                                                                 bool vec_dy, int direct) {
  conv3_wgrad_body<T, WBD, PIPED>(x, ldx, dy, lddy, slabs, dw, g, Cin, Cout, ncib, nsplit, rowb, vec_x, vec_dy, blockIdx.x, blockIdx.y, direct,
                                  (nsplit & 7) == 0);
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of NARROW layers (round 3; bf16, Cin and Cout in {16, 32}: every 3x3x3 convolution of C-UNETR's image- and half-resolution
// blocks, networks/nets/unetr.py:254-276).  The 48 x 48 channel-pair kernel above spends 9 MFMAs per (tap, k-step) on a tile of which 1/9
// (16 x 16) is real and walks its non-pipelined loop: 0.9 ms per 96^3 layer, 27 % of the C-UNETR step's kernel time.  Here a workgroup owns
// whole 16 x NCO by 16 x NCI tiles: per brick (4 x 8 x 8 voxels = 8 k-steps of 32) each of its four waves takes 7 of the 27 taps,
// 8 x 7 x NCO x NCI MFMAs; bricks are walked persistently, two workgroups per CU (one stages while the other multiplies), the partial
// sums of a workgroup leave as ONE fp32 slab [tap][co][ci] and a second launch adds the slabs up in a fixed order (no atomics: reproducible).
// Operands are the transposed LDS reads of the kernel above (k = voxels: 4 h-rows x 8 w of one depth per k-step).
// ---------------------------------------------------------------------------------------------------------
template <int NCO, int NCI>
__global__ void __launch_bounds__(WG_THREADS, 2) conv3_wgrad_narrow_kernel(const bf16* __restrict__ x, int64_t ldx, const bf16* __restrict__ dy, int64_t lddy,
                                                                           float* __restrict__ slabs, ConvGeom g) {
  constexpr int WBD = 4, NVOX = WBD * BH * BW, HROWS = (WBD + 2) * HH * HW;
  constexpr int CI = 16 * NCI, CO = 16 * NCO;
  constexpr int RBX = CI * 2 + 16, RBD = CO * 2 + 16;      // padded LDS rows (bytes): conflict-free transposed reads
  constexpr int GX = CI / 8, GD = CO / 8;                  // 16-byte groups per row
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* lx = lds;
  char* ld = lds + HROWS * RBX;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fq = lane >> 4, qq = fi >> 2, p4 = (fi & 3) * 4;
  f32x4 acc[WG_TPW][NCO][NCI];
#pragma unroll
  for (int t = 0; t < WG_TPW; ++t)
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
      for (int j = 0; j < NCI; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int toff[WG_TPW];
#pragma unroll
  for (int t = 0; t < WG_TPW; ++t) {
    const int tap = wave + WG_WAVES * t;
    toff[t] = tap < 27 ? (((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3) : 0;
  }
  const int nbricks = g.B * g.nbd * g.nbh * g.nbw;
  // staging: a thread's 16-byte items (halo rows of x, rows of dy) are the same for every brick - their LDS offsets, their offsets from the
  // brick's origin and their position inside the halo are worked out ONCE; per brick an item costs three compares and a select, and the
  // loads of brick i + 1 are in flight while brick i is multiplied (round 3, first form: index arithmetic with four divisions per item and
  // load -> store -> multiply in sequence: 15 us per brick and workgroup for 0.75 us of MFMAs)
  constexpr int NXI = (HROWS * GX + WG_THREADS - 1) / WG_THREADS, NDI = (NVOX * GD + WG_THREADS - 1) / WG_THREADS;
  int xlds[NXI], xrel[NXI], xpos[NXI], dlds[NDI], drel[NDI], dpos[NDI];
#pragma unroll
  for (int j = 0; j < NXI; ++j) {
    const int idx = tid + j * WG_THREADS, row = min(idx, HROWS * GX - 1) / GX, cg = min(idx, HROWS * GX - 1) % GX;
    const int hd = row / (HH * HW), rem = row - hd * (HH * HW), hh = rem / HW, hw = rem - hh * HW;
    xlds[j] = row * RBX + cg * 16;
    xrel[j] = (int)((((int64_t)hd * g.H + hh) * g.W + hw) * ldx + cg * 8);
    xpos[j] = hd | (hh << 8) | (hw << 16) | (idx < HROWS * GX ? 0 : 1 << 24);
  }
#pragma unroll
  for (int j = 0; j < NDI; ++j) {
    const int idx = tid + j * WG_THREADS, row = min(idx, NVOX * GD - 1) / GD, cg = min(idx, NVOX * GD - 1) % GD;
    const int vd = row / (BH * BW), rem = row - vd * (BH * BW), vh = rem / BW, vw = rem - vh * BW;
    dlds[j] = row * RBD + cg * 16;
    drel[j] = (int)((((int64_t)vd * g.H + vh) * g.W + vw) * lddy + cg * 8);
    dpos[j] = vd | (vh << 8) | (vw << 16) | (idx < NVOX * GD ? 0 : 1 << 24);
  }
  const int xsafe = (int)((((int64_t)1 * g.H + 1) * g.W + 1) * ldx);      // the brick's first voxel: always inside the volume
  bf16x8 rx[NXI], rd[NDI];
  unsigned okx = 0, okd = 0;
  auto gload = [&](int brick) {
    int bid = brick;
    const int bw = bid % g.nbw; bid /= g.nbw;
    const int bh = bid % g.nbh; bid /= g.nbh;
    const int bd = bid % g.nbd;
    const int b = bid / g.nbd;
    const int d0 = bd * WBD, h0 = bh * BH, w0 = bw * BW;
    const bf16* xo = x + ((((int64_t)b * g.D + d0 - 1) * g.H + h0 - 1) * g.W + w0 - 1) * ldx;      // halo origin (dereferenced in-bounds only)
    const bf16* yo = dy + ((((int64_t)b * g.D + d0) * g.H + h0) * g.W + w0) * lddy;
    okx = okd = 0;
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
      const int d = d0 - 1 + (xpos[j] & 255), h = h0 - 1 + ((xpos[j] >> 8) & 255), w = w0 - 1 + ((xpos[j] >> 16) & 255);
      const bool ok = !(xpos[j] >> 24) && d >= 0 && d < g.D && h >= 0 && h < g.H && w >= 0 && w < g.W;
      rx[j] = *reinterpret_cast<const bf16x8*>(xo + (ok ? xrel[j] : xsafe));
      okx |= (ok ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < NDI; ++j) {
      const int d = d0 + (dpos[j] & 255), h = h0 + ((dpos[j] >> 8) & 255), w = w0 + ((dpos[j] >> 16) & 255);
      const bool ok = !(dpos[j] >> 24) && d < g.D && h < g.H && w < g.W;
      rd[j] = *reinterpret_cast<const bf16x8*>(yo + (ok ? drel[j] : 0));
      okd |= (ok ? 1u : 0u) << j;
    }
  };
  if ((int)blockIdx.x < nbricks) gload(blockIdx.x);
  for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
    const int d0 = ((brick / (g.nbw * g.nbh)) % g.nbd) * WBD, h0 = ((brick / g.nbw) % g.nbh) * BH;
    __syncthreads();      // every wave is done with the previous brick's images
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NXI; ++j)
      if (!(xpos[j] >> 24)) *reinterpret_cast<bf16x8*>(lx + xlds[j]) = (okx >> j) & 1 ? rx[j] : zero8;
#pragma unroll
    for (int j = 0; j < NDI; ++j)
      if (!(dpos[j] >> 24)) *reinterpret_cast<bf16x8*>(ld + dlds[j]) = (okd >> j) & 1 ? rd[j] : zero8;
    __syncthreads();
    if (brick + (int)gridDim.x < nbricks) gload(brick + gridDim.x);      // in flight during the multiplication below
#pragma unroll 1
    for (int ks = 0; ks < NVOX / 32; ++ks) {
      if (d0 + (ks >> 1) >= g.D || h0 + (ks & 1) * 4 >= g.H) continue;      // k-step entirely outside the volume: all zeros
      const int vd = ks >> 1, vh = (ks & 1) * 4 + fq;
      const int vrow = (vd * BH + vh) * BW + qq;       // rows supplied by this lane for the two transposed reads: w = qq and w = 4 + qq
      const int hrow = (vd * HH + vh) * HW + qq;       // halo row at tap (0, 0, 0)
      bf16x8 af[NCO];
#pragma unroll
      for (int mt = 0; mt < NCO; ++mt) {
        const char* a1 = ld + vrow * RBD + (mt * 16 + p4) * 2;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1 + 4 * RBD));
        af[mt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int t = 0; t < WG_TPW; ++t) {       // (the last wave's 7th slot re-computes tap 0 into accumulators nobody stores)
        bf16x8 bfr[NCI];
#pragma unroll
        for (int nt = 0; nt < NCI; ++nt) {
          const char* a1 = lx + (hrow + toff[t]) * RBX + (nt * 16 + p4) * 2;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a1 + 4 * RBX));
          bfr[nt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int mt = 0; mt < NCO; ++mt)
#pragma unroll
          for (int nt = 0; nt < NCI; ++nt) acc[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[t][mt][nt], 0, 0, 0);
      }
    }
  }
  // slab[workgroup][tap][co][ci]; swapped operands => lane holds ci = 16 nt + 4 fq + r, co = 16 mt + fi
  float* slab = slabs + (int64_t)blockIdx.x * 27 * CO * CI;
#pragma unroll
  for (int t = 0; t < WG_TPW; ++t) {
    const int tap = wave + WG_WAVES * t;
    if (tap < 27) {
#pragma unroll
      for (int mt = 0; mt < NCO; ++mt)
#pragma unroll
        for (int nt = 0; nt < NCI; ++nt) *reinterpret_cast<f32x4*>(slab + ((int64_t)tap * CO + mt * 16 + fi) * CI + nt * 16 + fq * 4) = acc[t][mt][nt];
    }
  }
}

// dw[co][ci][tap] (torch layout) (+)= sum over the workgroups' slabs [tap][co][ci], in slab order
__global__ void __launch_bounds__(256) conv3_wgrad_narrow_reduce_kernel(const float* __restrict__ slabs, int nslabs, float* __restrict__ dw, int Cin, int Cout, int accumulate) {
  // 64 elements per workgroup, the slabs dealt to its four waves (wave w sums slabs w, w + 4, ...: 8 loads in flight per lane), the four
  // partial sums meet in LDS.  (One thread per element over all 512 slabs was a 26 us chain of dependent loads on 27 - 108 workgroups.)
  __shared__ float part[4][64];
  const int n = 27 * Cout * Cin;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;      // e = (tap, co, ci), ci fastest: coalesced slab reads
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.f;
  if (e < n) {
    int k = w;
    for (; k + 28 < nslabs; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += slabs[(int64_t)(k + 4 * u) * n + e];
    }
    for (; k < nslabs; k += 4) a[0] += slabs[(int64_t)k * n + e];
  }
  part[w][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (w == 0 && e < n) {
    const float v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const int ci = e % Cin, co = (e / Cin) % Cout, tap = e / (Cin * Cout);
    float* dst = dw + ((int64_t)co * Cin + ci) * 27 + tap;
    *dst = accumulate == 1 ? *dst + v : v;
  }
}

static constexpr int WG_NARROW_MAX_WG = 512;      // two workgroups per CU
static bool wgrad_narrow(const miseg_conv3_wgrad_params* p) {
  return p->dtype == MISEG_BF16 && (p->Cin == 16 || p->Cin == 32) && (p->Cout == 16 || p->Cout == 32) && ((uintptr_t)p->x % 16 == 0) && ((uintptr_t)p->dy % 16 == 0) &&
         p->ldx % 8 == 0 && p->lddy % 8 == 0;
}
static int wgrad_narrow_workgroups(const miseg_conv3_wgrad_params* p) {
  const int nbricks = p->B * cdiv(p->D, 4) * cdiv(p->H, BH) * cdiv(p->W, BW);
    static const int cap = WG_NARROW_MAX_WG;
  int wg = nbricks < cap ? nbricks : cap;
  if (p->max_workgroups > 0 && p->max_workgroups < wg) wg = p->max_workgroups;      // background form
  return wg;
}
static int conv3_wgrad_narrow_launch(const miseg_conv3_wgrad_params* p, hipStream_t s) {
  ConvGeom g{p->B, p->D, p->H, p->W, cdiv(p->D, 4), cdiv(p->H, BH), cdiv(p->W, BW)};
  const int wg = wgrad_narrow_workgroups(p);
  const size_t lds = (size_t)6 * HH * HW * (p->Cin * 2 + 16) + (size_t)4 * BH * BW * (p->Cout * 2 + 16);
#define NARROW_LAUNCH(A, B_)                                                                                                              \
  do {                                                                                                                                    \
    MISEG_SET_SMEM((conv3_wgrad_narrow_kernel<A, B_>), lds);                                                                                \
    conv3_wgrad_narrow_kernel<A, B_><<<wg, WG_THREADS, lds, s>>>((const bf16*)p->x, p->ldx, (const bf16*)p->dy, p->lddy, (float*)p->workspace, g); \
  } while (0)
  if (p->Cout == 16 && p->Cin == 16) NARROW_LAUNCH(1, 1);
  else if (p->Cout == 16) NARROW_LAUNCH(1, 2);
  else if (p->Cin == 16) NARROW_LAUNCH(2, 1);
  else NARROW_LAUNCH(2, 2);
#undef NARROW_LAUNCH
  const int n = 27 * p->Cout * p->Cin;
  conv3_wgrad_narrow_reduce_kernel<<<cdiv(n, 64), 256, 0, s>>>((const float*)p->workspace, wg, p->dw, p->Cin, p->Cout, p->accumulate);
  MISEG_LAUNCH_CHECK("conv3_wgrad (narrow)");
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of TINY volumes (round 4; bf16, S^3 voxels with S = 3 or 6: encoder10 / decoder5 of C-Swin-UNETR, 384 - 768 channels,
// networks/blocks/dynunet_block.py:100-126).  27 / 216 voxels against a 16 - 64 MB fp32 gradient: the layer is bound by the WRITE of dw, and
// the 48 x 48 channel-pair kernel above - one 96 KB workgroup per CU, MFMAs over a 256-voxel brick that is nine tenths padding, then a
// 249 KB epilogue with nothing beside it - spent 25 - 45 us per unit on it (139 of the 437 us of a step's grouped launch).
// Here the TAPS are the N side of the matrix product: D[co][tap] = sum_v dy^T[co][v] * X_ci[v][tap] with X_ci[v][tap] = x[v + tap][ci], one
// product per input channel - the accumulator tile of a lane (co = 4 fq + r, tap = fi) is then a run of consecutive addresses of the torch
// layout [co][ci][27] over the 16 lanes of a quarter wave: no transposition, no LDS round trip, stores as they come.  MFMA efficiency is
// beside the point (27 of 32 columns, 27 of 32 k at 3^3): the whole 768 -> 768 layer is 74 K MFMAs.  A workgroup = 48 out-channels x 16
// in-channels (4 per wave), ~40 KB of LDS at 6^3: four workgroups per CU, so one's stores overlap another's staging and products.
// ---------------------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(256) conv3_wgrad_tiny_kernel(const bf16* __restrict__ x, int64_t ldx, const bf16* __restrict__ dy, int64_t lddy,
                                                               float* __restrict__ dw, int B, int Cin, int Cout, int add) {
  constexpr int P = S + 2, NP = P * P * P, V = S * S * S, NK = (V + 31) / 32, KP = NK * 32 + 8;
  __shared__ __attribute__((aligned(16))) bf16 xh[NP * 16];      // zero-padded halo of x, [row][16 in-channels]
  __shared__ __attribute__((aligned(16))) bf16 dyT[48 * KP];     // dy transposed, [out-channel][voxel] (voxels >= V: zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fq = lane >> 4;
  const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * 48;
  f32x4 acc[4][3][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) acc[c][mt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // tap of this lane in the two 16-wide tap tiles -> row offset in the halo (taps >= 27 read row 0 and are never stored: whatever they hold)
  int toff[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int tap = 16 * tt + fi;
    toff[tt] = tap < 27 ? ((tap / 9) * P + (tap / 3) % 3) * P + tap % 3 : 0;
  }
  const bf16 zero = (bf16)0.f;
  for (int b = 0; b < B; ++b) {
    if (b) __syncthreads();
    for (int idx = tid; idx < NP * 2; idx += 256) {
      const int row = idx >> 1, half = idx & 1;
      const int pd = row / (P * P), ph = (row / P) % P, pw = row % P;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = zero;
      if (pd >= 1 && pd <= S && ph >= 1 && ph <= S && pw >= 1 && pw <= S)
        v = *reinterpret_cast<const bf16x8*>(x + ((((int64_t)b * S + pd - 1) * S + ph - 1) * S + pw - 1) * ldx + ci0 + 8 * half);
      *reinterpret_cast<bf16x8*>(xh + row * 16 + 8 * half) = v;
    }
    for (int idx = tid; idx < NK * 32 * 6; idx += 256) {
      const int v = idx / 6, part = idx - v * 6;
      bf16x8 t;
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = zero;
      if (v < V) t = *reinterpret_cast<const bf16x8*>(dy + ((int64_t)b * V + v) * lddy + co0 + 8 * part);
#pragma unroll
      for (int e = 0; e < 8; ++e) dyT[(8 * part + e) * KP + v] = t[e];
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < NK; ++s) {
      // the 8 voxels (k) this lane supplies: their halo rows at tap (0, 0, 0); voxels beyond the volume read row 0 against a zero dy column
      int rb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int v = 32 * s + 8 * fq + j;
        rb[j] = v < V ? ((v / (S * S)) * P + (v / S) % S) * P + v % S : 0;
      }
      bf16x8 afr[3];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) afr[mt] = *reinterpret_cast<const bf16x8*>(dyT + (16 * mt + fi) * KP + 32 * s + 8 * fq);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16* xc = xh + 4 * wave + c;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          bf16x8 bfr;
#pragma unroll
          for (int j = 0; j < 8; ++j) bfr[j] = xc[(rb[j] + toff[tt]) * 16];
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) acc[c][mt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[mt], bfr, acc[c][mt][tt], 0, 0, 0);
        }
      }
    }
  }
  // lane holds co = 16 mt + 4 fq + r, tap = 16 tt + fi of in-channel ci0 + 4 wave + c: 16 lanes write 16 consecutive floats
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ci = ci0 + 4 * wave + c;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int tap = 16 * tt + fi;
        if (tap < 27) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = dw + ((int64_t)(co0 + 16 * mt + 4 * fq + r) * Cin + ci) * 27 + tap;
            *dst = add ? *dst + acc[c][mt][tt][r] : acc[c][mt][tt][r];
          }
        }
      }
  }
}

static bool wgrad_tiny_shape(int B, int D, int H, int W, int Cin, int Cout, int dtype) {
  return dtype == MISEG_BF16 && D == H && H == W && (D == 3 || D == 6) && B >= 1 && B <= 64 && Cin % 16 == 0 && Cout % 48 == 0;
}
static bool wgrad_tiny(const miseg_conv3_wgrad_params* p) {
  return wgrad_tiny_shape(p->B, p->D, p->H, p->W, p->Cin, p->Cout, p->dtype) && ((uintptr_t)p->x % 16 == 0) && ((uintptr_t)p->dy % 16 == 0) && p->ldx % 8 == 0 &&
         p->lddy % 8 == 0;
}
static int conv3_wgrad_tiny_launch(const miseg_conv3_wgrad_params* p, hipStream_t s) {
  const dim3 grid(p->Cin / 16, p->Cout / 48);
  const int add = p->accumulate == 1 ? 1 : 0;      // 0 / 2 (a slot known to hold zeros): plain stores
  if (p->D == 3)
    conv3_wgrad_tiny_kernel<3><<<grid, 256, 0, s>>>((const bf16*)p->x, p->ldx, (const bf16*)p->dy, p->lddy, p->dw, p->B, p->Cin, p->Cout, add);
  else
    conv3_wgrad_tiny_kernel<6><<<grid, 256, 0, s>>>((const bf16*)p->x, p->ldx, (const bf16*)p->dy, p->lddy, p->dw, p->B, p->Cin, p->Cout, add);
  MISEG_LAUNCH_CHECK("conv3_wgrad (tiny)");
  return MISEG_OK;
}

// Several layers in one launch (the small-grid weight gradients of a backward pass, queued by the host): descriptors travel
// in the kernel arguments, a workgroup finds its layer by its index range.
struct WgradLayer {
  const void* x; const void* dy; float* slabs; float* dw;
  int64_t ldx, lddy;
  ConvGeom g;
  int Cin, Cout, ncib, nsplit;
  int wg0, flags;                  // first workgroup; bit0 vec_x, bit1 vec_dy, bits 2-3 direct mode, bit 4 pipelined loop
  int tiles, groups, spg, rb0;     // reduce launch: (co, ci-block) tiles x split groups, splits per group, first reduce block
};
static constexpr int WG_GROUP_MAX = 24;
struct WgradGroup { WgradLayer l[WG_GROUP_MAX]; int n; };

// Round 5: ONE loop form per kernel (PIPED is a template parameter, the host launches the pipelined layers and the others as two grids).  With
// both bodies behind a per-unit branch the 252 accumulator registers of the two paths met and the kernel spilled 1.5 KB per lane - in the
// grouped launch only: the single-layer kernel always had one instantiation per form.
template <class T, int WBD, bool PIPED, bool WALK>
__global__ void __launch_bounds__(WG_THREADS) conv3_wgrad_group_kernel(const WgradGroup grp, int rowb, int total_units) {
  // one unit = one (layer, channel pair, split).  Normal form (WALK = false): one workgroup per unit.  Background form (max_workgroups of the
  // first descriptor: the launch runs on a side stream beside another stream's kernels): a capped grid walks the units, longest layers first.
  for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
    if (unit != (int)blockIdx.x) {
      if constexpr (!WALK) break;
      __syncthreads();      // the LDS images of the previous unit have been read
    }
    int li = 0;
    for (int i = 1; i < grp.n; ++i)
      if (unit >= grp.l[i].wg0) li = i;
    const WgradLayer& L = grp.l[li];
    const int local = unit - L.wg0;
    conv3_wgrad_body<T, WBD, PIPED>((const T*)L.x, L.ldx, (const T*)L.dy, L.lddy, L.slabs, L.dw, L.g, L.Cin, L.Cout, L.ncib, L.nsplit, rowb, PIPED || (L.flags & 1),
                                    PIPED || ((L.flags >> 1) & 1), local % L.nsplit, local / L.nsplit, (L.flags >> 2) & 3, ((L.nsplit | L.wg0) & 7) == 0);
  }
}

// dw[co][ci0..+48][tap] += sum over this group's splits of slab[pair][split][co%48][tap][0..48).
// block = (co, ci-block, split group): one contiguous 5 KB slab run per split, the 27x48 result is transposed through LDS
// so the atomics land on one contiguous 5 KB run of the torch-layout gradient.
__device__ __forceinline__ void conv3_wgrad_reduce_body(const float* __restrict__ slabs, float* __restrict__ dw, int Cin, int Cout, int ncib, int nsplit,
                                                        int splits_per_group, int tile_idx, int group_idx) {
  __shared__ float tile[WG_CB * 28];
  constexpr int64_t SLAB = 27 * WG_CB * WG_CB;
  constexpr int NE = 27 * WG_CB;            // elements of one (co, ci-block) tile, (tap, cil) with cil fastest
  const int co = tile_idx / ncib, cib = tile_idx % ncib;
  const int pair = (co / WG_CB) * ncib + cib;
  const int k0 = group_idx * splits_per_group, k1 = min(nsplit, k0 + splits_per_group);
  const float* base = slabs + (int64_t)pair * nsplit * SLAB + (int64_t)(co % WG_CB) * NE;
  float acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) acc[j] = 0.f;
  int k = k0;
  for (; k + 3 < k1; k += 4) {     // 24 independent loads in flight per lane
    float v[4][6];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* s = base + (int64_t)(k + u) * SLAB;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int e = threadIdx.x + 256 * j;
        v[u][j] = e < NE ? s[e] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] += (v[0][j] + v[1][j]) + (v[2][j] + v[3][j]);
  }
  for (; k < k1; ++k) {
    const float* s = base + (int64_t)k * SLAB;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int e = threadIdx.x + 256 * j;
      if (e < NE) acc[j] += s[e];
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int e = threadIdx.x + 256 * j;
    if (e < NE) tile[(e % WG_CB) * 28 + e / WG_CB] = acc[j];
  }
  __syncthreads();
  const int ci0 = cib * WG_CB;
  const int nci = min(WG_CB, Cin - ci0);
  float* out = dw + ((int64_t)co * Cin + ci0) * 27;
  if (splits_per_group >= nsplit) {      // the only block of this tile: plain read-modify-write
    for (int o = threadIdx.x; o < nci * 27; o += 256) out[o] += tile[(o / 27) * 28 + o % 27];
  } else {
    for (int o = threadIdx.x; o < nci * 27; o += 256) atomicAdd(out + o, tile[(o / 27) * 28 + o % 27]);
  }
}

__global__ void __launch_bounds__(256) conv3_wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int Cin, int Cout, int ncib, int nsplit,
                                                                 int splits_per_group) {
  conv3_wgrad_reduce_body(slabs, dw, Cin, Cout, ncib, nsplit, splits_per_group, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(256) conv3_wgrad_reduce_group_kernel(const WgradGroup grp) {
  int li = -1;
  for (int i = 0; i < grp.n; ++i)
    if (grp.l[i].tiles > 0 && (int)blockIdx.x >= grp.l[i].rb0) li = i;
  const WgradLayer& L = grp.l[li];
  const int local = blockIdx.x - L.rb0;
  conv3_wgrad_reduce_body(L.slabs, L.dw, L.Cin, L.Cout, L.ncib, L.nsplit, L.spg, local % L.tiles, local / L.tiles);
}

}  // namespace miseg

using namespace miseg;

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// fast-path plan: output-channel tiles per block and the split over 96-byte channel chunks (small grids)
static void fwd96_plan(int nbricks, int Cout, int nchunks, int* nt, int* ksplit) {
  *nt = Cout <= 16 ? 1 : Cout <= 32 ? 2 : 3;
  int blocks = nbricks * cdiv(Cout, 16 * (*nt));
  int ks = 1;
  if (nbricks <= 4 && nchunks > 1) {
    // a volume of a few bricks (6^3, 3^3: 384 - 768 channels): the launch is one serial chain of staging + 14 phases per chunk however it is
    // cut, so every chunk gets its own workgroup and the 48-channel tile keeps the halo staging shared by three times the MFMAs
    // (scripts/micro/f96_plan_sweep.py: 768 -> 384 at 6^3 33.8 -> 28.2 us, 768 -> 768 at 3^3 22.7 -> 19.7 us)
    ks = nchunks;
  } else {
    if (blocks < 256 && *nt > 1) { *nt = 1; blocks = nbricks * cdiv(Cout, 16); }
    if (blocks < 256 && nchunks > 1) {
      ks = cdiv(512, blocks);       // small grids: about two workgroups per CU (384 -> 192 at 12^3: split 8 / 4 / 2 = 30.1 / 25.7 / 31.5 us)
      if (ks > nchunks) ks = nchunks;
    }
  }
  *ksplit = ks;
}

extern "C" size_t miseg_conv3_fwd_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int dtype) {
  const int esz = dtype == MISEG_F32 ? 4 : 2;
  const int rowbytes = conv3_k96(Cin, esz, conv3_pad_min_bytes()) * esz;
  if (!rowbytes) return 0;
  int nt, ks;
  fwd96_plan(B * cdiv(D, FBD) * cdiv(H, FBH) * cdiv(W, FBW), Cout, rowbytes / (16 * conv3_gpt(Cin, esz, conv3_pad_min_bytes())), &nt, &ks);
  return ks > 1 ? (size_t)ks * B * D * H * W * Cout * sizeof(float) : 0;
}

extern "C" int miseg_conv3_fuses_shortcut(int B, int D, int H, int W, int Cin, int Cout, int Csc, int dtype) {
  if (dtype != MISEG_BF16) return 0;
  const int k96 = conv3_k96(Cin, 2, conv3_pad_min_bytes());
  if (!k96 || conv3_gpt(Cin, 2, conv3_pad_min_bytes()) != 6 || Csc <= 0 || Csc % 48 != 0) return 0;
  int nt, ks;
  fwd96_plan(B * cdiv(D, FBD) * cdiv(H, FBH) * cdiv(W, FBW), Cout, k96 * 2 / 96, &nt, &ks);
  return ks == 1 ? 1 : 0;
}

// 1 when miseg_conv3_fwd serves this problem with the tiny-volume weight-streaming kernel (conv3_fwd_tiny_kernel): the same predicate as its launch
extern "C" int miseg_conv3_fwd_tiny(int B, int D, int H, int W, int Cin, int Cout, int dtype) {
  if (dtype != MISEG_BF16 || Cin % 8 != 0) return 0;
  const int k96 = conv3_k96(Cin, 2, conv3_pad_min_bytes());
  if (!k96 || conv3_gpt(Cin, 2, conv3_pad_min_bytes()) != 6) return 0;
  const int nchunks = k96 * 2 / 96, nvox = D * H * W;
  int nt, ks;
  fwd96_plan(B * cdiv(D, FBD) * cdiv(H, FBH) * cdiv(W, FBW), Cout, nchunks, &nt, &ks);
  return (ks > 1 && ks == nchunks && nvox <= 256 && (size_t)(D + 2) * (H + 2) * (W + 2) * TINY_ROWB <= 64 * 1024) ? 1 : 0;
}

extern "C" int miseg_conv3_fuses_fwd_shortcut(int B, int D, int H, int W, int Cin, int Cout, int dtype) {
  return miseg_conv3_fuses_shortcut(B, D, H, W, Cin, Cout, Cin, dtype);      // (the same launch conditions; the 1x1x1 term's K side is Cin itself)
}

extern "C" int miseg_conv3_fuses_s2c(int B, int D, int H, int W, int Cin, int Cout, int s2c_C, int dtype) {
  const int esz = dtype == MISEG_F32 ? 4 : 2;
  const int k96 = conv3_k96(Cin, esz, conv3_pad_min_bytes());
  if (!k96 || (D | H | W) & 1 || s2c_C <= 0 || s2c_C >= Cout || s2c_C % (16 / esz) != 0) return 0;
  int nt, ks;
  fwd96_plan(B * cdiv(D, FBD) * cdiv(H, FBH) * cdiv(W, FBW), Cout, k96 * esz / (16 * conv3_gpt(Cin, esz, conv3_pad_min_bytes())), &nt, &ks);
  return (ks == 1 && s2c_C % (16 * nt) == 0) ? 1 : 0;
}

extern "C" int miseg_conv3_fwd_splits(int B, int D, int H, int W, int Cin, int Cout, int dtype) {
  const int esz = dtype == MISEG_F32 ? 4 : 2;
  const int rowbytes = conv3_k96(Cin, esz, conv3_pad_min_bytes()) * esz;
  if (!rowbytes) return 1;
  int nt, ks;
  const int nchunks = rowbytes / (16 * conv3_gpt(Cin, esz, conv3_pad_min_bytes()));
  fwd96_plan(B * cdiv(D, FBD) * cdiv(H, FBH) * cdiv(W, FBW), Cout, nchunks, &nt, &ks);
  const int cps = cdiv(nchunks, ks);
  return cdiv(nchunks, cps);
}

template <class T>
static int conv3_fwd_launch(const miseg_conv3_params* p, hipStream_t s) {
  constexpr int KPC = Vec16<T>::N;
  const int k96 = conv3_k96(p->Cin, (int)sizeof(T), conv3_pad_min_bytes());
  const int CinP = k96 ? k96 : round_up(p->Cin, KPC);
  const int rowbytes = CinP * (int)sizeof(T);
  ConvGeom g{p->B, p->D, p->H, p->W, cdiv(p->D, BD), cdiv(p->H, BH), cdiv(p->W, BW)};
  const int nbricks = g.B * g.nbd * g.nbh * g.nbw;
  const bool vec_x = ((uintptr_t)p->x % 16 == 0) && (p->ldx % KPC == 0);
  // ---- fast path: 96-byte channel chunks, planar LDS images (weights must come from the planar pack: same predicate
  // in miseg_pack_conv3_weight: conv3_k96)
  if (k96) {
    ConvGeom gf{p->B, p->D, p->H, p->W, cdiv(p->D, FBD), cdiv(p->H, FBH), cdiv(p->W, FBW)};
    const int nbr = gf.B * gf.nbd * gf.nbh * gf.nbw;
    int nt, ksplit;
    const int gpt = conv3_gpt(p->Cin, (int)sizeof(T), conv3_pad_min_bytes());
    const int nchunks = rowbytes / (16 * gpt);
    fwd96_plan(nbr, p->Cout, nchunks, &nt, &ksplit);
    const int cps = cdiv(nchunks, ksplit);
    ksplit = cdiv(nchunks, cps);
    float* scratch = nullptr;
    if (ksplit > 1) {
      MISEG_REQUIRE(p->workspace, MISEG_E_BADARG, "conv3_fwd: workspace required (miseg_conv3_fwd_workspace_bytes)");
      scratch = (float*)p->workspace;
    }
    const bool vec_y = ((uintptr_t)p->y % 16 == 0) && (p->ldy % KPC == 0);
    const int CoP = round_up(p->Cout, 16);
    if (p->sc_x) {      // the fused 1x1x1 shortcut term (miseg_conv3_fuses_shortcut)
      constexpr bool is_bf16 = std::is_same<T, bf16>::value;
      MISEG_REQUIRE(is_bf16 && gpt == 6 && ksplit == 1 && p->sc_w && p->sc_C > 0 && p->sc_C % (6 * KPC) == 0,
                    MISEG_E_UNSUPPORTED, "conv3_fwd: fused shortcut on this shape / dtype (ask miseg_conv3_fuses_shortcut first)");
      MISEG_REQUIRE((uintptr_t)p->sc_x % 16 == 0 && p->ld_sc_x % KPC == 0 && (uintptr_t)p->sc_w % 16 == 0, MISEG_E_BADARG, "conv3_fwd: shortcut operands must be 16-byte aligned");
    }
    if (p->fs_w) {      // the block's 1x1x1 shortcut convolution as a second output (miseg_conv3_fuses_fwd_shortcut)
      constexpr bool is_bf16_ = std::is_same<T, bf16>::value;
      MISEG_REQUIRE(is_bf16_ && gpt == 6 && ksplit == 1 && p->fs_y && p->Cin % (6 * KPC) == 0 && vec_x, MISEG_E_UNSUPPORTED,
                    "conv3_fwd: second (1x1x1) output on this shape / dtype (ask miseg_conv3_fuses_fwd_shortcut first)");
      MISEG_REQUIRE((uintptr_t)p->fs_w % 16 == 0, MISEG_E_BADARG, "conv3_fwd: fs_w must be 16-byte aligned");
    }
    if (p->s2c_out) {   // the left channels stored in space-to-channel order (miseg_conv3_fuses_s2c)
      MISEG_REQUIRE(ksplit == 1 && p->s2c_C > 0 && p->s2c_C < p->Cout && p->s2c_C % (16 * nt) == 0 && p->D % 2 == 0 && p->H % 2 == 0 && p->W % 2 == 0,
                    MISEG_E_UNSUPPORTED, "conv3_fwd: space-to-channel store on this shape (ask miseg_conv3_fuses_s2c first)");
      MISEG_REQUIRE((uintptr_t)p->s2c_out % 16 == 0 && p->s2c_C % KPC == 0, MISEG_E_BADARG, "conv3_fwd: s2c_out must be 16-byte aligned");
    }
    size_t lds = (size_t)gpt * FPS * 16 + (size_t)2 * 12 * 16 * nt * 16;
    if (p->background && lds < 83 * 1024) lds = 83 * 1024;      // more than half of the 160 KB: one workgroup per CU
    MISEG_REQUIRE((int64_t)p->B * p->D * p->H * p->W < (1LL << 31), MISEG_E_UNSUPPORTED, "conv3_fwd: more than 2^31 voxels");
    const int ny = cdiv(p->Cout, 16 * nt);
    if constexpr (std::is_same<T, bf16>::value) {
      // tiny volumes whose every chunk is a split of its own: the weight-streaming kernel (same pack, same slabs)
      const int nvox = p->D * p->H * p->W;
      if (scratch && gpt == 6 && ksplit == nchunks && nvox <= 256 && vec_x && !p->background && p->Cin % 8 == 0 &&
          (size_t)(p->D + 2) * (p->H + 2) * (p->W + 2) * TINY_ROWB <= 64 * 1024) {
        const size_t lds_t = (size_t)(p->D + 2) * (p->H + 2) * (p->W + 2) * TINY_ROWB;
        if (nvox <= 32) {
          dim3 gt(cdiv(CoP / 16, 4), nchunks, p->B);
          (void)hipFuncSetAttribute((const void*)conv3_fwd_tiny_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
          conv3_fwd_tiny_kernel<2, 1><<<gt, 256, lds_t, s>>>((const bf16*)p->x, p->ldx, (const bf16*)p->wpk, p->D, p->H, p->W, p->Cin, p->Cout, CoP, scratch, p->B * nvox);
        } else {
          dim3 gt(CoP / 16, nchunks, p->B);
          (void)hipFuncSetAttribute((const void*)conv3_fwd_tiny_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
          conv3_fwd_tiny_kernel<4, 4><<<gt, 256, lds_t, s>>>((const bf16*)p->x, p->ldx, (const bf16*)p->wpk, p->D, p->H, p->W, p->Cin, p->Cout, CoP, scratch, p->B * nvox);
        }
        MISEG_LAUNCH_CHECK("conv3_fwd_tiny");
        if (p->defer_slabs) {
          MISEG_REQUIRE(!p->res, MISEG_E_UNSUPPORTED, "conv3_fwd: defer_slabs with a fused residual");
          return MISEG_OK;
        }
        return slabs_to_out_stats(scratch, ksplit, p->y, p->ldy, p->res, p->ldres, p->B, nvox, p->Cout, p->dtype, (double*)p->stat, s);
      }
    }
    dim3 grid(nbr * ny, 1, ksplit);
#define F96_LAUNCH(n, wd, epi, gp)                                                                                                           \
    (void)hipFuncSetAttribute((const void*)conv3_fwd96_kernel<T, n, wd, epi, gp>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
    conv3_fwd96_kernel<T, n, wd, epi, gp><<<grid, 256, lds, s>>>((const T*)p->x, p->ldx, (T*)p->y, p->ldy, (const T*)p->wpk, gf, p->Cin, CinP, \
                                                                 p->Cout, CoP, vec_x, vec_y, scratch, cps, scratch ? nullptr : (const T*)p->res, \
                                                                 p->ldres, (double*)p->stat, ny, (const T*)p->sc_x, p->ld_sc_x, (const T*)p->sc_w, p->sc_C, \
                                                                 (T*)p->s2c_out, p->s2c_C, (const T*)p->fs_w, (T*)p->fs_y, p->ld_fs_y, (double*)p->fs_stat)
#define F96_CASE(n, wd, gp)                                                                                                                  \
  case n:                                                                                                                                   \
    if (!scratch && (p->res || p->stat || p->sc_x || p->s2c_out || p->fs_w)) { F96_LAUNCH(n, wd, true, gp); } else { F96_LAUNCH(n, wd, false, gp); }   \
    break;
    if (gpt == 6) { switch (nt) { F96_CASE(1, 3, 6) F96_CASE(2, 2, 6) F96_CASE(3, 3, 6) } }
    else if (gpt == 4) { switch (nt) { F96_CASE(1, 3, 4) F96_CASE(2, 2, 4) F96_CASE(3, 3, 4) } }
    else { switch (nt) { F96_CASE(1, 3, 2) F96_CASE(2, 2, 2) F96_CASE(3, 3, 2) } }
#undef F96_CASE
#undef F96_LAUNCH
    MISEG_LAUNCH_CHECK("conv3_fwd96");
    if (scratch && p->defer_slabs) {      // the caller's next launch sums the slabs itself (miseg_instnorm_fwd_slabs)
      MISEG_REQUIRE(!p->res, MISEG_E_UNSUPPORTED, "conv3_fwd: defer_slabs with a fused residual");
      return MISEG_OK;
    }
    if (scratch)      // sum of the slabs + residual -> y, with the statistics of y when asked for (one launch)
      return slabs_to_out_stats(scratch, ksplit, p->y, p->ldy, p->res, p->ldres, p->B, p->D * p->H * p->W, p->Cout, p->dtype, (double*)p->stat, s);
    return MISEG_OK;
  }
  MISEG_REQUIRE(!p->res && !p->stat, MISEG_E_UNSUPPORTED, "conv3_fwd: fused residual / statistics need 96-byte channel chunks");
  int chunk_bytes;
  if (rowbytes <= 128) chunk_bytes = rowbytes;
  else if (rowbytes % 128 == 0) chunk_bytes = 128;
  else chunk_bytes = 64;
  const int chunk_elems = chunk_bytes / (int)sizeof(T);
  const int rowb = chunk_bytes + 16;
  const size_t lds = (size_t)(BD + 2) * HH * HW * rowb;
  const int nt = p->Cout <= 16 ? 1 : p->Cout <= 32 ? 2 : p->Cout <= 48 ? 3 : p->Cout <= 64 ? 4 : (p->Cout % 96 == 0 || p->Cout > 128) ? 6 : 4;
  dim3 grid(nbricks, cdiv(p->Cout, 16 * nt));
#define FWD_CASE(n)                                                                                                                             \
  case n:                                                                                                                                       \
    hipFuncSetAttribute((const void*)conv3_fwd_kernel<T, n>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
    conv3_fwd_kernel<T, n><<<grid, 256, lds, s>>>((const T*)p->x, p->ldx, (T*)p->y, p->ldy, (const T*)p->wpk, g, p->Cin, CinP, p->Cout, chunk_elems, \
                                                  rowb, vec_x);                                                                                 \
    break;
  switch (nt) { FWD_CASE(1) FWD_CASE(2) FWD_CASE(3) FWD_CASE(4) FWD_CASE(6) }
#undef FWD_CASE
  MISEG_LAUNCH_CHECK("conv3_fwd");
  return MISEG_OK;
}

extern "C" int miseg_conv3_fwd(const miseg_conv3_params* p, miseg_stream_t s_) {
  MISEG_REQUIRE(p && p->x && p->y && p->wpk, MISEG_E_BADARG, "conv3_fwd: null pointer");
  MISEG_REQUIRE(p->B > 0 && p->D > 0 && p->H > 0 && p->W > 0 && p->Cin > 0 && p->Cout > 0, MISEG_E_BADARG, "conv3_fwd: bad shape");
  MISEG_REQUIRE(p->ldx >= p->Cin && p->ldy >= p->Cout, MISEG_E_BADARG, "conv3_fwd: row stride smaller than channel count");
  if (p->dtype == MISEG_F32) return conv3_fwd_launch<float>(p, (hipStream_t)s_);
  if (p->dtype == MISEG_BF16) return conv3_fwd_launch<bf16>(p, (hipStream_t)s_);
  return set_error(MISEG_E_BADARG, "conv3_fwd: dtype %d", p->dtype);
}

extern "C" size_t miseg_pack_conv3_elems(int Cin, int Cout, int dtype, int which) {
  const int per = dtype == MISEG_F32 ? 24 : 48;
  const int CinP = round_up(Cin, per), CoutP = round_up(Cout, per);      // a padded K side ends on a whole 96-byte chunk
  // large enough for either layout (the phase-ordered pack of the fast path holds 14 x 12 = 168 group slots per chunk for 27 x 6 = 162 groups)
  return which == 0 ? (size_t)28 * CinP * round_up(Cout, 16) : (size_t)28 * CoutP * round_up(Cin, 16);
}

extern "C" int miseg_conv3_k96(int C, int dtype) { return conv3_k96(C, dtype == MISEG_F32 ? 4 : 2, conv3_pad_min_bytes()); }

extern "C" int miseg_pack_conv3_tiles(int Cin, int Cout, int dtype) {
  const int kf = miseg_conv3_k96(Cin, dtype), kb = miseg_conv3_k96(Cout, dtype);
  return cdiv(kf > Cin ? kf : Cin, PK_T) * cdiv(kb > Cout ? kb : Cout, PK_T);
}

extern "C" int miseg_pack_conv3_weight(const miseg_pack_conv3_params* p, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->w && (p->fwd_pack || p->bwd_pack) && p->Cin > 0 && p->Cout > 0, MISEG_E_BADARG, "pack_conv3_weight: bad args");
  return dispatch_dtype(p->dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    constexpr int KPC = Vec16<T>::N;
    const int kf = conv3_k96(p->Cin, (int)sizeof(T), conv3_pad_min_bytes()), kb = conv3_k96(p->Cout, (int)sizeof(T), conv3_pad_min_bytes());
    const int CinP = kf ? kf : round_up(p->Cin, KPC), CoutP = kb ? kb : round_up(p->Cout, KPC);
    const int fgpt = kf ? conv3_gpt(p->Cin, (int)sizeof(T), conv3_pad_min_bytes()) : 0, bgpt = kb ? conv3_gpt(p->Cout, (int)sizeof(T), conv3_pad_min_bytes()) : 0;
    MISEG_REQUIRE((!p->fwd_pack || (uintptr_t)p->fwd_pack % 16 == 0) && (!p->bwd_pack || (uintptr_t)p->bwd_pack % 16 == 0), MISEG_E_BADARG,
                  "pack_conv3_weight: packs must be 16-byte aligned");
    dim3 grid(cdiv(kf > p->Cin ? kf : p->Cin, PK_T), cdiv(kb > p->Cout ? kb : p->Cout, PK_T));     // the padded K groups get their (zero) tiles
    pack_conv3_kernel<T><<<grid, 256, 0, s>>>(p->w, (T*)p->fwd_pack, (T*)p->bwd_pack, p->Cin, p->Cout, CinP, CoutP, round_up(p->Cin, 16), round_up(p->Cout, 16),
                                              fgpt, bgpt);
    MISEG_LAUNCH_CHECK("pack_conv3_weight");
    return MISEG_OK;
  });
}

extern "C" int miseg_pack_conv3_batch(const miseg_pack_conv3_desc* descs, int n, int total_tiles, int dtype, const int64_t* params_version, int64_t* state,
                                      miseg_stream_t s_) {
  MISEG_REQUIRE(descs && n > 0 && total_tiles > 0, MISEG_E_BADARG, "pack_conv3_batch: bad args");
  MISEG_REQUIRE((params_version == nullptr) == (state == nullptr), MISEG_E_BADARG, "pack_conv3_batch: params_version and state go together");
  return dispatch_dtype(dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    static const int cap = refresh_max_wg("MISEG_PACK_WG", REFRESH_PACK_WG);
    pack_conv3_batch_kernel<T><<<total_tiles < cap ? total_tiles : cap, 256, 0, (hipStream_t)s_>>>(descs, n, total_tiles, params_version, state,
                                                                                                                         conv3_pad_min_bytes());
    MISEG_LAUNCH_CHECK("pack_conv3_batch");
    return MISEG_OK;
  });
}

extern "C" int miseg_opt_step_pack_conv3(const miseg_opt_step_params* p, const miseg_pack_conv3_desc* descs, const miseg_opt_pack_map* map, int n, int total_tiles,
                                         int dtype, int64_t* pack_state, miseg_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  MISEG_REQUIRE(p && p->struct_size == sizeof(miseg_opt_step_params), MISEG_E_BADARG, "opt_step_pack_conv3: struct_size %u != %zu", p ? p->struct_size : 0u,
                sizeof(miseg_opt_step_params));
  MISEG_REQUIRE(descs && map && n > 0 && total_tiles > 0 && p->grad && p->state1 && p->steps, MISEG_E_BADARG, "opt_step_pack_conv3: null pointer / empty table");
  MISEG_REQUIRE(p->kind == MISEG_OPT_ADAMW || p->kind == MISEG_OPT_ADAM || p->kind == MISEG_OPT_SGD_NESTEROV, MISEG_E_BADARG, "opt_step_pack_conv3: kind %d", p->kind);
  MISEG_REQUIRE(p->kind == MISEG_OPT_SGD_NESTEROV || p->state2, MISEG_E_BADARG, "opt_step_pack_conv3: Adam needs state2");
  MISEG_REQUIRE(p->count_n >= 0, MISEG_E_BADARG, "opt_step_pack_conv3: count_n must name the number of parameters of the whole step (or 0)");
  const int rc = dispatch_dtype(dtype, [&](auto* tag) -> int {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    const int cap = 2048;      // (7 loads + 3 stores of 16 bytes per thread and tile in flight: more workgroups than the refresh kernel's 512)
    opt_pack_conv3_batch_kernel<T><<<total_tiles < cap ? total_tiles : cap, 256, 0, s>>>(descs, map, n, total_tiles, conv3_pad_min_bytes(), p->kind, p->grad, p->state1,
                                                                                         p->kind == MISEG_OPT_SGD_NESTEROV ? nullptr : p->state2, p->used, p->steps, p->lr,
                                                                                         p->beta1, p->beta2, p->eps, p->weight_decay, p->momentum, p->lr_dev);
    MISEG_LAUNCH_CHECK("opt_step_pack_conv3");
    return MISEG_OK;
  });
  if (rc != MISEG_OK) return rc;
  if (p->count_n > 0) return opt_count_launch(p->used, p->steps, p->count_n, p->params_version, pack_state, s);
  return MISEG_OK;
}

static void wgrad_plan(int B, int D, int H, int W, int Cin, int Cout, int wbd, int* ncob, int* ncib, int* nsplit, int max_wg = 0) {
  *ncob = cdiv(Cout, WG_CB);
  *ncib = cdiv(Cin, WG_CB);
  const int nbricks = B * cdiv(D, wbd) * cdiv(H, BH) * cdiv(W, BW);
  int pairs = (*ncob) * (*ncib);
  int ns = 256 / pairs;
  if (max_wg > 0 && max_wg < 256) ns = max_wg / pairs;
  if (ns < 1) ns = 1;
  if (ns > nbricks) ns = nbricks;
  *nsplit = ns;
}

extern "C" size_t miseg_conv3_wgrad_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  int ncob, ncib, ns2, ns4;
  wgrad_plan(B, D, H, W, Cin, Cout, 2, &ncob, &ncib, &ns2);
  wgrad_plan(B, D, H, W, Cin, Cout, 4, &ncob, &ncib, &ns4);
  const int ns = ns2 > ns4 ? ns2 : ns4;
  size_t need = (size_t)ncob * ncib * ns * 27 * WG_CB * WG_CB * sizeof(float);
  if ((Cin == 16 || Cin == 32) && (Cout == 16 || Cout == 32)) {      // the narrow-layer kernel: one slab per workgroup
    const size_t nar = (size_t)WG_NARROW_MAX_WG * 27 * Cin * Cout * sizeof(float);
    if (nar > need) need = nar;
  }
  return need;
}

template <class T, int WBD>
static bool wgrad_piped(const miseg_conv3_wgrad_params* p, bool vec_x, bool vec_dy) {
  const int64_t span = (int64_t)(WBD + 2) * p->H * p->W * (p->ldx > p->lddy ? p->ldx : p->lddy) * (int64_t)sizeof(T);   // 32-bit item offsets
  // (round 5: the last brick along an axis may be partial; every axis holds at least one whole brick, so a brick's first interior voxel -
  // what an invalid item reads instead - is inside the volume)
  return std::is_same<T, bf16>::value && vec_x && vec_dy && p->Cin % WG_CB == 0 && p->Cout % WG_CB == 0 && p->D >= WBD && p->H >= BH && p->W >= BW &&
         span < ((int64_t)1 << 31);
}

template <class T, int WBD>
static int conv3_wgrad_launch(const miseg_conv3_wgrad_params* p, hipStream_t s) {
  constexpr int KPC = Vec16<T>::N;
  int ncob, ncib, nsplit;
  wgrad_plan(p->B, p->D, p->H, p->W, p->Cin, p->Cout, WBD, &ncob, &ncib, &nsplit, p->max_workgroups);
  ConvGeom g{p->B, p->D, p->H, p->W, cdiv(p->D, WBD), cdiv(p->H, BH), cdiv(p->W, BW)};
  const int rowb = WG_CB * (int)sizeof(T) + 16;
  const size_t lds = (size_t)((WBD + 2) * HH * HW + WBD * BH * BW) * rowb;
  const bool vec_x = ((uintptr_t)p->x % 16 == 0) && (p->ldx % KPC == 0);
  const bool vec_dy = ((uintptr_t)p->dy % 16 == 0) && (p->lddy % KPC == 0);
  const int direct = nsplit == 1 ? (p->accumulate == 1 ? 1 : 2) : 0;     // a single producer per element: no slabs, no second launch
  dim3 grid(nsplit, ncob * ncib);
  if (wgrad_piped<T, WBD>(p, vec_x, vec_dy)) {
    hipFuncSetAttribute((const void*)conv3_wgrad_kernel<T, WBD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    conv3_wgrad_kernel<T, WBD, true><<<grid, WG_THREADS, lds, s>>>((const T*)p->x, p->ldx, (const T*)p->dy, p->lddy, (float*)p->workspace, p->dw, g, p->Cin, p->Cout,
                                                                   ncib, nsplit, rowb, vec_x, vec_dy, direct);
  } else {
    hipFuncSetAttribute((const void*)conv3_wgrad_kernel<T, WBD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    conv3_wgrad_kernel<T, WBD, false><<<grid, WG_THREADS, lds, s>>>((const T*)p->x, p->ldx, (const T*)p->dy, p->lddy, (float*)p->workspace, p->dw, g, p->Cin, p->Cout,
                                                                    ncib, nsplit, rowb, vec_x, vec_dy, direct);
  }
  if (!direct) {
    const int64_t total = (int64_t)p->Cout * p->Cin * 27;
    if (!p->accumulate) MISEG_REQUIRE(fill_words_async(p->dw, 0, (size_t)total, s) == hipSuccess, MISEG_E_LAUNCH, "conv3_wgrad: fill");
    const int tiles = p->Cout * ncib;
    int groups = cdiv(2048, tiles);
    if (groups > nsplit) groups = nsplit;
    const int spg = cdiv(nsplit, groups);
    groups = cdiv(nsplit, spg);
    conv3_wgrad_reduce_kernel<<<dim3(tiles, groups), 256, 0, s>>>((const float*)p->workspace, p->dw, p->Cin, p->Cout, ncib, nsplit, spg);
  }
  MISEG_LAUNCH_CHECK("conv3_wgrad");
  return MISEG_OK;
}

// grouped launch: every layer gets about `target` bricks per workgroup (the group as a whole fills the chip, so a layer no longer needs 256
// workgroups of its own and the slab traffic shrinks with the split count).  Round 4: the target follows from the group - the brick x
// channel-pair units of all its multi-brick layers, dealt to the CUs in ONE round (main group of a C-Swin-UNETR step: 3336 units -> 14 bricks
// each = 256 long workgroups, and the 12^3 layers become single-split: direct epilogue, no slabs; measured 437 -> 358 us against the fixed 7
// of rounds 1-3, scripts/micro/wgrad_group_bench.py).
static int wgrad_group_target(const miseg_conv3_wgrad_params* descs, int n, int wbd) {
  long units = 0;
  for (int i = 0; i < n; ++i) {
    const miseg_conv3_wgrad_params* p = descs + i;
    const int nbricks = p->B * cdiv(p->D, wbd) * cdiv(p->H, BH) * cdiv(p->W, BW);
    if (nbricks > 2) units += (long)cdiv(p->Cout, WG_CB) * cdiv(p->Cin, WG_CB) * nbricks;
  }
  const int cap = descs[0].max_workgroups;
  const int slots = (cap > 0 && cap < 256) ? cap : 256;
  int t = cdiv(units, slots);
  const int tmin = 7;
  if (t < tmin) t = tmin;    // (small groups - the side branch's two 48^3 layers, 864 units: 4 / 5 bricks per unit measured no better than 7 in the step)
  if (t > 16) t = 16;
  return t;
}

static void wgrad_group_plan(const miseg_conv3_wgrad_params* p, int wbd, int target, int* ncob, int* ncib, int* nsplit) {
  *ncob = cdiv(p->Cout, WG_CB);
  *ncib = cdiv(p->Cin, WG_CB);
  const int nbricks = p->B * cdiv(p->D, wbd) * cdiv(p->H, BH) * cdiv(p->W, BW);
  int ns = cdiv(nbricks, target);
  if (ns >= 12 && ns + 7 <= nbricks) ns = (ns + 7) / 8 * 8;       // a multiple of 8: the XCD-contiguous brick order applies
  *nsplit = ns;
}

static size_t wgrad_group_slab_floats(const miseg_conv3_wgrad_params* p, int wbd, int target) {
  int ncob, ncib, nsplit;
  wgrad_group_plan(p, wbd, target, &ncob, &ncib, &nsplit);
  return nsplit == 1 ? 0 : (size_t)ncob * ncib * nsplit * 27 * WG_CB * WG_CB;
}

extern "C" size_t miseg_conv3_wgrad_group_workspace_bytes(const miseg_conv3_wgrad_params* descs, int n) {
  if (!descs || n <= 0) return 0;
  const int wbd = descs[0].dtype == MISEG_BF16 ? 4 : 2;
  const int target = wgrad_group_target(descs, n, wbd);
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += wgrad_group_slab_floats(descs + i, wbd, target);
  return total * sizeof(float);
}

template <class T, int WBD>
static int conv3_wgrad_group_launch(const miseg_conv3_wgrad_params* descs, int n, float* workspace, hipStream_t s) {
  constexpr int KPC = Vec16<T>::N;
  const int rowb = WG_CB * (int)sizeof(T) + 16;
  const size_t lds = (size_t)((WBD + 2) * HH * HW + WBD * BH * BW) * rowb;
  const int target = wgrad_group_target(descs, n, WBD);
  // longest workgroups first
  int order[WG_GROUP_MAX];
  int weight[WG_GROUP_MAX];
  for (int i = 0; i < n; ++i) {
    int ncob, ncib, nsplit;
    wgrad_group_plan(descs + i, WBD, target, &ncob, &ncib, &nsplit);
    const int nbricks = descs[i].B * cdiv(descs[i].D, WBD) * cdiv(descs[i].H, BH) * cdiv(descs[i].W, BW);
    order[i] = i;
    weight[i] = cdiv(nbricks, nsplit) * 8 - (descs[i].D < WBD ? 4 : 0) - (descs[i].H <= 4 ? 2 : 0);
  }
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && weight[order[j]] > weight[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  WgradGroup grp;
  grp.n = n;
  int wg = 0, rb = 0;
  size_t off = 0;
  for (int k = 0; k < n; ++k) {
    const miseg_conv3_wgrad_params* p = descs + order[k];
    WgradLayer& L = grp.l[k];
    int ncob, ncib, nsplit;
    wgrad_group_plan(p, WBD, target, &ncob, &ncib, &nsplit);
    L.x = p->x; L.dy = p->dy; L.dw = p->dw; L.ldx = p->ldx; L.lddy = p->lddy;
    L.g = ConvGeom{p->B, p->D, p->H, p->W, cdiv(p->D, WBD), cdiv(p->H, BH), cdiv(p->W, BW)};
    L.Cin = p->Cin; L.Cout = p->Cout; L.ncib = ncib; L.nsplit = nsplit;
    const int direct = nsplit == 1 ? (p->accumulate == 1 ? 1 : 2) : 0;
    const bool vec_x = ((uintptr_t)p->x % 16 == 0) && (p->ldx % KPC == 0);
    const bool vec_dy = ((uintptr_t)p->dy % 16 == 0) && (p->lddy % KPC == 0);
    L.flags = (vec_x ? 1 : 0) | (vec_dy ? 2 : 0) | (direct << 2) | (wgrad_piped<T, WBD>(p, vec_x, vec_dy) ? 16 : 0);
    L.wg0 = wg;
    wg += ncob * ncib * nsplit;
    L.slabs = workspace + off;
    L.tiles = 0; L.groups = 0; L.spg = 0; L.rb0 = rb;
    if (!direct) {
      off += (size_t)ncob * ncib * nsplit * 27 * WG_CB * WG_CB;
      if (!p->accumulate)
        MISEG_REQUIRE(fill_words_async(p->dw, 0, (size_t)p->Cout * p->Cin * 27, s) == hipSuccess, MISEG_E_LAUNCH, "conv3_wgrad_group: fill");
      L.tiles = p->Cout * ncib;
      int groups = cdiv(512, L.tiles);     // the layers of the group fill the chip together: few split groups each (fewer atomics)
      if (groups > nsplit) groups = nsplit;
      L.spg = cdiv(nsplit, groups);
      L.groups = cdiv(nsplit, L.spg);
      rb += L.tiles * L.groups;
    }
  }
  const int cap = descs[0].max_workgroups;      // > 0: background form (see the kernel)
  // the pipelined layers and the others: one grid each (same descriptors, workgroups renumbered; the reduce launch walks `grp`)
  for (int form = 0; form < 2; ++form) {
    WgradGroup sub;
    sub.n = 0;
    int nwg = 0;
    for (int k = 0; k < n; ++k) {
      if ((((grp.l[k].flags >> 4) & 1) != 0) != (form == 0)) continue;
      WgradLayer& L = sub.l[sub.n++];
      L = grp.l[k];
      const int units = (k + 1 < n ? grp.l[k + 1].wg0 : wg) - grp.l[k].wg0;
      L.wg0 = nwg;
      nwg += units;
    }
    if (!sub.n) continue;
    const bool walk = cap > 0 && cap < nwg;
    const int grid = walk ? cap : nwg;
#define MISEG_WG_GROUP_LAUNCH(PIPED_, WALK_)                                                                                                       \
    do {                                                                                                                                           \
      hipFuncSetAttribute((const void*)conv3_wgrad_group_kernel<T, WBD, PIPED_, WALK_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
      conv3_wgrad_group_kernel<T, WBD, PIPED_, WALK_><<<grid, WG_THREADS, lds, s>>>(sub, rowb, nwg);                                                \
    } while (0)
    if (form == 0) {
      if constexpr (std::is_same<T, bf16>::value) {
        if (walk) MISEG_WG_GROUP_LAUNCH(true, true); else MISEG_WG_GROUP_LAUNCH(true, false);
      }
    } else {
      if (walk) MISEG_WG_GROUP_LAUNCH(false, true); else MISEG_WG_GROUP_LAUNCH(false, false);
    }
#undef MISEG_WG_GROUP_LAUNCH
  }
  if (rb > 0) conv3_wgrad_reduce_group_kernel<<<rb, 256, 0, s>>>(grp);
  MISEG_LAUNCH_CHECK("conv3_wgrad_group");
  return MISEG_OK;
}

#ifdef MISEG_WGRAD_STAMPS
extern "C" int miseg_debug_wgrad_stamps(unsigned long long* out) {   // read and reset
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(miseg::miseg_wg_stamps), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(miseg::miseg_wg_stamps), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int miseg_conv3_wgrad_tiny(int B, int D, int H, int W, int Cin, int Cout, int dtype) { return wgrad_tiny_shape(B, D, H, W, Cin, Cout, dtype) ? 1 : 0; }

extern "C" int miseg_conv3_wgrad_group(const miseg_conv3_wgrad_params* descs, int n, void* workspace, miseg_stream_t s_) {
  MISEG_REQUIRE(descs && n > 0 && n <= WG_GROUP_MAX, MISEG_E_BADARG, "conv3_wgrad_group: 1..%d layers per launch", WG_GROUP_MAX);
  for (int i = 0; i < n; ++i) {
    const miseg_conv3_wgrad_params* p = descs + i;
    MISEG_REQUIRE(p->x && p->dy && p->dw, MISEG_E_BADARG, "conv3_wgrad_group: null pointer in layer %d", i);
    MISEG_REQUIRE(p->B > 0 && p->D > 0 && p->H > 0 && p->W > 0 && p->Cin > 0 && p->Cout > 0, MISEG_E_BADARG, "conv3_wgrad_group: bad shape in layer %d", i);
    MISEG_REQUIRE(p->dtype == descs[0].dtype, MISEG_E_BADARG, "conv3_wgrad_group: mixed dtypes");
  }
  MISEG_REQUIRE(workspace || miseg_conv3_wgrad_group_workspace_bytes(descs, n) == 0, MISEG_E_BADARG, "conv3_wgrad_group: workspace missing");
  if (descs[0].dtype == MISEG_F32) return conv3_wgrad_group_launch<float, 2>(descs, n, (float*)workspace, (hipStream_t)s_);
  if (descs[0].dtype == MISEG_BF16) return conv3_wgrad_group_launch<bf16, 4>(descs, n, (float*)workspace, (hipStream_t)s_);
  return set_error(MISEG_E_BADARG, "conv3_wgrad_group: dtype %d", descs[0].dtype);
}

extern "C" int miseg_conv3_wgrad(const miseg_conv3_wgrad_params* p, miseg_stream_t s_) {
  MISEG_REQUIRE(p && p->x && p->dy && p->dw && p->workspace, MISEG_E_BADARG, "conv3_wgrad: null pointer");
  MISEG_REQUIRE(p->B > 0 && p->D > 0 && p->H > 0 && p->W > 0 && p->Cin > 0 && p->Cout > 0, MISEG_E_BADARG, "conv3_wgrad: bad shape");
  if (p->dtype == MISEG_F32) return conv3_wgrad_launch<float, 2>(p, (hipStream_t)s_);
  if (p->dtype == MISEG_BF16 && wgrad_narrow(p)) return conv3_wgrad_narrow_launch(p, (hipStream_t)s_);
  if (wgrad_tiny(p)) return conv3_wgrad_tiny_launch(p, (hipStream_t)s_);
  if (p->dtype == MISEG_BF16) return conv3_wgrad_launch<bf16, 4>(p, (hipStream_t)s_);
  return set_error(MISEG_E_BADARG, "conv3_wgrad: dtype %d", p->dtype);
}
