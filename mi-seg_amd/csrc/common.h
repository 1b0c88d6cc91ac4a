// Shared device/host helpers for libmiseg_hip.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
#include "../../include/miseg_hip.h"

namespace miseg {

static constexpr int MAX_DEVICES = 64;   // device ordinals with a cached kernel attribute (MISEG_SET_SMEM); others set it on every launch

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

int set_error(int code, const char* fmt, ...);

#define MISEG_REQUIRE(cond, code, ...)                      \
  do {                                                      \
    if (!(cond)) return miseg::set_error((code), __VA_ARGS__); \
  } while (0)

#define MISEG_LAUNCH_CHECK(name)                                                        \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess)                                                              \
      return miseg::set_error(MISEG_E_LAUNCH, "%s: %s", (name), hipGetErrorString(e__)); \
  } while (0)

// 16-byte vector of T (the unit every HBM / LDS access moves per lane)
template <class T> struct Vec16;
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };
template <> struct Vec16<bf16>  { typedef bf16x8 type; static constexpr int N = 8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <class T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__host__ __device__ static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// grid of a persistent tile loop: at most `cap` workgroups, and as few as walk the tiles in the same number of rounds (3456 tiles under a cap
// of 2048 are two rounds either way: 1728 workgroups take two tiles each, 2048 leave 640 of them idle through the second round)
static inline int balanced_grid(long ntiles, int cap) {
  if (ntiles <= cap) return (int)(ntiles > 0 ? ntiles : 1);
  const long rounds = (ntiles + cap - 1) / cap;
  return (int)((ntiles + rounds - 1) / rounds);
}

// Word fills are plain kernels, never hipMemset*Async: under stream capture a memset becomes a memset node, and the ROCm 7.2
// graph runtime was observed to replay such nodes with a stale argument block (the statistics pool came back "zeroed" with
// the arguments of an unrelated later kernel), which silently corrupts every replay after the first.
static __global__ void __launch_bounds__(256) fill_words_kernel(uint32_t* __restrict__ dst, uint32_t v, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const size_t n4 = n >> 2;
    const u32x4 vv = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) reinterpret_cast<u32x4*>(dst)[i] = vv;
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
  }
}

static __global__ void __launch_bounds__(256) fill_words_2d_kernel(uint32_t* __restrict__ dst, size_t pitch, uint32_t v, size_t width, size_t rows) {
  const size_t total = width * rows;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / width;
    dst[r * pitch + (i - r * width)] = v;
  }
}

// n, pitch, width in 32-bit words
static inline hipError_t fill_words_async(void* dst, uint32_t v, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  fill_words_kernel<<<(int)blocks, 256, 0, s>>>((uint32_t*)dst, v, n);
  return hipGetLastError();
}

static inline hipError_t fill_words_2d_async(void* dst, size_t pitch, uint32_t v, size_t width, size_t rows, hipStream_t s) {
  if (width == 0 || rows == 0) return hipSuccess;
  if (pitch == width) return fill_words_async(dst, v, width * rows, s);
  size_t blocks = (width * rows + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  fill_words_2d_kernel<<<(int)blocks, 256, 0, s>>>((uint32_t*)dst, pitch, v, width, rows);
  return hipGetLastError();
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size) instead of on every launch: the attribute is sticky per
// device, the call is a driver round trip on the launch path.  One cache per call site (= per kernel instantiation), one slot per device
// ordinal; raising the size re-applies it.  Relaxed atomics: two threads racing on a slot both set the attribute, which is idempotent.
#define MISEG_SET_SMEM(fn, bytes)                                                                                  \
  do {                                                                                                             \
    static std::atomic<int> miseg_smem_set_[miseg::MAX_DEVICES];                                                   \
    int dev__ = 0;                                                                                                 \
    (void)hipGetDevice(&dev__);                                                                                    \
    const int slot__ = (dev__ >= 0 && dev__ < miseg::MAX_DEVICES) ? dev__ : -1;                                    \
    if (slot__ < 0 || (int)(bytes) + 1 > miseg_smem_set_[slot__].load(std::memory_order_relaxed)) {                \
      (void)hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));      \
      if (slot__ >= 0) miseg_smem_set_[slot__].store((int)(bytes) + 1, std::memory_order_relaxed);                 \
    }                                                                                                              \
  } while (0)

// Versioned refresh of parameter copies (miseg_hip.h, miseg_pack_conv3_batch): grid cap of the refresh kernels - a launch that finds its copies
// current costs the dispatch of this many empty workgroups - and the epilogue every workgroup of a working launch runs: the last one to arrive
// records the version the copies now hold (all of them read state[0] before their first tile, so nobody can see the new value in this launch).
static constexpr int REFRESH_MAX_WG = 2048;
// Round 4: the launch time of a WORKING refresh grows by ~34 ns per workgroup - every workgroup ends with one atomic on the same arrival counter
// (refresh_done), which the L2 serialises - so the grids are capped where the per-tile latency chain and that queue balance (live refresh of
// C-Swin-UNETR: casts 110 -> 81 us at 1024 workgroups, conv packs 237 -> 174 us at 512; 8192 workgroups: 324 / 438 us)
static constexpr int REFRESH_CAST_WG = 1024, REFRESH_PACK_WG = 512;
static inline int refresh_max_wg(const char*, int dflt) { return dflt; }      // (round 4 swept the caps through the environment; the library reads none any more)
__device__ __forceinline__ void refresh_done(const int64_t* params_version, int64_t* state, int64_t pv) {
  if (!params_version) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    // No __threadfence() in front of the arrival: the copies and the recorded version are read by LATER launches only, and the end of this
    // kernel publishes every workgroup's stores to them; the counter merely finds the workgroup that retires last.  (Round 5: the fence - a
    // write-back of the XCD's L2 per workgroup - was 21 of the 73 us of a live cast refresh of C-Swin-UNETR, scripts/debug/cast_table_probe.py;
    // what round 4 read as "34 ns per workgroup of serialised atomics" was this.)
    const unsigned long long prev = atomicAdd(reinterpret_cast<unsigned long long*>(state + 1), 1ull);
    if (prev == (unsigned long long)gridDim.x - 1) {
      state[0] = pv;
      state[1] = 0;
    }
  }
}

template <class F> static inline int dispatch_dtype(int dtype, F&& f) {
  if (dtype == MISEG_F32) return f((float*)nullptr);
  if (dtype == MISEG_BF16) return f((bf16*)nullptr);
  return set_error(MISEG_E_BADARG, "unknown dtype %d", dtype);
}

// counter-based dropout (training.hip: miseg_dropout; attention.hip: attention-probability dropout): splitmix64 finaliser, nothing stored
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t dropout_host_key(uint64_t seed, uint64_t stream_id) { return seed * 0x9E3779B97F4A7C15ull + stream_id * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull; }
__device__ __forceinline__ uint64_t dropout_step_key(uint64_t key, const uint64_t* step_dev) { return mix64(key ^ (step_dev ? (*step_dev) * 0x9E3779B97F4A7C15ull : 0ull)); }
// element (row, c) of a [rows][C] matrix: one hash per group of 4 consecutive columns, four 16-bit draws against the 16-bit threshold
__device__ __forceinline__ uint64_t dropout_group_hash(uint64_t k, int64_t row, int cg, int c) { return mix64(k + (uint64_t)(row * cg + (c >> 2)) * 0xD1342543DE82EF95ull); }
__device__ __forceinline__ bool dropout_keeps(uint64_t h, int c, unsigned thresh) { return (unsigned)((h >> (16 * (c & 3))) & 0xffff) >= thresh; }

// dropout on attention probabilities (attn_drop): thresh == 0 = off.  The mask is miseg_dropout's over the [windows * heads * n][n] matrix
struct AttnDrop { unsigned thresh; float scale; uint64_t key; const uint64_t* step_dev; };

// norm.hip: second launch of a split convolution - y = round(sum of nslabs fp32 slabs [B * S][C] (+ res)), and the instance-norm statistics
// of y into `stat` (replicated fp64 layout of miseg_instnorm_stats; may be null)
int slabs_to_out_stats(const float* slabs, int nslabs, void* y, int64_t ldy, const void* res, int64_t ldres, int B, int S, int C, int dtype, double* stat,
                       hipStream_t stream);

}  // namespace miseg
