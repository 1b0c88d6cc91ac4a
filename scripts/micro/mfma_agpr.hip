// micro-benchmark: one wave per SIMD, 63 accumulator tiles pinned to AGPRs, operands with random bf16 bit patterns vs small regular
// values: is the weight-gradient k-loop's ~11.5 ns per MFMA the matrix pipe under realistic data (clock / power) or the code around it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void __launch_bounds__(256) k(const bf16x8* __restrict__ src, float* out, int iters) {
  f32x4 acc[63];
#pragma unroll
  for (int i = 0; i < 63; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[3], b[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { a[i] = src[(threadIdx.x * 6 + i) & 4095]; b[i] = src[(threadIdx.x * 6 + 3 + i) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
      for (int i = 0; i < 9; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[t * 9 + i]) : "v"(b[i % 3]), "v"(a[i / 3]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 63; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; bf16x8* src; (void)hipMalloc(&d, 1024 * 256 * 4); (void)hipMalloc(&src, 4096 * 16);
  unsigned short h[4096 * 8];
  for (int mode = 0; mode < 3; ++mode) {
    for (int i = 0; i < 4096 * 8; ++i) {
      float v = mode == 0 ? 0.f : mode == 1 ? 0.001f * (i % 64) : (float)((rand() % 2001) - 1000) / 500.f;     // zeros / small regular / N(0,1)-like
      unsigned int u; memcpy(&u, &v, 4); h[i] = (unsigned short)(u >> 16);
    }
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 256>>>(src, d, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<256, 256>>>(src, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("data mode %d (0 zeros, 1 small regular, 2 random): %.3f ms, %.2f ns per MFMA per SIMD, %.0f TFLOP/s\n", mode, ms, ms * 1e6 / (63.0 * iters),
           2.0 * 16 * 16 * 32 * 63.0 * iters * 1024 / ms / 1e9);
  }
  return 0;
}
