// micro-benchmark: sustained dense bf16 MFMA rate of the chip (registers only) and the shader clock under that load
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = f32x4{0.f + i, 1.f, 2.f, 3.f * threadIdx.x}; asm volatile("" : "+v"(acc[i])); }
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int blocks_per_cu, float* d, unsigned long long* c) {
  const int iters = 20000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(d, c, 100); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(d, c, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hc[8]; (void)hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
  const double flops = 2.0 * 16 * 16 * 32 * (double)NACC * iters * blocks * 4;
  printf("NACC %2d blocks/CU %d: %7.3f ms  %7.1f TFLOP/s   block cycles %llu -> shader clock %.2f GHz, %.1f cycles per MFMA per SIMD\n", NACC, blocks_per_cu, ms,
         flops / ms / 1e9, hc[0], hc[0] / (ms * 1e6), (double)hc[0] / ((double)NACC * iters * blocks_per_cu));
}
int main() {
  float* d; unsigned long long* c; (void)hipMalloc(&d, 4096 * 256 * 4); (void)hipMalloc(&c, 4096 * 8);
  run<1>(1, d, c); run<2>(1, d, c); run<4>(1, d, c); run<8>(1, d, c); run<12>(1, d, c); run<12>(2, d, c); run<4>(2, d, c); run<4>(4, d, c);
  return 0;
}
