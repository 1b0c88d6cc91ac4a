// What does s_memtime count, and how fast does the shader clock run under a dense MFMA load?
// Each wave records s_memtime (the counter the in-kernel phase stamps use) and s_memrealtime (constant 100 MHz) around a long loop of
// dependent-free MFMAs; ticks(memtime) / ticks(realtime) * 100 MHz = frequency of the s_memtime counter; the loop's MFMA count / elapsed
// realtime = achieved MFMA rate, which against 16 cycles per v_mfma_f32_16x16x32_bf16 gives the shader clock under that load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void __launch_bounds__(256) probe(unsigned long long* out, int iters, int idle) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  if (idle) {
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(64);
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (threadIdx.x % 64 == 0) {
    const int w = (blockIdx.x * 256 + threadIdx.x) / 64;
    out[2 * w] = t1 - t0;
    out[2 * w + 1] = r1 - r0 + (s == 12345.f ? 1 : 0);
  }
}
int main() {
  const int blocks = 256 * 2, waves = blocks * 4;
  unsigned long long* d;
  hipMalloc(&d, waves * 16);
  unsigned long long* h = new unsigned long long[2 * waves];
  for (int idle = 1; idle >= 0; --idle) {
    const int iters = idle ? 20000 : 100000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      probe<<<blocks, 256>>>(d, iters, idle);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("[host events: kernel %.1f us] ", ms * 1e3);
    hipMemcpy(h, d, waves * 16, hipMemcpyDeviceToHost);
    double mt = 0, rt = 0;
    for (int w = 0; w < waves; ++w) { mt += h[2 * w]; rt += h[2 * w + 1]; }
    mt /= waves; rt /= waves;
    const double us = rt / 100.0;      // 100 MHz
    printf("%s: s_memtime ticks %.0f, s_memrealtime ticks %.0f (%.1f us): s_memtime runs at %.1f MHz", idle ? "idle (s_sleep)" : "MFMA load  ", mt, rt, us, mt / us);
    if (!idle) {
      const double mfma_per_wave = 8.0 * iters, cyc = 16.0 * mfma_per_wave * 2;      // 2 waves per SIMD share the pipe
      printf("; %.0f MFMAs per wave in %.1f us -> shader clock >= %.0f MHz if the pipe never idled; chip rate %.0f TFLOP/s", mfma_per_wave, us, cyc / us,
             waves * mfma_per_wave * 16384.0 / us / 1e6);
    }
    printf("\n");
  }
  return 0;
}
