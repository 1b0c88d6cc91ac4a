// micro-benchmark: cost of LDS atomic adds per wave-instruction (float vs u32 vs u64, address patterns)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, int PAT>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ float tab[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, fi = lane & 15, kg = lane >> 4;
  int idx;
  if (PAT == 0) idx = lane;                           // distinct, conflict-free
  else if (PAT == 1) idx = (lane * 37) & 2047;        // distinct, scattered
  else if (PAT == 2) idx = 1100 + 4 * kg - fi;        // toeplitz: 4-way same address
  else idx = 7;                                       // all same address
  float v = 1.0f + lane * 1e-3f;
  unsigned long long* t64 = reinterpret_cast<unsigned long long*>(tab);
  unsigned* t32 = reinterpret_cast<unsigned*>(tab);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int a = (idx + u * 64 + (it & 7) * 13) & 2047;
      if (MODE == 0) atomicAdd(&tab[a], v);
      else if (MODE == 1) atomicAdd(&t32[a], (unsigned)(v * 1024.f));
      else if (MODE == 2) atomicAdd(&t64[a], (unsigned long long)(v * 1024.f));
      else if (MODE == 3) tab[a] = v;                 // plain store baseline
      else if (MODE == 4) v += tab[a];                // plain load baseline
    }
  }
  __syncthreads();
  float s = v;
  for (int i = threadIdx.x; i < 4096; i += 256) s += tab[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int PAT>
void run(const char* name, float* d) {
  const int iters = 2000, blocks = 256 * 2;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, PAT><<<blocks, 256>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE, PAT><<<blocks, 256>>>(d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: 2 blocks x 4 waves x iters*8 wave-instrs
  const double instr_per_cu = 2.0 * 4 * iters * 8;
  printf("%-28s %8.3f ms  %7.1f cycles per wave-instr per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}
int main() {
  float* d; hipMalloc(&d, 512 * 256 * 4);
  run<0, 0>("f32 add  distinct", d); run<0, 1>("f32 add  scattered", d); run<0, 2>("f32 add  toeplitz4", d); run<0, 3>("f32 add  same", d);
  run<1, 0>("u32 add  distinct", d); run<1, 1>("u32 add  scattered", d); run<1, 2>("u32 add  toeplitz4", d); run<1, 3>("u32 add  same", d);
  run<2, 0>("u64 add  distinct", d); run<2, 1>("u64 add  scattered", d); run<2, 2>("u64 add  toeplitz4", d);
  run<3, 0>("store    distinct", d); run<3, 1>("store    scattered", d);
  run<4, 0>("load     distinct", d); run<4, 1>("load     scattered", d);
  return 0;
}
