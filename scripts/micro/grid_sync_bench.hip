// What does an all-to-all seam cost INSIDE one launch on MI355X, against a kernel boundary in a hipGraph?  (DESIGN R5: pricing the
// "whole-block persistent kernel" of the deep Swin stages before building it.)
//   build: hipcc -O3 --offload-arch=gfx950 -o grid_sync_bench grid_sync_bench.hip ; run: ./grid_sync_bench
// A chain of P dependent phases over G workgroups of 256 threads.  In phase p every workgroup reads ALL G chunks the previous phase wrote
// (an all-to-all dependency, like a GEMM whose A operand the other workgroups produced), and writes its own chunk.
//   mode 0  "graph"   : one kernel launch per phase, the chain captured in a hipGraph (what the step does today)
//   mode 1  "fence"   : one launch, plain stores + agent release fence + arrival counter + sc1 poll + agent acquire fence
//   mode 2  "sc1"     : one launch, write-through (sc1) stores, vmcnt(0) drain, arrival counter, sc1 poll, sc1 loads (no fences)
//   mode 3  "one-xcd" : as mode 2 but only the workgroups that find themselves on ONE XCD take part (8 G launched; XCC id read from
//                       the hardware register, participation counted - no placement is assumed), plain stores + sc1 loads: the XCD's L2 is
//                       the exchange
// The result of every mode is compared with the host's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16_sc1(const uint4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16_sc1(uint4* p, uint4 w) {
    u32x4 v = {w.x, w.y, w.z, w.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// one phase of the chain: read all G chunks of `in`, write my chunk of `out`.  W = words per chunk (multiple of 1024: 256 threads x uint4)
template <int MODE>
__device__ __forceinline__ void phase_body(const unsigned* in, unsigned* out, int G, int W, int me, int p) {
    unsigned s = 0;
    const int n16 = G * W / 4;
    const uint4* in4 = (const uint4*)in;
    if (MODE >= 2) {       // eight write-through-coherent 16-byte loads in flight per lane, one wait
        for (int i = threadIdx.x; i < n16; i += 256 * 8) {
            u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            const uint4* q = in4 + i;
            asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
                         "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                         "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                         "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                         : "v"(q), "v"(q + 256), "v"(q + 512), "v"(q + 768), "v"(q + 1024), "v"(q + 1280), "v"(q + 1536), "v"(q + 1792) : "memory");
            u32x4 t = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
            s += t.x + t.y + t.z + t.w;
        }
    } else {
        for (int i = threadIdx.x; i < n16; i += 256) {
            uint4 v = in4[i];
            s += v.x + v.y + v.z + v.w;
        }
    }
    // workgroup sum
    __shared__ unsigned red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    s = red[0];
    __syncthreads();
    uint4* out4 = (uint4*)(out + (size_t)me * W);
    for (int i = threadIdx.x; i < W / 4; i += 256) {
        unsigned b = s * 2654435761u + me * 131u + p * 7u + i * 4;
        uint4 v = make_uint4(b, b + 1, b + 2, b + 3);
        if (MODE == 2) st16_sc1(out4 + i, v); else out4[i] = v;
    }
}

__global__ void phase_kernel(const unsigned* in, unsigned* out, int G, int W, int p) { phase_body<0>(in, out, G, W, blockIdx.x, p); }

// counter barrier: every workgroup adds 1, waits for target
template <int MODE>
__device__ __forceinline__ bool barrier(unsigned* cnt, unsigned target, unsigned* tmo) {
    if (MODE == 1) { __syncthreads(); if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_sc1(cnt) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > 4000000u) { st_sc1(tmo, 1u); ok = false; break; } }
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok;
}

template <int MODE>
__global__ void chain_kernel(unsigned* buf0, unsigned* buf1, int G, int W, int P, unsigned* state /* [0] counter, [1] timeout, [2] home, [3] home count, [4] total, [5..] */) {
    int me = blockIdx.x;
    if (MODE == 3) {
        __shared__ int s_me;
        if (threadIdx.x == 0) {
            unsigned x = xcc_id();
            unsigned old = atomicCAS(&state[2], 0xffffffffu, x);
            unsigned home = old == 0xffffffffu ? x : old;
            int r = -1;
            if (x == home) r = (int)atomicAdd(&state[3], 1u);
            atomicAdd(&state[4], 1u);
            s_me = (r >= 0 && r < G) ? r : -1;
        }
        __syncthreads();
        me = s_me;
        if (me < 0) return;
        // (a production kernel would wait for state[4] == gridDim.x and use the actual count; here the host checks state[3] >= G)
    }
    unsigned* bufs[2] = {buf0, buf1};
    for (int p = 0; p < P; ++p) {
        phase_body<MODE>(bufs[p & 1], bufs[(p + 1) & 1], G, W, me, p);
        if (!barrier<MODE>(&state[0], (unsigned)(G * (p + 1)), &state[1])) return;
    }
}

static void host_chain(std::vector<unsigned>& a, std::vector<unsigned>& b, int G, int W, int P) {
    std::vector<unsigned>* bufs[2] = {&a, &b};
    for (int p = 0; p < P; ++p) {
        auto& in = *bufs[p & 1]; auto& out = *bufs[(p + 1) & 1];
        unsigned s = 0;
        for (int i = 0; i < G * W; ++i) s += in[i];
        for (int me = 0; me < G; ++me)
            for (int i = 0; i < W / 4; ++i) {
                unsigned v = s * 2654435761u + me * 131u + p * 7u + i * 4;
                for (int k = 0; k < 4; ++k) out[(size_t)me * W + i * 4 + k] = v + k;
            }
    }
}

int main() {
    const int P = 40;
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned* state; CK(hipMalloc(&state, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"graph (kernel per phase)", "one launch, plain + agent fences", "one launch, sc1 stores/loads", "one launch, one XCD (L2 exchange)"};
    printf("%d dependent all-to-all phases; us per phase\n", P);
    for (int W : {1024, 4096}) for (int G : {8, 16, 32, 64, 128}) {
        size_t words = (size_t)G * W;
        unsigned *b0, *b1; CK(hipMalloc(&b0, words * 4)); CK(hipMalloc(&b1, words * 4));
        std::vector<unsigned> h0(words), h1(words, 0), ref0, ref1;
        for (size_t i = 0; i < words; ++i) h0[i] = (unsigned)(i * 2246822519u + 12345u);
        ref0 = h0; ref1 = h1; host_chain(ref0, ref1, G, W, P);
        std::vector<unsigned>& ref = (P & 1) ? ref1 : ref0;
        printf("G %3d workgroups, %2d KB per workgroup (%4d KB read per workgroup and phase):", G, W * 4 / 1024, G * W * 4 / 1024);
        for (int mode = 0; mode < 4; ++mode) {
            if (mode == 3 && G > 32) { printf("  [3] -"); continue; }
            float best = 1e30f; bool good = true; unsigned hc = 0;
            hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
            if (mode == 0) {
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int p = 0; p < P; ++p) phase_kernel<<<G, 256, 0, st>>>((p & 1) ? b1 : b0, (p & 1) ? b0 : b1, G, W, p);
                CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            }
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipMemcpyAsync(b0, h0.data(), words * 4, hipMemcpyHostToDevice, st));
                CK(hipMemsetAsync(b1, 0, words * 4, st));
                unsigned init[16] = {0, 0, 0xffffffffu, 0, 0};
                CK(hipMemcpyAsync(state, init, 64, hipMemcpyHostToDevice, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                if (mode == 0) CK(hipGraphLaunch(exec, st));
                else if (mode == 1) chain_kernel<1><<<G, 256, 0, st>>>(b0, b1, G, W, P, state);
                else if (mode == 2) chain_kernel<2><<<G, 256, 0, st>>>(b0, b1, G, W, P, state);
                else chain_kernel<3><<<8 * G, 256, 0, st>>>(b0, b1, G, W, P, state);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
                std::vector<unsigned> got(words);
                CK(hipMemcpy(got.data(), (P & 1) ? b1 : b0, words * 4, hipMemcpyDeviceToHost));
                unsigned hs[16]; CK(hipMemcpy(hs, state, 64, hipMemcpyDeviceToHost));
                if (hs[1]) good = false;
                if (memcmp(got.data(), ref.data(), words * 4) != 0) good = false;
                hc = hs[3];
            }
            printf("  [%d] %6.2f%s", mode, best * 1e3f / P, good ? "" : " WRONG");
            if (mode == 3) printf(" (home count %u)", hc);
            if (exec) { CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); }
        }
        printf("\n");
        CK(hipFree(b0)); CK(hipFree(b1));
    }
    for (int m = 0; m < 4; ++m) printf("[%d] %s\n", m, names[m]);
    return 0;
}
