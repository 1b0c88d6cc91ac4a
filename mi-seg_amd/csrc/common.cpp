// Error reporting + library identity for libmiseg_hip.so
#include "common.h"
#include "../../include/miseg_hip_debug.h"
#include <string.h>

#ifndef MISEG_COMPILED_ARCH
#define MISEG_COMPILED_ARCH "gfx950"
#endif

namespace miseg {
static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace miseg

extern "C" int miseg_abi_version(void) { return MISEG_ABI_VERSION; }

// sizeof() of every params / descriptor struct of include/miseg_hip.h as compiled into this library (tests/test_abi.py and
// hip/lib.py::load compare them with the ctypes mirror)
extern "C" size_t miseg_abi_struct_size(const char* name) {
  if (!name) return 0;
#define MISEG_SZ(T) if (strcmp(name, #T) == 0) return sizeof(T);
  MISEG_SZ(miseg_instnorm_stats_params) MISEG_SZ(miseg_instnorm_apply_params) MISEG_SZ(miseg_instnorm_bwd_params) MISEG_SZ(miseg_instnorm_pair_bwd_params)
  MISEG_SZ(miseg_layernorm_fwd_params) MISEG_SZ(miseg_layernorm_bwd_params) MISEG_SZ(miseg_gemm_params) MISEG_SZ(miseg_tn_reduce_desc) MISEG_SZ(miseg_gemm_tn_desc)
  MISEG_SZ(miseg_mlp_params) MISEG_SZ(miseg_colsum_params) MISEG_SZ(miseg_colsum_desc) MISEG_SZ(miseg_conv3_params) MISEG_SZ(miseg_pack_conv3_params) MISEG_SZ(miseg_pack_conv3_desc)
  MISEG_SZ(miseg_conv3_wgrad_params) MISEG_SZ(miseg_winattn_params) MISEG_SZ(miseg_winattn_bwd_params) MISEG_SZ(miseg_add_params) MISEG_SZ(miseg_copy2d_params)
  MISEG_SZ(miseg_cast_params) MISEG_SZ(miseg_cast_desc) MISEG_SZ(miseg_gelu_fwd_params) MISEG_SZ(miseg_gelu_bwd_params) MISEG_SZ(miseg_s2c_params)
  MISEG_SZ(miseg_patch_embed_params) MISEG_SZ(miseg_patch_embed_bwd_params) MISEG_SZ(miseg_conv3_thin_params) MISEG_SZ(miseg_conv3_thin_wgrad_params)
  MISEG_SZ(miseg_resample2_params) MISEG_SZ(miseg_rowbias_params) MISEG_SZ(miseg_prelu_fwd_params) MISEG_SZ(miseg_prelu_bwd_params) MISEG_SZ(miseg_head_params)
  MISEG_SZ(miseg_head_bwd_params) MISEG_SZ(miseg_im2col3_params) MISEG_SZ(miseg_seg_loss_params) MISEG_SZ(miseg_dice_metric_params) MISEG_SZ(miseg_opt_desc) MISEG_SZ(miseg_opt_pack_map)
  MISEG_SZ(miseg_opt_step_params) MISEG_SZ(miseg_stitch_params) MISEG_SZ(miseg_aug_sample) MISEG_SZ(miseg_augment_params) MISEG_SZ(miseg_resample3d_params) MISEG_SZ(miseg_dropout_params)
  MISEG_SZ(miseg_affine2_params) MISEG_SZ(miseg_graph_split_info) MISEG_SZ(miseg_norm_ref)
#undef MISEG_SZ
  return 0;
}
extern "C" const char* miseg_last_error(void) { return miseg::g_err; }
extern "C" int miseg_device_arch(char* buf, size_t n) {
  if (!buf || n == 0) return MISEG_E_BADARG;
  strncpy(buf, MISEG_COMPILED_ARCH, n);      // the only code objects in this library (build.py: --offload-arch)
  buf[n - 1] = 0;
  return MISEG_OK;
}

extern "C" int miseg_device_check(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    (void)hipGetLastError();
    return miseg::set_error(MISEG_E_BADARG, "device_check: no HIP device %d", device);
  }
  if (strncmp(prop.gcnArchName, MISEG_COMPILED_ARCH, strlen(MISEG_COMPILED_ARCH)) != 0)
    return miseg::set_error(MISEG_E_UNSUPPORTED, "device %d is %s; this library holds %s code objects only", device, prop.gcnArchName, MISEG_COMPILED_ARCH);
  return MISEG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// In-situ kernel timing (bench.py's roofline leg; include/miseg_hip_debug.h).  ONLY the measurement build of the library,
// libmiseg_hip_prof.so, carries it: csrc/build.py compiles this file a second time with -DMISEG_PROF_WRAP and links that variant with
// -Wl,--wrap=hipLaunchKernel, so that every `<<<>>>` of every translation unit arrives at __wrap_hipLaunchKernel.  Disarmed, the call goes
// straight to the runtime.  Armed with a tag, each launch is issued through hipExtLaunchKernel with its own start / stop events: the runtime
// stamps them with the dispatch's begin / end timestamps (what rocprofv3 --kernel-trace reports), on the stream the kernel is launched on,
// ONCE, where the step issues it - beside whatever else is running.  Not capture-safe (events are created per launch): the roofline leg
// runs the step eagerly.  The product library, libmiseg_hip.so, is linked without the wrap and answers MISEG_E_UNSUPPORTED.
// ---------------------------------------------------------------------------------------------------------
#ifdef MISEG_PROF_WRAP
#include <mutex>
#include <vector>
extern "C" hipError_t __real_hipLaunchKernel(const void* f, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t s);
namespace {
struct ProfEntry { int tag; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
std::vector<ProfEntry> g_prof;
std::atomic<int> g_prof_tag{-1};
}  // namespace

extern "C" hipError_t __wrap_hipLaunchKernel(const void* f, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t s) {
  const int tag = g_prof_tag.load(std::memory_order_relaxed);
  if (tag < 0) return __real_hipLaunchKernel(f, grid, block, args, shmem, s);
  ProfEntry e{tag, nullptr, nullptr};
  if (hipEventCreate(&e.e0) != hipSuccess || hipEventCreate(&e.e1) != hipSuccess) return __real_hipLaunchKernel(f, grid, block, args, shmem, s);
  const hipError_t r = hipExtLaunchKernel(f, grid, block, args, shmem, s, e.e0, e.e1, 0);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(e);
  return r;
}

extern "C" int miseg_prof_arm(int tag) {
  g_prof_tag.store(tag < 0 ? -1 : tag, std::memory_order_relaxed);
  return MISEG_OK;
}

extern "C" int miseg_prof_read(int* tags, float* ms, int max) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  for (ProfEntry& e : g_prof) {
    float t = -1.f;
    if (hipEventSynchronize(e.e1) != hipSuccess || hipEventElapsedTime(&t, e.e0, e.e1) != hipSuccess) { (void)hipGetLastError(); t = -1.f; }
    if (n < max && tags && ms) { tags[n] = e.tag; ms[n] = t; }
    ++n;
    (void)hipEventDestroy(e.e0);
    (void)hipEventDestroy(e.e1);
  }
  g_prof.clear();
  return n;
}
extern "C" int miseg_prof_available(void) { return 1; }
#else
extern "C" int miseg_prof_arm(int tag) {
  if (tag < 0) return MISEG_OK;
  return miseg::set_error(MISEG_E_UNSUPPORTED, "prof_arm: this is the product library; in-place launch timing lives in libmiseg_hip_prof.so (csrc/build.py)");
}
extern "C" int miseg_prof_read(int*, float*, int) { return 0; }
extern "C" int miseg_prof_available(void) { return 0; }
#endif

// sha256 over the sources this library was compiled from (csrc/build.py: every csrc/*.hip, *.cpp, *.h and include/*.h, in name order);
// hip/lib.py::load() compares it with the sources it finds beside the library and refuses a stale build
#ifndef MISEG_SOURCE_DIGEST
#define MISEG_SOURCE_DIGEST "unknown"
#endif
extern "C" const char* miseg_source_digest(void) { return MISEG_SOURCE_DIGEST; }
