// The per-element optimiser update shared by opt_step_kernel (training.hip: element-wise over the flat arena) and
// opt_pack_conv3_kernel (conv3d.hip: the same update on the 16 x 16 x 27 tiles of the 3x3x3 conv weights, followed by their packs).
// lightning_monai.py:255-278: torch.optim.AdamW / Adam / SGD(nesterov=True) arithmetic, expression by expression.
#pragma once
#include "common.h"

namespace miseg {

struct OptHyper {
  int kind, step;                      // MISEG_OPT_*; the update count of this parameter INCLUDING this step (Adam bias correction)
  float lr, b1, b2, eps, wd, mom;
  float bc2s, step_size;               // sqrt(1 - b2^step), lr / (1 - b1^step)
};

__device__ __forceinline__ OptHyper opt_hyper(int kind, int step, float lr, float b1, float b2, float eps, float wd, float mom) {
  OptHyper h{kind, step, lr, b1, b2, eps, wd, mom, 1.f, lr};
  if (kind != MISEG_OPT_SGD_NESTEROV) {
    const float bc1 = 1.f - powf(b1, (float)step);
    h.bc2s = sqrtf(1.f - powf(b2, (float)step));
    h.step_size = lr / bc1;
  }
  return h;
}

// w: parameter, g: gradient, m / v: the two state slots (v unused by SGD); updated in place
__device__ __forceinline__ void opt_update(const OptHyper& h, float& w, float g, float& m, float& v) {
  if (h.kind == MISEG_OPT_ADAMW) {
    w *= 1.f - h.lr * h.wd;
  } else {
    g += h.wd * w;                                  // Adam / SGD: L2 term folded into the gradient
  }
  if (h.kind == MISEG_OPT_SGD_NESTEROV) {
    const float buf = h.step == 1 ? g : h.mom * m + g;      // torch.optim.SGD: the first momentum buffer is the gradient itself
    m = buf;
    w -= h.lr * (g + h.mom * buf);
  } else {
    m = m + (1.f - h.b1) * (g - m);               // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = h.b2 * v + (1.f - h.b2) * g * g;
    const float denom = sqrtf(v) / h.bc2s + h.eps;
    w -= h.step_size * (m / denom);
  }
}

// steps[i] += used[i], parameter version += 1 (training.hip); `pack_state` (optional): state[0] of a versioned pack table whose packs were
// written from the new parameters by the launch in front of this one - it then holds the new version and the next refresh launch finds it current
int opt_count_launch(const int32_t* used, int32_t* steps, int n, int64_t* params_version, int64_t* pack_state, hipStream_t s);

}  // namespace miseg
