// loss_terms.h -- one term of MDiceLoss's forward sums (dice.py:130-166), shared by the stand-alone reduction (loss_optim.hip) and
// the head kernel that folds the sums into its logits store (head_tail_fused.hip).
#pragma once
#include "common.h"

namespace micf {

// One term of MDiceLoss (dice.py:130-166) : p = sigmoid(z) -> sum p t, sum p^2, sum t^2 and the BCE
// of the sigmoid output with its logs clamped at -100 (nn.BCELoss), as loss_optim.hip's dice_bce_partial_kernel -- here on the
// hardware exp / log / rcp (the libm forms are ~100 VALU instructions per term: 85 us of the whole chip's VALU at 128^3 x 8 classes x
// batch 2, more than the tail kernel itself; these are ~35).  p is the same value to 2-3 ulp; where 1 - p underflows the clamp
// takes over exactly as in the reference.
struct LossAcc {
  float a, b, c, d;
  __device__ __forceinline__ void term(float z, float t) {
    const float p = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
    a += p * t; b += p * p; c += t * t;
    const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.0f - p), -100.f);
    d -= t * lp + (1.0f - t) * lq;
  }
};


}  // namespace micf
