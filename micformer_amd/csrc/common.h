// common.h -- shared helpers for the C-ABI translation units of libmicformer_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/micformer_hip.h"
#include "gemm_core.h"

#define MICF_RETURN_LAUNCH()                          \
  do {                                                \
    return hipGetLastError() == hipSuccess ? MICF_OK : MICF_ELAUNCH; \
  } while (0)

namespace micf {

// token <-> (b, d, h, w) on a channels-last grid
struct Geo {
  int B, D, H, W;
  __host__ __device__ int64_t tokens() const { return (int64_t)B * D * H * W; }
  __device__ __forceinline__ void decode(int t, int& b, int& d, int& h, int& w) const {
    w = t % W; t /= W;
    h = t % H; t /= H;
    d = t % D; b = t / D;
  }
  __device__ __forceinline__ int token(int b, int d, int h, int w) const { return ((b * D + d) * H + h) * W + w; }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// column sums of (scale * dy) [M, N] accumulated into out[N] (bias gradients)
int colsum_atomic(const float* dy, const float* scale, int64_t rps, float* out, int64_t M, int N, hipStream_t s);

}  // namespace micf
