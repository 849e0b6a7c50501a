// layernorm.hip -- nn.LayerNorm over the channel dim of channels-last token rows (MS.py:308,321,461,468,540,569,987,988),
// one 64-lane wave per row, optional two-source rows (replaces torch.cat + norm2, MS.py:1033-1034).
#include "common.h"

namespace micf {

// rows per workgroup: 32 (4 waves x 8) amortises the dgamma/dbeta flush on big token grids; 4 (one row per wave) keeps
// all 256 CUs busy on the 8^3 / 4^3 stages
static inline int ln_rows_per_block(int64_t rows) { return rows >= 16384 ? 32 : 4; }
constexpr int kLnMaxC = 4096;

__device__ __forceinline__ float ln_fetch(const float* x1, const float* x2, int c1, int c2, int64_t row, int c) {
  return c < c1 ? x1[row * c1 + c] : x2[row * c2 + (c - c1)];
}

__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int c1,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows, int C, float eps, int rpb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c2 = C - c1;
  for (int k = 0; k < rpb / 4; ++k) {
    const int64_t row = (int64_t)blockIdx.x * rpb + k * 4 + wave;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ln_fetch(x1, x2, c1, c2, row, c);
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = ln_fetch(x1, x2, c1, c2, row, c) - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    for (int c = lane; c < C; c += 64)
      y[row * C + c] = (ln_fetch(x1, x2, c1, c2, row, c) - mu) * rs * gamma[c] + beta[c];
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
  }
}

__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* dy, const float* __restrict__ x1,
                                                     const float* __restrict__ x2, int c1, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     float* dx1, float* dx2,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows,
                                                     int C, const float* add, int rpb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [2][C] block partials of dgamma, dbeta
  float* sg = sm;
  float* sb = sm + C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c2 = C - c1;
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  for (int k = 0; k < rpb / 4; ++k) {
    const int64_t row = (int64_t)blockIdx.x * rpb + k * 4 + wave;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    float a = 0.f, b = 0.f;       // sum g*dy, sum g*dy*xhat
    for (int c = lane; c < C; c += 64) {
      const float xh = (ln_fetch(x1, x2, c1, c2, row, c) - mu) * rs;
      const float gd = gamma[c] * dy[row * C + c];
      a += gd; b += gd * xh;
    }
    a = wave_sum(a) / (float)C;
    b = wave_sum(b) / (float)C;
    for (int c = lane; c < C; c += 64) {
      const float xh = (ln_fetch(x1, x2, c1, c2, row, c) - mu) * rs;
      const float d = dy[row * C + c];
      const float v = rs * (gamma[c] * d - a - xh * b);
      float* dst = c < c1 ? dx1 + row * c1 + c : dx2 + row * c2 + (c - c1);
      *dst = add ? add[row * C + c] + v : v;
      atomicAdd(&sg[c], d * xh);
      atomicAdd(&sb[c], d);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    if (dgamma) atomicAdd(dgamma + c, sg[c]);
    if (dbeta) atomicAdd(dbeta + c, sb[c]);
  }
}

}  // namespace micf
using namespace micf;

extern "C" int micf_layernorm_fwd(const float* x1, const float* x2, int c1, const float* gamma, const float* beta,
                                  float* y, float* mean, float* rstd, int64_t rows, int C, float eps,
                                  micf_stream_t stream) {
  if (!x1 || !gamma || !beta || !y || rows < 0 || C <= 0 || c1 <= 0 || c1 > C || (c1 < C && !x2)) return MICF_EINVAL;
  if (rows == 0) return MICF_OK;
  const int rpb = ln_rows_per_block(rows);
  const int blocks = ceil_div(rows, rpb);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x1, x2 ? x2 : x1, c1, gamma, beta, y,
                     mean, rstd, rows, C, eps, rpb);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_layernorm_bwd(const float* dy, const float* x1, const float* x2, int c1, const float* mean,
                                  const float* rstd, const float* gamma, float* dx1, float* dx2, float* dgamma,
                                  float* dbeta, int64_t rows, int C, const float* add, micf_stream_t stream) {
  if (!dy || !x1 || !mean || !rstd || !gamma || !dx1 || rows < 0 || C <= 0 || c1 <= 0 || c1 > C ||
      (c1 < C && (!x2 || !dx2)))
    return MICF_EINVAL;
  if (C > kLnMaxC) return MICF_EUNSUPPORTED;
  if (rows == 0) return MICF_OK;
  const int rpb = ln_rows_per_block(rows);
  const int blocks = ceil_div(rows, rpb);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, dy, x1,
                     x2 ? x2 : x1, c1, mean, rstd, gamma, dx1, dx2 ? dx2 : dx1, dgamma, dbeta, rows, C, add, rpb);
  MICF_RETURN_LAUNCH();
}
