// linear.hip -- nn.Linear forward / data-gradient / weight-gradient on the fp32 MFMA GEMM core.
// Reference ops replaced: F.linear in (Cross)WindowAttention3D q/kv/proj (MS.py:188-201, 246-259),
// Mlp fc1/GELU/fc2 (MS.py:28-34), concat_back_dim on torch.cat (MS.py:1027-1030), and the
// residual + DropPath adds of the blocks (MS.py:419,424,517,522) fused into the epilogue.
#include "common.h"

namespace micf {

struct LinFwdEpi {
  const float* bias; const float* resid; const float* scale; int64_t rps;
  float* y; float* pre; int N; int act;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    if (bias) v += bias[j];
    const int64_t o = (int64_t)i * N + j;
    if (pre) pre[o] = v;
    if (act) v = gelu_f(v);
    if (resid) v = resid[o] + (scale ? scale[i / rps] : 1.f) * v;
    y[o] = v;
  }
};

struct LinBwdDataEpi {
  const float* pre; float* d1; float* d2; int k1, K, acc;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    if (pre) v *= gelu_grad_f(pre[(int64_t)i * K + j]);
    float* dst = (j < k1) ? d1 + (int64_t)i * k1 + j : d2 + (int64_t)i * (K - k1) + (j - k1);
    *dst = acc ? *dst + v : v;
  }
};

struct AtomicEpi {
  float* out; int64_t ld;
  __device__ __forceinline__ void operator()(int i, int j, float v) const { atomicAdd(out + (int64_t)i * ld + j, v); }
};

__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ dy, const float* __restrict__ scale,
                                                     int64_t rps, float* __restrict__ out, int64_t M, int N,
                                                     int rows_per_block) {
  const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t m1 = (m0 + rows_per_block < M) ? m0 + rows_per_block : M;
  for (int c = threadIdx.x; c < N; c += blockDim.x) {
    float s = 0.f;
    for (int64_t m = m0; m < m1; ++m) s += dy[m * N + c] * (scale ? scale[m / rps] : 1.f);
    atomicAdd(out + c, s);
  }
}

int colsum_atomic(const float* dy, const float* scale, int64_t rps, float* out, int64_t M, int N, hipStream_t s) {
  if (M <= 0 || N <= 0) return MICF_OK;
  int rpb = (int)((M + 1023) / 1024);
  if (rpb < 32) rpb = 32;
  const int blocks = (int)((M + rpb - 1) / rpb);
  const int threads = N >= 256 ? 256 : ((N + 63) / 64) * 64;
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(threads), 0, s, dy, scale, rps, out, M, N, rpb);
  MICF_RETURN_LAUNCH();
}

}  // namespace micf

using namespace micf;

extern "C" int micf_linear_fwd(const float* a1, const float* a2, int k1, const float* w, const float* bias,
                               const float* resid, const float* dp_scale, int64_t rows_per_sample, float* y,
                               float* pre_act, int64_t M, int N, int K, int act, micf_stream_t stream) {
  if (!a1 || !w || !y || M < 0 || N <= 0 || K <= 0 || k1 <= 0 || k1 > K || (k1 < K && !a2)) return MICF_EINVAL;
  if (M >= (1LL << 31)) return MICF_EUNSUPPORTED;
  if (rows_per_sample <= 0) rows_per_sample = M > 0 ? M : 1;
  const int k2 = K - k1;
  const int avec = (k1 % 4 == 0) && (k2 % 4 == 0) && aligned16(a1) && (!a2 || aligned16(a2));
  RowsT pa{a1, a2 ? a2 : a1, k1, k1, k2 > 0 ? k2 : 1, (int)M, nullptr, 1, 0, avec};
  RowsT qa{w, w, K, K, 1, N, nullptr, 1, 0, (K % 4 == 0) && aligned16(w)};
  LinFwdEpi epi{bias, resid, dp_scale, rows_per_sample, y, pre_act, N, act};
  return launch_gemm(pa, qa, epi, M, N, K, 1, (hipStream_t)stream) == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

extern "C" int micf_linear_bwd_data(const float* dy, const float* dp_scale, int64_t rows_per_sample, const float* w,
                                    const float* pre_act, float* da1, float* da2, int k1, int accumulate, int64_t M,
                                    int N, int K, micf_stream_t stream) {
  if (!dy || !w || !da1 || M < 0 || N <= 0 || K <= 0 || k1 <= 0 || k1 > K || (k1 < K && !da2)) return MICF_EINVAL;
  if (M >= (1LL << 31)) return MICF_EUNSUPPORTED;
  if (rows_per_sample <= 0) rows_per_sample = M > 0 ? M : 1;
  // dA[m, k] = sum_n (s*dy)[m, n] * W[n, k]:  P = dy rows (r = n contiguous), Q(x = k, r = n) = W[r*K + x] (x contiguous)
  RowsT pa{dy, dy, N, N, 1, (int)M, dp_scale, rows_per_sample, 0, (N % 4 == 0) && aligned16(dy)};
  RowsD qa{w, w, K, K, 1, K, nullptr, 1, 0, (K % 4 == 0) && aligned16(w)};
  LinBwdDataEpi epi{pre_act, da1, da2 ? da2 : da1, k1, K, accumulate};
  return launch_gemm(pa, qa, epi, M, K, N, 1, (hipStream_t)stream) == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

extern "C" int micf_linear_bwd_weight(const float* dy, const float* dp_scale, int64_t rows_per_sample, const float* a1,
                                      const float* a2, int k1, int a_gelu, float* dw, float* dbias, int64_t M, int N,
                                      int K, micf_stream_t stream) {
  if (!dy || !a1 || !dw || M < 0 || N <= 0 || K <= 0 || k1 <= 0 || k1 > K || (k1 < K && !a2)) return MICF_EINVAL;
  if (M >= (1LL << 31)) return MICF_EUNSUPPORTED;
  if (M == 0) return MICF_OK;
  if (rows_per_sample <= 0) rows_per_sample = M;
  const int k2 = K - k1;
  // dW[n, k] = sum_m (s*dy)[m, n] * A[m, k]:  P(x = n, r = m) = dy[r*N + x],  Q(x = k, r = m) = A[r, x]
  RowsD pa{dy, dy, N, N, 1, N, dp_scale, rows_per_sample, 0, (N % 4 == 0) && aligned16(dy)};
  const int avec = (k1 % 4 == 0) && (k2 % 4 == 0) && aligned16(a1) && (!a2 || aligned16(a2));
  RowsD qa{a1, a2 ? a2 : a1, k1, k1, k2 > 0 ? k2 : 1, K, nullptr, 1, a_gelu, avec};
  AtomicEpi epi{dw, K};
  // dbias = column sums of (s*dy): taken from the dy slab already staged in LDS by the same launch
  if (launch_gemm(pa, qa, epi, N, K, (int)M, pick_splits(N, K, M), (hipStream_t)stream, dbias) != hipSuccess) return MICF_ELAUNCH;
  return MICF_OK;
}
