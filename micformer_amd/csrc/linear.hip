// linear.hip -- nn.Linear forward / data-gradient / weight-gradient on the fp32 MFMA GEMM core.
// Reference ops replaced: F.linear in (Cross)WindowAttention3D q/kv/proj (MS.py:188-201, 246-259),
// Mlp fc1/GELU/fc2 (MS.py:28-34), concat_back_dim on torch.cat (MS.py:1027-1030), and the
// residual + DropPath adds of the blocks (MS.py:419,424,517,522) fused into the epilogue.
// Orientation (gemm_core.h): the feature axis of the OUTPUT is the tile's I side, tokens are J, so every output row
// segment leaves as 16-byte stores and bias / residual / saved pre-activation arrive as 16-byte loads.
#include "common.h"

namespace micf {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }

// y[j, i] for token j, feature i.   ACT: 0 none, 1 GELU.  RESID: 0 none, 1 resid + s * v
template <int ACT, int RESID>
struct LinFwdEpi {
  const float* bias; const float* resid; const float* scale; FastDiv rps;
  float* y; float* pre; int N; int vec;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    const int64_t o = (int64_t)j * N + i;
    float s = 1.f;
    if constexpr (RESID) { if (scale) s = scale[rps.div(j)]; }
    if (vec && n == 4) {
      if (bias) { const float4 b = ld4(bias + i); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
      if constexpr (ACT) {
        if (pre) st4(pre + o, v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
      }
      if constexpr (RESID) {
        const float4 r = ld4(resid + o);
        v[0] = r.x + s * v[0]; v[1] = r.y + s * v[1]; v[2] = r.z + s * v[2]; v[3] = r.w + s * v[3];
      }
      st4(y + o, v[0], v[1], v[2], v[3]);
    } else {
      MICF_FOR_N(n, e) {
        float t = v[e] + (bias ? bias[i + e] : 0.f);
        if constexpr (ACT) { if (pre) pre[o + e] = t; t = gelu_f(t); }
        if constexpr (RESID) t = resid[o + e] + s * t;
        y[o + e] = t;
      }
    }
  }
};

// dA[j, i] (token j, input feature i) (=|+=) v (* GELU'(pre[j, i])), split over two destinations at k1
template <int GELU_GRAD>
struct LinBwdDataEpi {
  const float* pre; float* d1; float* d2; int k1, K, acc, vec;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    if constexpr (GELU_GRAD) {
      const float* pp = pre + (int64_t)j * K + i;
      if (vec && n == 4) { const float4 h = ld4(pp); v[0] *= gelu_grad_f(h.x); v[1] *= gelu_grad_f(h.y); v[2] *= gelu_grad_f(h.z); v[3] *= gelu_grad_f(h.w); }
      else { MICF_FOR_N(n, e) v[e] *= gelu_grad_f(pp[e]); }
    }
    if (vec && n == 4) {            // k1 % 4 == 0: the 4 outputs never straddle the two destinations
      float* dst = (i < k1) ? d1 + (int64_t)j * k1 + i : d2 + (int64_t)j * (K - k1) + (i - k1);
      if (acc) { const float4 o = ld4(dst); v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
      st4(dst, v[0], v[1], v[2], v[3]);
    } else {
      MICF_FOR_N(n, e) {
        const int ii = i + e;
        float* dst = (ii < k1) ? d1 + (int64_t)j * k1 + ii : d2 + (int64_t)j * (K - k1) + (ii - k1);
        *dst = acc ? *dst + v[e] : v[e];
      }
    }
  }
};

// dW[j, i] += v  (weight row j = output feature n, column i = input feature k)
struct LinWgtEpi {
  float* out; int64_t ld;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    float* p = out + (int64_t)j * ld + i;
    MICF_FOR_N(n, e) atomicAdd(p + e, v[e]);
  }
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
#define RC(e) ((e) == hipSuccess ? MICF_OK : MICF_ELAUNCH)

extern "C" int micf_linear_fwd(const float* a1, const float* a2, int k1, const float* w, const float* bias,
                               const float* resid, const float* dp_scale, int64_t rows_per_sample, float* y,
                               float* pre_act, int64_t M, int N, int K, int act, micf_stream_t stream) {
  if (!a1 || !w || !y || M < 0 || N <= 0 || K <= 0 || k1 <= 0 || k1 > K || (k1 < K && !a2)) return MICF_EINVAL;
  if (M >= (1LL << 31)) return MICF_EUNSUPPORTED;
  if (rows_per_sample <= 0) rows_per_sample = M > 0 ? M : 1;
  const int k2 = K - k1;
  const int avec = (k1 % 4 == 0) && (k2 % 4 == 0) && aligned16(a1) && (!a2 || aligned16(a2));
  // C[i = n, j = m] = sum_k W[n, k] * A[m, k]
  RowsT<false> pa = rows_t(w, K, N, K);
  RowsT<false> qa{a1, a2 ? a2 : a1, k1, k1, k2 > 0 ? k2 : 1, (int)M, avec, nullptr, FastDiv(1), 0};
  const int evec = (N % 4 == 0) && aligned16(y) && (!bias || aligned16(bias)) && (!resid || aligned16(resid)) &&
                   (!pre_act || aligned16(pre_act));
  const FastDiv rps((uint32_t)rows_per_sample);
  hipStream_t s = (hipStream_t)stream;
  if (act && resid) return MICF_EUNSUPPORTED;
  if (act) return RC(launch_gemm(pa, qa, LinFwdEpi<1, 0>{bias, nullptr, nullptr, rps, y, pre_act, N, evec}, N, M, K, 1, s));
  if (resid) return RC(launch_gemm(pa, qa, LinFwdEpi<0, 1>{bias, resid, dp_scale, rps, y, nullptr, N, evec}, N, M, K, 1, s));
  return RC(launch_gemm(pa, qa, LinFwdEpi<0, 0>{bias, nullptr, nullptr, rps, y, nullptr, N, evec}, N, M, K, 1, s));
}

extern "C" int micf_linear_bwd_data(const float* dy, const float* dp_scale, int64_t rows_per_sample, const float* w,
                                    const float* pre_act, float* da1, float* da2, int k1, int accumulate, int64_t M,
                                    int N, int K, micf_stream_t stream) {
  if (!dy || !w || !da1 || M < 0 || N <= 0 || K <= 0 || k1 <= 0 || k1 > K || (k1 < K && !da2)) return MICF_EINVAL;
  if (M >= (1LL << 31)) return MICF_EUNSUPPORTED;
  if (rows_per_sample <= 0) rows_per_sample = M > 0 ? M : 1;
  // dA[m, k] = sum_n (s*dy)[m, n] * W[n, k]:  C[i = k, j = m];  P(x = k, r = n) = W[r*K + x],  Q(x = m, r = n) = dy[x*N + r]
  RowsD<false> pa = rows_d(w, K, K);
  const int dvec = (N % 4 == 0) && aligned16(dy);
  const int evec = (k1 % 4 == 0) && ((K - k1) % 4 == 0) && aligned16(da1) && (!da2 || aligned16(da2)) &&
                   (!pre_act || aligned16(pre_act));
  hipStream_t s = (hipStream_t)stream;
  float* d2 = da2 ? da2 : da1;
  if (dp_scale) {
    RowsT<true> qa{dy, dy, N, N, 1, (int)M, dvec, dp_scale, FastDiv((uint32_t)rows_per_sample), 0};
    if (pre_act) return RC(launch_gemm(pa, qa, LinBwdDataEpi<1>{pre_act, da1, d2, k1, K, accumulate, evec}, K, M, N, 1, s));
    return RC(launch_gemm(pa, qa, LinBwdDataEpi<0>{nullptr, da1, d2, k1, K, accumulate, evec}, K, M, N, 1, s));
  }
  RowsT<false> qa = rows_t(dy, N, (int)M, N);
  if (pre_act) return RC(launch_gemm(pa, qa, LinBwdDataEpi<1>{pre_act, da1, d2, k1, K, accumulate, evec}, K, M, N, 1, s));
  return RC(launch_gemm(pa, qa, LinBwdDataEpi<0>{nullptr, da1, d2, k1, K, accumulate, evec}, K, M, N, 1, s));
}

extern "C" int micf_linear_bwd_weight(const float* dy, const float* dp_scale, int64_t rows_per_sample, const float* a1,
                                      const float* a2, int k1, int a_gelu, float* dw, float* dbias, int64_t M, int N,
                                      int K, micf_stream_t stream) {
  if (!dy || !a1 || !dw || M < 0 || N <= 0 || K <= 0 || k1 <= 0 || k1 > K || (k1 < K && !a2)) return MICF_EINVAL;
  if (M >= (1LL << 31)) return MICF_EUNSUPPORTED;
  if (M == 0) return MICF_OK;
  if (rows_per_sample <= 0) rows_per_sample = M;
  const int k2 = K - k1;
  // dW[n, k] = sum_m (s*dy)[m, n] * A[m, k]:  C[i = k, j = n];  P(x = k, r = m) = A[r, x],  Q(x = n, r = m) = dy[r*N + x].
  // dbias = column sums of the (scaled) dy slab staged in LDS (colsum side 2), from the workgroups of the first i-tile.
  const int avec = (k1 % 4 == 0) && (k2 % 4 == 0) && aligned16(a1) && (!a2 || aligned16(a2));
  const int dvec = (N % 4 == 0) && aligned16(dy);
  const int splits = pick_splits(K, N, M);
  hipStream_t s = (hipStream_t)stream;
  LinWgtEpi epi{dw, K};
  const FastDiv rps((uint32_t)rows_per_sample);
  if (a_gelu || dp_scale) {
    RowsD<true> pa{a1, a2 ? a2 : a1, k1, k1, k2 > 0 ? k2 : 1, K, avec, nullptr, FastDiv(1), a_gelu};
    RowsD<true> qa{dy, dy, N, N, 1, N, dvec, dp_scale, rps, 0};
    return RC(launch_gemm(pa, qa, epi, K, N, (int)M, splits, s, dbias, dbias ? 2 : 0));
  }
  RowsD<false> pa{a1, a2 ? a2 : a1, k1, k1, k2 > 0 ? k2 : 1, K, avec, nullptr, FastDiv(1), 0};
  RowsD<false> qa = rows_d(dy, N, N);
  return RC(launch_gemm(pa, qa, epi, K, N, (int)M, splits, s, dbias, dbias ? 2 : 0));
}
