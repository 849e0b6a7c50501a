// linear.hip -- nn.Linear forward / data-gradient / weight-gradient on the fp32 MFMA GEMM core.
// Reference ops replaced: F.linear in (Cross)WindowAttention3D q/kv/proj (MS.py:188-201, 246-259),
// Mlp fc1/GELU/fc2 (MS.py:28-34), concat_back_dim on torch.cat (MS.py:1027-1030), and the
// residual + DropPath adds of the blocks (MS.py:419,424,517,522) fused into the epilogue.
// Orientation (gemm_core.h): the feature axis of the OUTPUT is the tile's I side, tokens are J, so every output row
// segment leaves as 16-byte stores and bias / residual / saved pre-activation arrive as 16-byte loads.
#include "common.h"
#include "gemm_dma.h"

namespace micf {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }

// y[j, i] for token j, feature i.   ACT: 0 none, 1 GELU.  RESID: 0 none, 1 resid + s * v
template <int ACT, int RESID>
struct LinFwdEpi {
  const float* bias; const float* resid; const float* scale; FastDiv rps;
  float* y; float* pre; int N; int vec;
  __device__ __forceinline__ float block_scale() const { return 1.f; }
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
      if (vec == 2) {          // output larger than the 256 MB MALL (the composed head's patch matrix): stream it past the caches
        typedef float f32x4_nt __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f32x4_nt{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4_nt*>(y + o));
      } else {
        st4(y + o, v[0], v[1], v[2], v[3]);
      }
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
  const float* row_scale; FastDiv rps;        // DMA path: DropPath scale applied per output token (s * (dy W) == (s dy) W)
  __device__ __forceinline__ float block_scale() const { return 1.f; }
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    if (row_scale) { const float s = row_scale[rps.div(j)]; v[0] *= s; v[1] *= s; v[2] *= s; v[3] *= s; }
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
  // DMA path: the reduction range of a workgroup lies inside ONE sample, so the DropPath scale is a per-workgroup constant
  const float* blk_scale; int r_chunk; FastDiv rps;
  __device__ __forceinline__ float block_scale() const { return blk_scale ? blk_scale[rps.div(blockIdx.y * r_chunk)] : 1.f; }
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    float* p = out + (int64_t)j * ld + i;
    const float s = block_scale();
    MICF_FOR_N(n, e) atomicAdd(p + e, v[e] * s);
  }
};

// partial dW of one token split, stored (not added) at ws[split][j][i]
struct LinWgtWsEpi {
  float* ws; int64_t ld; int64_t split_stride;
  const float* blk_scale; int r_chunk; FastDiv rps;
  __device__ __forceinline__ float block_scale() const { return blk_scale ? blk_scale[rps.div(blockIdx.y * r_chunk)] : 1.f; }
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    float* p = ws + (int64_t)blockIdx.y * split_stride + (int64_t)j * ld + i;
    const float s = block_scale();
    if (n == 4) st4(p, v[0] * s, v[1] * s, v[2] * s, v[3] * s);
    else { MICF_FOR_N(n, e) p[e] = v[e] * s; }
  }
};

// dw[e] += sum_split ws[split][e]   (dw is only touched by this stream: plain read-modify-write, 16 bytes per lane)
__global__ void __launch_bounds__(256) reduce_splits_kernel(const float* __restrict__ ws, float* __restrict__ dw, int64_t n4,
                                                            int splits, int64_t split_stride) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
    float4 acc = ld4(dw + 4 * e);
    int s = 0;
    for (; s + 8 <= splits; s += 8) {          // 8 independent 16-byte loads in flight per lane
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4(ws + (s + u) * split_stride + 4 * e);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; s < splits; ++s) {
      const float4 v = ld4(ws + s * split_stride + 4 * e);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    st4(dw + 4 * e, acc.x, acc.y, acc.z, acc.w);
  }
}

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

// narrow matrices (N % 4 == 0, N <= 256, no scale): thread (column group g, row lane r) streams float4 rows, LDS-reduces over r
__global__ void __launch_bounds__(256) colsum4_kernel(const float* __restrict__ dy, float* __restrict__ out, int64_t M, int N,
                                                      int64_t rows_per_block) {
  __shared__ float4 part[256];
  const int ng = N / 4, rl = 256 / ng;                     // row lanes per column group
  const int g = threadIdx.x % ng, r = threadIdx.x / ng;
  const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t m1 = (m0 + rows_per_block < M) ? m0 + rows_per_block : M;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < rl) {
    for (int64_t m = m0 + r; m < m1; m += rl) {
      const float4 v = *reinterpret_cast<const float4*>(dy + m * N + 4 * g);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < ng) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < rl; ++k) { const float4 v = part[k * ng + threadIdx.x]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    atomicAdd(out + 4 * threadIdx.x + 0, s.x); atomicAdd(out + 4 * threadIdx.x + 1, s.y);
    atomicAdd(out + 4 * threadIdx.x + 2, s.z); atomicAdd(out + 4 * threadIdx.x + 3, s.w);
  }
}

int colsum_atomic(const float* dy, const float* scale, int64_t rps, float* out, int64_t M, int N, hipStream_t s) {
  if (M <= 0 || N <= 0) return MICF_OK;
  if (!scale && N % 4 == 0 && N <= 256 && aligned16(dy)) {
    int64_t rpb = (M + 1023) / 1024;
    if (rpb < 256) rpb = 256;
    const int blocks = (int)((M + rpb - 1) / rpb);
    hipLaunchKernelGGL(colsum4_kernel, dim3(blocks), dim3(256), 0, s, dy, out, M, N, rpb);
    MICF_RETURN_LAUNCH();
  }
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
                               float* pre_act, int64_t M, int N, int K, int act, int dtype, micf_stream_t stream) {
  if (!a1 || !w || !y || M < 0 || N <= 0 || K <= 0 || k1 <= 0 || k1 > K || (k1 < K && !a2) || (dtype != 0 && dtype != 1)) return MICF_EINVAL;
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
  {   // LDS-DMA core: plain operands, K a multiple of 16
    const DmaOperand P{w, K, N}, Q{a1, K, (int)M};
    if (!a2 && N >= 48 && M >= 64 && dma_ok(P, false, Q, false, K, K)) {
      if (act) return RC((launch_gemm_dma<false, false>(P, Q, LinFwdEpi<1, 0>{bias, nullptr, nullptr, rps, y, pre_act, N, evec}, N, M, K, 1, s, nullptr, dtype)));
      if (resid) return RC((launch_gemm_dma<false, false>(P, Q, LinFwdEpi<0, 1>{bias, resid, dp_scale, rps, y, nullptr, N, evec}, N, M, K, 1, s, nullptr, dtype)));
      const int big = (evec && (int64_t)M * N * (int64_t)sizeof(float) > ((int64_t)256 << 20)) ? 2 : evec;
      return RC((launch_gemm_dma<false, false>(P, Q, LinFwdEpi<0, 0>{bias, nullptr, nullptr, rps, y, nullptr, N, big}, N, M, K, 1, s, nullptr, dtype)));
    }
  }
  if (act) return RC(launch_gemm(pa, qa, LinFwdEpi<1, 0>{bias, nullptr, nullptr, rps, y, pre_act, N, evec}, N, M, K, 1, s));
  if (resid) return RC(launch_gemm(pa, qa, LinFwdEpi<0, 1>{bias, resid, dp_scale, rps, y, nullptr, N, evec}, N, M, K, 1, s));
  return RC(launch_gemm(pa, qa, LinFwdEpi<0, 0>{bias, nullptr, nullptr, rps, y, nullptr, N, evec}, N, M, K, 1, s));
}

extern "C" int micf_linear_bwd_data(const float* dy, const float* dp_scale, int64_t rows_per_sample, const float* w,
                                    const float* pre_act, float* da1, float* da2, int k1, int accumulate, int64_t M,
                                    int N, int K, int dtype, micf_stream_t stream) {
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
  {   // LDS-DMA core: P = W as (x = k, r = n) [kind X], Q = dy rows [kind R]; DropPath scale moves to the epilogue
    const DmaOperand P{w, K, K}, Q{dy, N, (int)M};
    if (K >= 48 && M >= 64 && dma_ok(P, true, Q, false, N, N)) {
      const FastDiv rpsd((uint32_t)rows_per_sample);
      if (pre_act) return RC((launch_gemm_dma<true, false>(P, Q, LinBwdDataEpi<1>{pre_act, da1, d2, k1, K, accumulate, evec, dp_scale, rpsd}, K, M, N, 1, s, nullptr, dtype)));
      return RC((launch_gemm_dma<true, false>(P, Q, LinBwdDataEpi<0>{nullptr, da1, d2, k1, K, accumulate, evec, dp_scale, rpsd}, K, M, N, 1, s, nullptr, dtype)));
    }
  }
  if (dp_scale) {
    RowsT<true> qa{dy, dy, N, N, 1, (int)M, dvec, dp_scale, FastDiv((uint32_t)rows_per_sample), 0};
    if (pre_act) return RC(launch_gemm(pa, qa, LinBwdDataEpi<1>{pre_act, da1, d2, k1, K, accumulate, evec, nullptr, FastDiv(1)}, K, M, N, 1, s));
    return RC(launch_gemm(pa, qa, LinBwdDataEpi<0>{nullptr, da1, d2, k1, K, accumulate, evec, nullptr, FastDiv(1)}, K, M, N, 1, s));
  }
  RowsT<false> qa = rows_t(dy, N, (int)M, N);
  if (pre_act) return RC(launch_gemm(pa, qa, LinBwdDataEpi<1>{pre_act, da1, d2, k1, K, accumulate, evec, nullptr, FastDiv(1)}, K, M, N, 1, s));
  return RC(launch_gemm(pa, qa, LinBwdDataEpi<0>{nullptr, da1, d2, k1, K, accumulate, evec, nullptr, FastDiv(1)}, K, M, N, 1, s));
}

static int64_t wgt_chunk(int64_t M, int N, int K) {   // tokens per split of the DMA weight-gradient path
  const int64_t tiles = (int64_t)ceil_div(K, 64) * ceil_div(N, 64);
  int64_t want = (512 + tiles - 1) / tiles;            // ~2 workgroups per CU
  int64_t chunk = (M + want - 1) / want;
  chunk = (chunk + kDmaBR - 1) / kDmaBR * kDmaBR;
  if (chunk < 128) chunk = 128;
  if (chunk > M) chunk = (M + kDmaBR - 1) / kDmaBR * kDmaBR;
  return chunk;
}

extern "C" int64_t micf_linear_bwd_weight_workspace(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int64_t splits = (M + 127) / 128;              // upper bound on the number of token splits
  const int64_t cap = splits < 512 ? splits : 512;
  return cap * (int64_t)N * K;
}

extern "C" int micf_linear_bwd_weight(const float* dy, const float* dp_scale, int64_t rows_per_sample, const float* a1,
                                      const float* a2, int k1, int a_gelu, float* dw, float* dbias, int64_t M, int N,
                                      int K, float* workspace, int64_t workspace_floats, int dtype, micf_stream_t stream) {
  if (!dy || !a1 || !dw || M < 0 || N <= 0 || K <= 0 || k1 <= 0 || k1 > K || (k1 < K && !a2)) return MICF_EINVAL;
  if (M >= (1LL << 31)) return MICF_EUNSUPPORTED;
  if (M == 0) return MICF_OK;
  if (rows_per_sample <= 0) rows_per_sample = M;
  const int k2 = K - k1;
  // dW[n, k] = sum_m (s*dy)[m, n] * A[m, k]:  C[i = k, j = n];  P(x = k, r = m) = A[r, x],  Q(x = n, r = m) = dy[r*N + x].
  // dbias = column sums of the (scaled) dy slab staged in LDS (colsum side 2), from the workgroups of the first i-tile.
  const int avec = (k1 % 4 == 0) && (k2 % 4 == 0) && aligned16(a1) && (!a2 || aligned16(a2));
  const int dvec = (N % 4 == 0) && aligned16(dy);
  hipStream_t s = (hipStream_t)stream;
  const FastDiv rps((uint32_t)rows_per_sample);
  {   // LDS-DMA core: P = A as (x = k, r = m), Q = dy as (x = n, r = m), both kind X; tokens split over workgroups.
      // With DropPath the split size divides rows_per_sample so every workgroup sees one sample (scale in the epilogue).
    const DmaOperand P{a1, K, K}, Q{dy, N, N};
    if (!a2 && !a_gelu && K >= 48 && N >= 48 && M >= 64) {
      // Token splits: with a workspace the partials are STORED (16-byte stores) and summed by a tiny second launch, so many
      // splits are cheap; without one every split costs N*K device-scope atomics (one fabric transaction each), so few.
      const bool vec_ok = (K % 4 == 0) && aligned16(dw) && workspace && aligned16(workspace);
      int64_t chunk = wgt_chunk(M, N, K);
      if (!vec_ok) {
        const int64_t tiles = (int64_t)ceil_div(K, 64) * ceil_div(N, 64);
        int64_t want = (256 + tiles - 1) / tiles;
        if (want > 48) want = 48;
        chunk = (M + want - 1) / want;
        chunk = (chunk + kDmaBR - 1) / kDmaBR * kDmaBR;
        if (chunk < 256) chunk = 256;
        if (chunk > M) chunk = (M + kDmaBR - 1) / kDmaBR * kDmaBR;
      }
      if (dp_scale) {          // largest divisor of rows_per_sample (halving) that is <= chunk: one sample per workgroup
        int64_t c = rows_per_sample;
        while (c > chunk && c % 2 == 0) c /= 2;
        chunk = c;
      }
      const int dsplits = ceil_div(M, chunk);
      if (chunk % kDmaBR == 0 && (!dp_scale || rows_per_sample % chunk == 0) && dma_ok(P, true, Q, true, M, (int)chunk)) {
        const int64_t NK = (int64_t)N * K;
        if (vec_ok && dsplits > 1 && (int64_t)dsplits * NK <= workspace_floats) {
          LinWgtWsEpi wepi{workspace, K, NK, dp_scale, (int)chunk, rps};
          if (launch_gemm_dma<true, true>(P, Q, wepi, K, N, (int)M, dsplits, s, dbias, dtype) != hipSuccess) return MICF_ELAUNCH;
          int blocks = (int)((NK / 4 + 255) / 256);
          if (blocks > 2048) blocks = 2048;
          hipLaunchKernelGGL(reduce_splits_kernel, dim3(blocks), dim3(256), 0, s, workspace, dw, NK / 4, dsplits, NK);
          MICF_RETURN_LAUNCH();
        }
        LinWgtEpi depi{dw, K, dp_scale, (int)chunk, rps};
        return RC((launch_gemm_dma<true, true>(P, Q, depi, K, N, (int)M, dsplits, s, dbias, dtype)));
      }
    }
  }
  const int splits = pick_splits(K, N, M);
  LinWgtEpi epi{dw, K, nullptr, 0, FastDiv(1)};
  if (a_gelu || dp_scale) {
    RowsD<true> pa{a1, a2 ? a2 : a1, k1, k1, k2 > 0 ? k2 : 1, K, avec, nullptr, FastDiv(1), a_gelu};
    RowsD<true> qa{dy, dy, N, N, 1, N, dvec, dp_scale, rps, 0};
    return RC(launch_gemm(pa, qa, epi, K, N, (int)M, splits, s, dbias, dbias ? 2 : 0));
  }
  RowsD<false> pa{a1, a2 ? a2 : a1, k1, k1, k2 > 0 ? k2 : 1, K, avec, nullptr, FastDiv(1), 0};
  RowsD<false> qa = rows_d(dy, N, N);
  return RC(launch_gemm(pa, qa, epi, K, N, (int)M, splits, s, dbias, dbias ? 2 : 0));
}
