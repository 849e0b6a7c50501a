// gemm_core.h -- fp32 MFMA tiled GEMM skeleton for gfx950 (CDNA4), shared by every GEMM-shaped kernel.
//
//   C[i, j] = sum_r P[i, r] * Q[r, j]        i < I, j < J, r in [r_begin, r_end)
//
// One 256-thread workgroup (4 waves of 64) computes a BI x BJ tile with v_mfma_f32_16x16x4_f32
// (exact fp32: bitwise a k-ordered fmaf chain, 64 FLOP/clk/SIMD), staging 16-deep reduction slabs of
// P and Q through LDS in r-major order (Ps[r][i], Qs[r][j]) so that every MFMA fragment read
// (lane l -> i = l & 15, r = l >> 4) is a conflict-free ds_read_b32: the row stride is == 16 (mod 32) banks.
// Operands are described by ACCESSOR functors (how element (x, r) is found in HBM: plain rows, two-source
// concatenation, 3x3x3 halo gather, stride==kernel patch gather, ...) and results leave through an EPILOGUE
// functor (bias / GELU / residual + DropPath scale / scatter / atomicAdd), so windowing, im2col and cat are
// index arithmetic and are never materialised.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace micf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kBR = 16;  // reduction depth staged per LDS slab

constexpr int lds_stride(int b) { return (b % 32 == 16) ? b : b + 16; }

template <int BI_, int BJ_, int WI_, int WJ_>
struct Tile {
  static constexpr int BI = BI_, BJ = BJ_, WI = WI_, WJ = WJ_;
  static constexpr int TI = BI / WI / 16, TJ = BJ / WJ / 16;
  static constexpr int SP = lds_stride(BI), SQ = lds_stride(BJ);
  static_assert(WI * WJ == 4, "4 waves per workgroup");
  static_assert(TI * 16 * WI == BI && TJ * 16 * WJ == BJ, "tile must split into 16x16 MFMA tiles");
};

// ------------------------------------------------------------------ LDS fill helpers
// "T" mapping: the source is contiguous along r.  Thread handles 4 consecutive r of one x.
//   f4(x, r, out[4]) must zero-fill anything out of range.
template <int BX, int SX, class F4>
__device__ __forceinline__ void fill_T(float* S, int x0, int r0, int tid, F4 f4) {
#pragma unroll
  for (int idx = tid; idx < BX * 4; idx += kThreads) {
    const int x = idx >> 2, r4 = (idx & 3) * 4;
    float v[4];
    f4(x0 + x, r0 + r4, v);
    S[(r4 + 0) * SX + x] = v[0];
    S[(r4 + 1) * SX + x] = v[1];
    S[(r4 + 2) * SX + x] = v[2];
    S[(r4 + 3) * SX + x] = v[3];
  }
}
// "D" mapping: the source is contiguous along x.  Thread handles 4 consecutive x of one r.
template <int BX, int SX, class F4>
__device__ __forceinline__ void fill_D(float* S, int x0, int r0, int tid, F4 f4) {
  constexpr int XV = BX / 4;
#pragma unroll
  for (int idx = tid; idx < kBR * XV; idx += kThreads) {
    const int r = idx / XV, x4 = (idx % XV) * 4;
    float v[4];
    f4(x0 + x4, r0 + r, v);
    *reinterpret_cast<float4*>(&S[r * SX + x4]) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// ------------------------------------------------------------------ plain row-major accessors
// Element (x, r) = p1[x*ld1 + r] for r < k1, else p2[x*ld2 + (r - k1)]   (r contiguous; concat along r).
// Optional per-x scale s[x / rps] (DropPath) and optional GELU on load.
struct RowsT {
  const float* p1;
  const float* p2;
  int k1;
  int64_t ld1, ld2;
  int X;
  const float* scale;   // per-sample scale indexed by x / rps, or nullptr
  int64_t rps;
  int gelu;
  int vec;              // 1: ld1, ld2, k1 multiples of 4 and bases 16-B aligned -> float4 loads
  template <int BX, int SX>
  __device__ __forceinline__ void load(float* S, int x0, int r0, int r_end, int tid) const {
    fill_T<BX, SX>(S, x0, r0, tid, [&](int x, int r, float* v) {
      v[0] = v[1] = v[2] = v[3] = 0.f;
      if (x >= X || r >= r_end) return;
      if (vec && r + 3 < r_end) {
        const float4 q = (r < k1) ? *reinterpret_cast<const float4*>(p1 + (int64_t)x * ld1 + r)
                                  : *reinterpret_cast<const float4*>(p2 + (int64_t)x * ld2 + (r - k1));
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rr = r + e;
          if (rr < r_end) v[e] = (rr < k1) ? p1[(int64_t)x * ld1 + rr] : p2[(int64_t)x * ld2 + (rr - k1)];
        }
      }
      if (gelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
      }
      if (scale) {
        const float s = scale[x / rps];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= s;
      }
    });
  }
};

// Element (x, r) = p1[r*ld1 + x] for x < k1, else p2[r*ld2 + (x - k1)]   (x contiguous; concat along x).
// Optional per-r scale s[r / rps] and optional GELU on load.  Used when the REDUCTION runs over rows (dW = dY^T A).
struct RowsD {
  const float* p1;
  const float* p2;
  int k1;
  int64_t ld1, ld2;
  int X;
  const float* scale;
  int64_t rps;
  int gelu;
  int vec;
  template <int BX, int SX>
  __device__ __forceinline__ void load(float* S, int x0, int r0, int r_end, int tid) const {
    fill_D<BX, SX>(S, x0, r0, tid, [&](int x, int r, float* v) {
      v[0] = v[1] = v[2] = v[3] = 0.f;
      if (r >= r_end || x >= X) return;
      if (vec && x + 3 < X) {
        const float4 q = (x < k1) ? *reinterpret_cast<const float4*>(p1 + (int64_t)r * ld1 + x)
                                  : *reinterpret_cast<const float4*>(p2 + (int64_t)r * ld2 + (x - k1));
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int xx = x + e;
          if (xx < X) v[e] = (xx < k1) ? p1[(int64_t)r * ld1 + xx] : p2[(int64_t)r * ld2 + (xx - k1)];
        }
      }
      if (gelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
      }
      if (scale) {
        const float s = scale[r / rps];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= s;
      }
    });
  }
};

// Generic per-element accessor: f(x, r) -> float, called only for in-range (x < X, r < r_end).
// MAP_T: threads walk r fastest (4 per thread) -- use when the source is contiguous along r;
// otherwise threads walk x fastest.
template <class F, bool MAP_T>
struct Elem {
  F f;
  int X;
  template <int BX, int SX>
  __device__ __forceinline__ void load(float* S, int x0, int r0, int r_end, int tid) const {
    if constexpr (MAP_T) {
      fill_T<BX, SX>(S, x0, r0, tid, [&](int x, int r, float* v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (x < X && r + e < r_end) ? f(x, r + e) : 0.f;
      });
    } else {
#pragma unroll
      for (int idx = tid; idx < BX * kBR; idx += kThreads) {
        const int r = idx / BX, x = idx % BX;
        S[r * SX + x] = (x0 + x < X && r0 + r < r_end) ? f(x0 + x, r0 + r) : 0.f;
      }
    }
  }
};
template <bool MAP_T, class F>
__host__ __device__ inline Elem<F, MAP_T> make_elem(F f, int X) { return Elem<F, MAP_T>{f, X}; }

// ------------------------------------------------------------------ the kernel
template <class T, class PAcc, class QAcc, class Epi>
__global__ void __launch_bounds__(kThreads) gemm_kernel(PAcc pa, QAcc qa, Epi epi, int I, int J, int R, int r_chunk) {
  __shared__ __attribute__((aligned(16))) float Ps[kBR * T::SP];
  __shared__ __attribute__((aligned(16))) float Qs[kBR * T::SQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / T::WJ, wj = wave % T::WJ;
  const int i0 = blockIdx.x * T::BI, j0 = blockIdx.y * T::BJ;
  const int r_begin = blockIdx.z * r_chunk;
  const int r_end = (r_begin + r_chunk < R) ? r_begin + r_chunk : R;

  f32x4 acc[T::TI][T::TJ];
#pragma unroll
  for (int a = 0; a < T::TI; ++a)
#pragma unroll
    for (int b = 0; b < T::TJ; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, lr = lane >> 4;
  for (int r0 = r_begin; r0 < r_end; r0 += kBR) {
    pa.template load<T::BI, T::SP>(Ps, i0, r0, r_end, tid);
    qa.template load<T::BJ, T::SQ>(Qs, j0, r0, r_end, tid);
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < kBR; rr += 4) {
      float a[T::TI], b[T::TJ];
#pragma unroll
      for (int t = 0; t < T::TI; ++t) a[t] = Ps[(rr + lr) * T::SP + wi * (T::BI / T::WI) + t * 16 + li];
#pragma unroll
      for (int t = 0; t < T::TJ; ++t) b[t] = Qs[(rr + lr) * T::SQ + wj * (T::BJ / T::WJ) + t * 16 + li];
#pragma unroll
      for (int ta = 0; ta < T::TI; ++ta)
#pragma unroll
        for (int tb = 0; tb < T::TJ; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D fragment: lane l, reg v -> row (l >> 4) * 4 + v, col l & 15
#pragma unroll
  for (int ta = 0; ta < T::TI; ++ta)
#pragma unroll
    for (int tb = 0; tb < T::TJ; ++tb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = i0 + wi * (T::BI / T::WI) + ta * 16 + lr * 4 + v;
        const int j = j0 + wj * (T::BJ / T::WJ) + tb * 16 + li;
        if (i < I && j < J) epi(i, j, acc[ta][tb][v]);
      }
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Launch with a tile picked from J (the narrow dimension on this path); splits > 1 only with an atomic epilogue.
template <class PAcc, class QAcc, class Epi>
inline hipError_t launch_gemm(PAcc pa, QAcc qa, Epi epi, int64_t I, int J, int R, int splits, hipStream_t stream) {
  if (I <= 0 || J <= 0 || R <= 0) return hipSuccess;
  if (splits < 1) splits = 1;
  int r_chunk = ceil_div(ceil_div(R, splits), kBR) * kBR;
  splits = ceil_div(R, r_chunk);
  if (J <= 16) {
    using T = Tile<64, 16, 4, 1>;
    dim3 g(ceil_div(I, T::BI), ceil_div(J, T::BJ), splits);
    hipLaunchKernelGGL((gemm_kernel<T, PAcc, QAcc, Epi>), g, dim3(kThreads), 0, stream, pa, qa, epi, (int)I, J, R, r_chunk);
  } else if (J <= 32) {
    using T = Tile<64, 32, 2, 2>;
    dim3 g(ceil_div(I, T::BI), ceil_div(J, T::BJ), splits);
    hipLaunchKernelGGL((gemm_kernel<T, PAcc, QAcc, Epi>), g, dim3(kThreads), 0, stream, pa, qa, epi, (int)I, J, R, r_chunk);
  } else if (J % 64 != 0 && J % 48 == 0) {
    using T = Tile<64, 48, 4, 1>;
    dim3 g(ceil_div(I, T::BI), ceil_div(J, T::BJ), splits);
    hipLaunchKernelGGL((gemm_kernel<T, PAcc, QAcc, Epi>), g, dim3(kThreads), 0, stream, pa, qa, epi, (int)I, J, R, r_chunk);
  } else {
    using T = Tile<64, 64, 2, 2>;
    dim3 g(ceil_div(I, T::BI), ceil_div(J, T::BJ), splits);
    hipLaunchKernelGGL((gemm_kernel<T, PAcc, QAcc, Epi>), g, dim3(kThreads), 0, stream, pa, qa, epi, (int)I, J, R, r_chunk);
  }
  return hipGetLastError();
}

// How many reduction splits to use when R (rows) is huge and the output tile grid is tiny (weight gradients).
inline int pick_splits(int64_t I, int J, int64_t R) {
  const int64_t tiles = (int64_t)ceil_div(I, 64) * ceil_div(J, 64);
  int64_t want = (1024 + tiles - 1) / tiles;          // aim for ~1024 workgroups (4 per CU)
  int64_t maxs = (R + 255) / 256;                      // at least 256 reduction rows per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  return (int)want;
}

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace micf
