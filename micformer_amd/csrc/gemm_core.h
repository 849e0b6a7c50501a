// gemm_core.h -- fp32 MFMA tiled GEMM skeleton for gfx950 (CDNA4), shared by every GEMM-shaped kernel.
//
//   C[i, j] = sum_r P[i, r] * Q[r, j]        i < I, j < J, r in [r_begin, r_end)
//
// One 256-thread workgroup (4 waves of 64) computes a BI x BJ tile with v_mfma_f32_16x16x4_f32 (exact fp32: bitwise
// a k-ordered fmaf chain, 64 FLOP/clk/SIMD).  32-deep reduction slabs of P and Q are staged through LDS in r-major
// order (Ps[r][i], Qs[r][j]); the next slab is fetched from HBM into registers BEFORE the MFMAs of the current slab
// and committed to the other LDS buffer after them (one barrier per slab), so HBM/L2 latency hides behind the
// matrix pipe even when only one workgroup fits the problem (the small-token stages of MicFormer).
//
// LDS layout: row stride == 16 (mod 32) banks and the column rotated by 8*(r>>2), col' = (x + 8*(r>>2)) mod BX:
//   * MFMA fragment reads (lane l -> x = l & 15, r = l >> 4) are conflict-free ds_read_b32;
//   * the transposing stores of r-contiguous sources (thread = 4 consecutive r of one x) are <= 2-way (free);
//   * float4 stores of x-contiguous sources stay 16-byte aligned.
//
// Operands are described by ACCESSOR functors (how element (x, r) is found in HBM: plain rows, two-source
// concatenation, 3x3x3 halo gather, stride==kernel patch gather, ...) and results leave through an EPILOGUE functor
// (bias / GELU / residual + DropPath scale / scatter / atomicAdd), so windowing, im2col and cat are index arithmetic
// and are never materialised.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace micf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kBR = 32;  // reduction depth staged per LDS slab

constexpr int lds_stride(int b) { return (b % 32 == 16) ? b : b + 16; }

template <int BI_, int BJ_, int WI_, int WJ_>
struct Tile {
  static constexpr int BI = BI_, BJ = BJ_, WI = WI_, WJ = WJ_;
  static constexpr int TI = BI / WI / 16, TJ = BJ / WJ / 16;
  static constexpr int SP = lds_stride(BI), SQ = lds_stride(BJ);
  static_assert(WI * WJ == 4, "4 waves per workgroup");
  static_assert(TI * 16 * WI == BI && TJ * 16 * WJ == BJ, "tile must split into 16x16 MFMA tiles");
};

// rotated column of element (r, x) inside a slab of width BX
template <int BX>
__device__ __forceinline__ int swz(int r, int x) {
  int c = x + 8 * (r >> 2);
  if constexpr ((BX & (BX - 1)) == 0) return c & (BX - 1);
  else return c % BX;
}

// ------------------------------------------------------------------ staging: fetch (HBM -> registers), commit (-> LDS)
// "T" mapping: the source is contiguous along r.  A thread handles 4 consecutive r of one x (float4 from HBM).
template <int BX>
struct StageT {
  static constexpr int NIT = (BX * (kBR / 4) + kThreads - 1) / kThreads;
  float v[NIT][4];
  template <class F4>   // f4(x, r, out[4]) zero-fills anything out of range
  __device__ __forceinline__ void fetch(int x0, int r0, int tid, F4 f4) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < BX * (kBR / 4)) f4(x0 + idx / (kBR / 4), r0 + (idx % (kBR / 4)) * 4, v[it]);
    }
  }
  template <int SX>
  __device__ __forceinline__ void commit(float* S, int tid) const {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < BX * (kBR / 4)) {
        const int x = idx / (kBR / 4), r4 = (idx % (kBR / 4)) * 4;
        const int c = swz<BX>(r4, x);                 // r4..r4+3 share r >> 2
#pragma unroll
        for (int e = 0; e < 4; ++e) S[(r4 + e) * SX + c] = v[it][e];
      }
    }
  }
};
// "D" mapping: the source is contiguous along x.  A thread handles 4 consecutive x of one r (float4 both sides).
template <int BX>
struct StageD {
  static constexpr int XV = BX / 4;
  static constexpr int NIT = (kBR * XV + kThreads - 1) / kThreads;
  float v[NIT][4];
  template <class F4>
  __device__ __forceinline__ void fetch(int x0, int r0, int tid, F4 f4) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < kBR * XV) f4(x0 + (idx % XV) * 4, r0 + idx / XV, v[it]);
    }
  }
  template <int SX>
  __device__ __forceinline__ void commit(float* S, int tid) const {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < kBR * XV) {
        const int r = idx / XV, x4 = (idx % XV) * 4;
        *reinterpret_cast<float4*>(&S[r * SX + swz<BX>(r, x4)]) = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
      }
    }
  }
};
// "E" mapping: per-element functor, threads walk x fastest (scalar both sides).
template <int BX>
struct StageE {
  static constexpr int NIT = (BX * kBR + kThreads - 1) / kThreads;
  float v[NIT];
  template <class F1>   // f1(x, r) -> value (zero when out of range)
  __device__ __forceinline__ void fetch(int x0, int r0, int tid, F1 f1) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < BX * kBR) v[it] = f1(x0 + idx % BX, r0 + idx / BX);
    }
  }
  template <int SX>
  __device__ __forceinline__ void commit(float* S, int tid) const {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < BX * kBR) { const int r = idx / BX, x = idx % BX; S[r * SX + swz<BX>(r, x)] = v[it]; }
    }
  }
};

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
  template <int BX> using Stage = StageT<BX>;
  template <int BX>
  __device__ __forceinline__ void fetch(Stage<BX>& st, int x0, int r0, int r_end, int tid) const {
    st.fetch(x0, r0, tid, [&](int x, int r, float* v) {
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
  template <int BX> using Stage = StageD<BX>;
  template <int BX>
  __device__ __forceinline__ void fetch(Stage<BX>& st, int x0, int r0, int r_end, int tid) const {
    st.fetch(x0, r0, tid, [&](int x, int r, float* v) {
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
  template <int BX> using Stage = typename std::conditional<MAP_T, StageT<BX>, StageE<BX>>::type;
  template <int BX>
  __device__ __forceinline__ void fetch(Stage<BX>& st, int x0, int r0, int r_end, int tid) const {
    if constexpr (MAP_T) {
      st.fetch(x0, r0, tid, [&](int x, int r, float* v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (x < X && r + e < r_end) ? f(x, r + e) : 0.f;
      });
    } else {
      st.fetch(x0, r0, tid, [&](int x, int r) { return (x < X && r < r_end) ? f(x, r) : 0.f; });
    }
  }
};
template <bool MAP_T, class F>
__host__ __device__ inline Elem<F, MAP_T> make_elem(F f, int X) { return Elem<F, MAP_T>{f, X}; }

// ------------------------------------------------------------------ the kernel
// colsum (optional): out[i] += sum_r P[i, r] over this block's reduction range (blocks with blockIdx.y == 0 only) --
// the bias gradient of a weight-gradient GEMM comes for free from the P slab already sitting in LDS.
template <class T, class PAcc, class QAcc, class Epi>
__global__ void __launch_bounds__(kThreads) gemm_kernel(PAcc pa, QAcc qa, Epi epi, int I, int J, int R, int r_chunk,
                                                        float* colsum) {
  __shared__ __attribute__((aligned(16))) float Ps[2][kBR * T::SP];
  __shared__ __attribute__((aligned(16))) float Qs[2][kBR * T::SQ];
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

  typename PAcc::template Stage<T::BI> sp;
  typename QAcc::template Stage<T::BJ> sq;
  const bool do_colsum = (colsum != nullptr) && (blockIdx.y == 0) && (tid < T::BI);
  float csum = 0.f;

  const int li = lane & 15, lr = lane >> 4;
  pa.template fetch<T::BI>(sp, i0, r_begin, r_end, tid);
  qa.template fetch<T::BJ>(sq, j0, r_begin, r_end, tid);
  sp.template commit<T::SP>(Ps[0], tid);
  sq.template commit<T::SQ>(Qs[0], tid);
  __syncthreads();
  int cur = 0;
  for (int r0 = r_begin; r0 < r_end; r0 += kBR) {
    const bool more = r0 + kBR < r_end;
    if (more) {                                   // prefetch the next slab into registers
      pa.template fetch<T::BI>(sp, i0, r0 + kBR, r_end, tid);
      qa.template fetch<T::BJ>(sq, j0, r0 + kBR, r_end, tid);
    }
    const float* P = Ps[cur];
    const float* Q = Qs[cur];
#pragma unroll
    for (int rr = 0; rr < kBR; rr += 4) {
      float a[T::TI], b[T::TJ];
      const int r = rr + lr;
#pragma unroll
      for (int t = 0; t < T::TI; ++t) a[t] = P[r * T::SP + swz<T::BI>(r, wi * (T::BI / T::WI) + t * 16 + li)];
#pragma unroll
      for (int t = 0; t < T::TJ; ++t) b[t] = Q[r * T::SQ + swz<T::BJ>(r, wj * (T::BJ / T::WJ) + t * 16 + li)];
#pragma unroll
      for (int ta = 0; ta < T::TI; ++ta)
#pragma unroll
        for (int tb = 0; tb < T::TJ; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
    if (do_colsum) {
#pragma unroll 8
      for (int r = 0; r < kBR; ++r) csum += P[r * T::SP + swz<T::BI>(r, tid)];
    }
    if (more) {
      sp.template commit<T::SP>(Ps[cur ^ 1], tid);
      sq.template commit<T::SQ>(Qs[cur ^ 1], tid);
    }
    __syncthreads();
    cur ^= 1;
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
  if (do_colsum && i0 + tid < I) atomicAdd(colsum + i0 + tid, csum);
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

template <class T, class PAcc, class QAcc, class Epi>
inline void launch_tile(PAcc pa, QAcc qa, Epi epi, int64_t I, int J, int R, int r_chunk, int splits, float* colsum,
                        hipStream_t stream) {
  dim3 g(ceil_div(I, T::BI), ceil_div(J, T::BJ), splits);
  hipLaunchKernelGGL((gemm_kernel<T, PAcc, QAcc, Epi>), g, dim3(kThreads), 0, stream, pa, qa, epi, (int)I, J, R, r_chunk,
                     colsum);
}

// Launch with a tile picked from the problem shape.  splits > 1 only with an accumulating (atomic) epilogue.
// Small problems (the 8^3 / 4^3 token stages) take 32x32 tiles so that more of the 256 CUs get a workgroup.
template <class PAcc, class QAcc, class Epi>
inline hipError_t launch_gemm(PAcc pa, QAcc qa, Epi epi, int64_t I, int J, int R, int splits, hipStream_t stream,
                              float* colsum = nullptr) {
  if (I <= 0 || J <= 0 || R <= 0) return hipSuccess;
  if (splits < 1) splits = 1;
  int r_chunk = ceil_div(ceil_div(R, splits), kBR) * kBR;
  splits = ceil_div(R, r_chunk);
  if (J <= 16) {
    launch_tile<Tile<64, 16, 4, 1>>(pa, qa, epi, I, J, R, r_chunk, splits, colsum, stream);
  } else if (J <= 32 || (int64_t)ceil_div(I, 64) * ceil_div(J, 64) * splits < 256) {
    if (J <= 32 && (int64_t)ceil_div(I, 64) * splits >= 256)
      launch_tile<Tile<64, 32, 2, 2>>(pa, qa, epi, I, J, R, r_chunk, splits, colsum, stream);
    else
      launch_tile<Tile<32, 32, 2, 2>>(pa, qa, epi, I, J, R, r_chunk, splits, colsum, stream);
  } else if (J % 64 != 0 && J % 48 == 0) {
    launch_tile<Tile<64, 48, 4, 1>>(pa, qa, epi, I, J, R, r_chunk, splits, colsum, stream);
  } else {
    launch_tile<Tile<64, 64, 2, 2>>(pa, qa, epi, I, J, R, r_chunk, splits, colsum, stream);
  }
  return hipGetLastError();
}

// How many reduction splits to use when R is long and the output tile grid is small (weight gradients, small-grid
// convolutions): aim for >= 512 workgroups while keeping >= 64 reduction rows per split.
inline int pick_splits(int64_t I, int J, int64_t R) {
  int64_t tiles = (int64_t)ceil_div(I, 64) * ceil_div(J, 64);
  if (tiles < 256) tiles = (int64_t)ceil_div(I, 32) * ceil_div(J, 32);
  int64_t want = (512 + tiles - 1) / tiles;
  int64_t maxs = (R + 63) / 64;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  return (int)want;
}

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace micf
