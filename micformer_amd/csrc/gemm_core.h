// gemm_core.h -- fp32 MFMA tiled GEMM skeleton for gfx950 (CDNA4), shared by every GEMM-shaped kernel.
//
//   C[i, j] = sum_r P[i, r] * Q[r, j]        i < I, j < J, r in [r_begin, r_end)
//
// ORIENTATION RULE: I is the dimension that is CONTIGUOUS in the output tensor (the feature / channel axis of a
// channels-last activation), J is the token axis.  v_mfma_f32_16x16x4_f32 leaves 4 consecutive i of one j in a lane's 4
// accumulator registers, so every result leaves the kernel as one 16-byte store (and bias / residual / saved
// pre-activation arrive as 16-byte loads) -- the epilogue functor is called with (i, j, float4).
//
// One 256-thread workgroup (4 waves of 64) computes a BI x BJ tile (exact fp32 MFMA: bitwise a k-ordered fmaf chain,
// 64 FLOP/clk/SIMD).  32-deep reduction slabs of P and Q are staged through LDS in r-major order (Ps[r][i], Qs[r][j]);
// the next slab is fetched from HBM into registers BEFORE the MFMAs of the current slab and committed to the other
// LDS buffer after them (one barrier per slab), so HBM/L2 latency hides behind the matrix pipe even when only one
// workgroup fits the problem (the small-token stages of MicFormer).
//
// LDS layout: row stride == 16 (mod 32) banks and the column rotated by 8*(r>>2), col' = (x + 8*(r>>2)) mod BX:
//   * MFMA fragment reads (lane l -> x = l & 15, r = l >> 4) are conflict-free ds_read_b32;
//   * the transposing stores of r-contiguous sources (thread = 4 consecutive r of one x) are <= 2-way (free);
//   * float4 stores of x-contiguous sources stay 16-byte aligned.
//
// Operands are described by ACCESSOR functors (how element (x, r) is found in HBM: plain rows, two-source
// concatenation, 3x3x3 halo gather, stride==kernel patch gather, ...), so windowing, im2col and cat are index
// arithmetic and are never materialised.  Integer divisions by run-time extents go through FastDiv (mul-hi + shift).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace micf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;

// LDS row strides (in floats), chosen per staging kind so that both the MFMA fragment reads and the commits are
// (nearly) conflict-free WITHOUT any address swizzle -- fragment reads are then `base + compile-time offset`:
//   x-contiguous sources (float4 commits along x): stride == 16 (mod 32): reads of rows r, r+1 hit disjoint bank halves;
//   r-contiguous sources (a thread commits 4 consecutive r of one x as scalars): stride == 18 (mod 32): the 8 r-groups of
//   a wave land 8 banks apart (2-way = free on ds_write_b32) and rows r, r+1 of a read overlap in only 2 banks.
constexpr int lds_stride_d(int b) { return b + ((16 - b % 32) + 32) % 32; }
constexpr int lds_stride_t(int b) { return b + ((18 - b % 32) + 32) % 32; }

// n / d for 0 <= n < 2^31, d >= 1, without the ~40-instruction hardware-less integer division
struct FastDiv {
  uint32_t d, m, s;
  FastDiv() : d(1), m(1), s(0) {}
  explicit FastDiv(uint32_t d_) : d(d_ ? d_ : 1) {
    s = 0;
    while ((1ull << s) < d) ++s;
    m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (__umulhi(n, m) + n) >> s; }
  __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const { q = div(n); r = n - q * d; }
};

template <int BI_, int BJ_, int WI_, int WJ_, int BR_ = 32>
struct Tile {
  static constexpr int BI = BI_, BJ = BJ_, WI = WI_, WJ = WJ_;
  static constexpr int BR = BR_;   // reduction depth staged per LDS slab (128 for the latency-bound small-token stages)
  static constexpr int TI = BI / WI / 16, TJ = BJ / WJ / 16;
  static_assert(WI * WJ == 4, "4 waves per workgroup");
  static_assert(TI * 16 * WI == BI && TJ * 16 * WJ == BJ, "tile must split into 16x16 MFMA tiles");
};

// ------------------------------------------------------------------ staging: fetch (HBM -> registers), commit (-> LDS)
// "T" mapping: the source is contiguous along r.  A thread handles 4 consecutive r of one x (float4 from HBM).
template <int BX, int kBR>
struct StageT {
  static constexpr int NIT = (BX * (kBR / 4) + kThreads - 1) / kThreads;
  static constexpr int STRIDE = lds_stride_t(BX);
  float v[NIT][4];
  template <class F4>   // f4(x, r, out[4]) zero-fills anything out of range
  __device__ __forceinline__ void fetch(int x0, int r0, int tid, F4 f4) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < BX * (kBR / 4)) f4(x0 + idx / (kBR / 4), r0 + (idx % (kBR / 4)) * 4, v[it]);
    }
  }
  __device__ __forceinline__ void commit(float* S, int tid) const {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < BX * (kBR / 4)) {
        const int x = idx / (kBR / 4), r4 = (idx % (kBR / 4)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) S[(r4 + e) * STRIDE + x] = v[it][e];
      }
    }
  }
};
// "D" mapping: the source is contiguous along x.  A thread handles 4 consecutive x of one r (float4 both sides).
template <int BX, int kBR>
struct StageD {
  static constexpr int XV = BX / 4;
  static constexpr int NIT = (kBR * XV + kThreads - 1) / kThreads;
  static constexpr int STRIDE = lds_stride_d(BX);
  float v[NIT][4];
  template <class F4>
  __device__ __forceinline__ void fetch(int x0, int r0, int tid, F4 f4) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < kBR * XV) f4(x0 + (idx % XV) * 4, r0 + idx / XV, v[it]);
    }
  }
  __device__ __forceinline__ void commit(float* S, int tid) const {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < kBR * XV) {
        const int r = idx / XV, x4 = (idx % XV) * 4;
        *reinterpret_cast<float4*>(&S[r * STRIDE + x4]) = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
      }
    }
  }
};
// "E" mapping: per-element functor, threads walk x fastest (scalar both sides).
template <int BX, int kBR>
struct StageE {
  static constexpr int NIT = (BX * kBR + kThreads - 1) / kThreads;
  static constexpr int STRIDE = lds_stride_d(BX);
  float v[NIT];
  template <class F1>   // f1(x, r) -> value (zero when out of range)
  __device__ __forceinline__ void fetch(int x0, int r0, int tid, F1 f1) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < BX * kBR) v[it] = f1(x0 + idx % BX, r0 + idx / BX);
    }
  }
  __device__ __forceinline__ void commit(float* S, int tid) const {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < BX * kBR) { const int r = idx / BX, x = idx % BX; S[r * STRIDE + x] = v[it]; }
    }
  }
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// bf16 mode's GELU: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7) on the hardware exp / rcp -- ~14 instructions instead of
// erff's ~40, and the derivative reuses the same exponential (exp(-u^2) with u = x / sqrt 2 IS the Gaussian of the pdf term).
// The 32^3 / 16^3 fused block kernels are VALU-bound on exactly this (65-75 VALU instructions per MFMA).  fp32 (parity) mode
// keeps erff / expf.
__device__ __forceinline__ void gelu_fast_parts(float x, float& cdf, float& gauss) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * u);
  gauss = __expf(-u * u);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float half_erfc = 0.5f * poly * gauss;                    // 0.5 * erfc(u)
  cdf = x < 0.f ? half_erfc : 1.0f - half_erfc;                   // 0.5 * (1 + erf(x / sqrt 2))
}
template <bool FAST> __device__ __forceinline__ float gelu_t(float x) {
  if constexpr (FAST) { float cdf, g; gelu_fast_parts(x, cdf, g); return x * cdf; }
  else return gelu_f(x);
}
template <bool FAST> __device__ __forceinline__ float gelu_grad_t(float x) {
  if constexpr (FAST) { float cdf, g; gelu_fast_parts(x, cdf, g); return cdf + x * 0.39894228040143267794f * g; }
  else return gelu_grad_f(x);
}

// ------------------------------------------------------------------ plain row-major accessors
// XF selects, at COMPILE time, the transforming variant (per-sample DropPath scale and/or GELU on load); the plain
// variant carries none of that code.
//
// RowsT: element (x, r) = p1[x*ld1 + r] for r < k1, else p2[x*ld2 + (r - k1)]   (r contiguous; concat along r);
// scale indexed by x / rps.
template <bool XF>
struct RowsT {
  const float* p1;
  const float* p2;
  int k1;
  int64_t ld1, ld2;
  int X;
  int vec;              // 1: ld1, ld2, k1 multiples of 4 and bases 16-B aligned -> float4 loads
  const float* scale;   // XF only
  FastDiv rps;
  int gelu;
  template <int BX, int BR> using Stage = StageT<BX, BR>;
  template <int BX, int BR>
  __device__ __forceinline__ void fetch(Stage<BX, BR>& st, int x0, int r0, int r_end, int tid) const {
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
      if constexpr (XF) {
        if (gelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
        }
        if (scale) {
          const float s = scale[rps.div(x)];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= s;
        }
      }
    });
  }
};

// RowsD: element (x, r) = p1[r*ld1 + x] for x < k1, else p2[r*ld2 + (x - k1)]   (x contiguous; concat along x);
// scale indexed by r / rps.  Used when the REDUCTION runs over rows (dW = dY^T A) or for W seen transposed.
template <bool XF>
struct RowsD {
  const float* p1;
  const float* p2;
  int k1;
  int64_t ld1, ld2;
  int X;
  int vec;
  const float* scale;
  FastDiv rps;
  int gelu;
  template <int BX, int BR> using Stage = StageD<BX, BR>;
  template <int BX, int BR>
  __device__ __forceinline__ void fetch(Stage<BX, BR>& st, int x0, int r0, int r_end, int tid) const {
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
      if constexpr (XF) {
        if (gelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
        }
        if (scale) {
          const float s = scale[rps.div(r)];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= s;
        }
      }
    });
  }
};

inline RowsT<false> rows_t(const float* p, int64_t ld, int X, int R) {
  return RowsT<false>{p, p, R, ld, 1, X, (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0), nullptr, FastDiv(1), 0};
}
inline RowsD<false> rows_d(const float* p, int64_t ld, int X) {
  return RowsD<false>{p, p, X, ld, 1, X, (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0), nullptr, FastDiv(1), 0};
}

// Generic per-element accessor: f(x, r) -> float, called only for in-range (x < X, r < r_end).
// MAP_T: threads walk r fastest (4 per thread) -- use when the source is contiguous along r;
// otherwise threads walk x fastest.
template <class F, bool MAP_T>
struct Elem {
  F f;
  int X;
  template <int BX, int BR> using Stage = typename std::conditional<MAP_T, StageT<BX, BR>, StageE<BX, BR>>::type;
  template <int BX, int BR>
  __device__ __forceinline__ void fetch(Stage<BX, BR>& st, int x0, int r0, int r_end, int tid) const {
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
// Epilogue contract:  epi(i, j, f32x4 v, int n)  receives C[i .. i+n-1, j] (n = min(4, I - i), i % 4 == 0).
// colsum (optional): side 1: out[i] += sum_r P[i, r] (workgroups of the first j-tile only); side 2: out[j] += sum_r Q[r, j]
// (workgroups of the first i-tile only) -- the bias gradient of a weight-gradient GEMM comes for free from the dY slab
// already sitting in LDS.
template <class T, class PAcc, class QAcc, class Epi>
__global__ void __launch_bounds__(kThreads) gemm_kernel(PAcc pa, QAcc qa, Epi epi, int I, int J, int R, int r_chunk,
                                                        int tiles_i, int tiles_j, int jpb, float* colsum,
                                                        int colsum_side) {
  constexpr int kBR = T::BR;
  using SPt = typename PAcc::template Stage<T::BI, T::BR>;
  using SQt = typename QAcc::template Stage<T::BJ, T::BR>;
  constexpr int SP = SPt::STRIDE, SQ = SQt::STRIDE;
  __shared__ __attribute__((aligned(16))) float Ps[2][kBR * SP];
  __shared__ __attribute__((aligned(16))) float Qs[2][kBR * SQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / T::WJ, wj = wave % T::WJ;
  const int bi = blockIdx.x % tiles_i, bjg = blockIdx.x / tiles_i;
  const int i0 = bi * T::BI;
  // this workgroup walks jpb consecutive token tiles; the (tile, slab) pairs form ONE software pipeline, so the HBM
  // latency of tile t+1 hides behind the MFMAs + epilogue of tile t even when the reduction is only 1-2 slabs deep
  const int jt_begin = bjg * jpb;
  const int jt_end = (jt_begin + jpb < tiles_j) ? jt_begin + jpb : tiles_j;
  const int r_begin = blockIdx.y * r_chunk;
  const int r_end = (r_begin + r_chunk < R) ? r_begin + r_chunk : R;
  const int nslab = (r_end - r_begin + kBR - 1) / kBR;
  const int n_it = (jt_end - jt_begin) * nslab;

  f32x4 acc[T::TI][T::TJ];
#pragma unroll
  for (int a = 0; a < T::TI; ++a)
#pragma unroll
    for (int b = 0; b < T::TJ; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  SPt sp;
  SQt sq;
  const bool cs_p = (colsum_side == 1) && (bjg == 0) && (tid < T::BI);
  const bool cs_q = (colsum_side == 2) && (bi == 0) && (tid < T::BJ);
  float csum = 0.f;

  const int li = lane & 15, lr = lane >> 4;
  if (n_it <= 0) return;
  pa.template fetch<T::BI, T::BR>(sp, i0, r_begin, r_end, tid);
  qa.template fetch<T::BJ, T::BR>(sq, jt_begin * T::BJ, r_begin, r_end, tid);
  sp.commit(Ps[0], tid);
  sq.commit(Qs[0], tid);
  __syncthreads();
  int cur = 0, jt = jt_begin, slab = 0;
  for (int it = 0; it < n_it; ++it) {
    const bool more = it + 1 < n_it;
    int njt = jt, nslb = slab + 1;
    if (nslb == nslab) { nslb = 0; ++njt; }
    if (more) {                                   // prefetch the next slab (possibly of the next token tile) into registers
      pa.template fetch<T::BI, T::BR>(sp, i0, r_begin + nslb * kBR, r_end, tid);
      qa.template fetch<T::BJ, T::BR>(sq, njt * T::BJ, r_begin + nslb * kBR, r_end, tid);
    }
    // per-lane fragment bases: every read below is base + compile-time offset
    const float* P = Ps[cur] + lr * SP + wi * (T::BI / T::WI) + li;
    const float* Q = Qs[cur] + lr * SQ + wj * (T::BJ / T::WJ) + li;
    const int kvalid = r_end - (r_begin + slab * kBR);      // >= 1; slabs shorter than kBR skip the zero-padded k-steps
    auto kstep = [&](int rr) {
      float a[T::TI], b[T::TJ];
#pragma unroll
      for (int t = 0; t < T::TI; ++t) a[t] = P[rr * SP + t * 16];
#pragma unroll
      for (int t = 0; t < T::TJ; ++t) b[t] = Q[rr * SQ + t * 16];
#pragma unroll
      for (int ta = 0; ta < T::TI; ++ta)
#pragma unroll
        for (int tb = 0; tb < T::TJ; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    };
    if (kvalid >= kBR) {
#pragma unroll
      for (int rr = 0; rr < kBR; rr += 4) kstep(rr);
    } else {
#pragma unroll 2
      for (int rr = 0; rr < kvalid; rr += 4) kstep(rr);
    }
    if (cs_p && jt == 0) {
#pragma unroll 8
      for (int r = 0; r < kBR; ++r) csum += Ps[cur][r * SP + tid];
    }
    if (cs_q) {
#pragma unroll 8
      for (int r = 0; r < kBR; ++r) csum += Qs[cur][r * SQ + tid];
    }
    if (slab == nslab - 1) {
      // C/D fragment: lane l, reg v -> i = (l >> 4) * 4 + v, j = l & 15
      const int j0 = jt * T::BJ;
#pragma unroll
      for (int ta = 0; ta < T::TI; ++ta)
#pragma unroll
        for (int tb = 0; tb < T::TJ; ++tb) {
          const int i = i0 + wi * (T::BI / T::WI) + ta * 16 + lr * 4;
          const int j = j0 + wj * (T::BJ / T::WJ) + tb * 16 + li;
          if (i < I && j < J) epi(i, j, acc[ta][tb], (I - i < 4) ? I - i : 4);
          acc[ta][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      if (cs_q && j0 + tid < J) { atomicAdd(colsum + j0 + tid, csum); csum = 0.f; }
    }
    if (more) {
      sp.commit(Ps[cur ^ 1], tid);
      sq.commit(Qs[cur ^ 1], tid);
    }
    __syncthreads();
    cur ^= 1; jt = njt; slab = nslb;
  }
  if (cs_p && i0 + tid < I) atomicAdd(colsum + i0 + tid, csum);
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

template <class T, class PAcc, class QAcc, class Epi>
inline void launch_tile(PAcc pa, QAcc qa, Epi epi, int I, int64_t J, int R, int r_chunk, int splits, float* colsum,
                        int colsum_side, hipStream_t stream) {
  const int tiles_i = ceil_div(I, T::BI);
  const int tiles_j = ceil_div(J, T::BJ);
  // token tiles per workgroup: keep >= ~1024 workgroups (4 per CU) in the grid, at most 8 tiles deep
  int jpb = 1;
  while (jpb < 8 && (int64_t)tiles_i * (tiles_j / (jpb * 2)) * splits >= 1024) jpb *= 2;
  const int64_t blocks = (int64_t)tiles_i * ceil_div(tiles_j, jpb);
  dim3 g((unsigned)blocks, splits, 1);
  hipLaunchKernelGGL((gemm_kernel<T, PAcc, QAcc, Epi>), g, dim3(kThreads), 0, stream, pa, qa, epi, I, (int)J, R, r_chunk,
                     tiles_i, tiles_j, jpb, colsum, colsum_side);
}

// Launch with a tile picked from the problem shape: I = contiguous output axis (features), J = tokens.
// splits > 1 only with an accumulating (atomic) epilogue.  Small problems (the 8^3 / 4^3 token stages) take 32x32
// tiles so that more of the 256 CUs get a workgroup.
template <class PAcc, class QAcc, class Epi>
inline hipError_t launch_gemm(PAcc pa, QAcc qa, Epi epi, int I, int64_t J, int R, int splits, hipStream_t stream,
                              float* colsum = nullptr, int colsum_side = 0) {
  if (I <= 0 || J <= 0 || R <= 0) return hipSuccess;
  if (splits < 1) splits = 1;
  int r_chunk = ceil_div(ceil_div(R, splits), 32) * 32;
  splits = ceil_div(R, r_chunk);
#define MICF_LT(...) launch_tile<Tile<__VA_ARGS__>>(pa, qa, epi, I, J, R, r_chunk, splits, colsum, colsum_side, stream)
  const int64_t big = (int64_t)ceil_div(I, 64) * ceil_div(J, 64) * splits;
  if (I <= 16) {
    MICF_LT(16, 128, 1, 4);
  } else if (J <= 16) {
    MICF_LT(64, 16, 4, 1);
  } else if (I <= 32 || J <= 32 || big < 256) {
    if (I <= 32 && (int64_t)ceil_div(J, 64) * splits >= 256) MICF_LT(32, 64, 2, 2);
    else if (J <= 32 && (int64_t)ceil_div(I, 64) * splits >= 256) MICF_LT(64, 32, 2, 2);
    else MICF_LT(32, 32, 2, 2);
  } else if (I % 64 != 0 && I % 48 == 0) {
    MICF_LT(48, 64, 1, 4);
  } else {
    MICF_LT(64, 64, 2, 2);
  }
#undef MICF_LT
  return hipGetLastError();
}

// How many reduction splits to use when R is long and the output tile grid is small (weight gradients, small-grid
// convolutions): aim for >= 512 workgroups while keeping >= 64 reduction rows per split.
inline int pick_splits(int64_t I, int64_t J, int64_t R) {
  int64_t tiles = (int64_t)ceil_div(I, 64) * ceil_div(J, 64);
  if (tiles < 256) tiles = (int64_t)ceil_div(I, 32) * ceil_div(J, 32);
  int64_t want = (512 + tiles - 1) / tiles;
  int64_t maxs = (R + 63) / 64;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  return (int)want;
}

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// scalar tail helper for epilogues: apply f(e) for e < n
#define MICF_FOR_N(n, e) _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < (n))

}  // namespace micf
