// gemm_dma.h -- fp32 MFMA GEMM for plain row-major operands, fed by the gfx950 LDS-DMA engine.
//
//   C[i, j] = sum_r P[i, r] * Q[r, j]            (same orientation rule as gemm_core.h: I = contiguous output axis)
//
// Why a second core: the register-staged skeleton (gemm_core.h) pays one full L2/HBM round trip per 32-deep slab
// (load -> ds_write -> barrier -> ds_read -> MFMA is a dependent chain, ~0.85 us per slab measured), which dominates the
// 8^3 / 4^3 token stages and caps the big stages at ~2 TB/s.  Here every wave issues `global_load_lds_dwordx4`
// (16 B per lane, no VGPR round trip, no ds_write) for slab it+3 while the matrix pipe works on slab it: a 4-deep LDS ring,
// one counted `s_waitcnt vmcnt(N)` + one raw `s_barrier` per slab.
//
// LDS images are the operands' NATURAL layouts (the DMA writes lane-linear, so no transposition is possible):
//   kind R (r contiguous in HBM, e.g. A[m][k], W[n][k]):   [64 x][16 r]  -- 64-byte rows
//   kind X (x contiguous in HBM, e.g. W[n][k] seen as (x = k, r = n), dY[m][n] as (x = n, r = m)):   [16 r][64 x]
// and the MFMA fragments are read with ONE ds_read_b128 per 4 k-steps thanks to a k-permutation: in k-step s (0..3) the
// lane group lr supplies reduction index r = 4*lr + s -- both operands use the same map, so the sum is unchanged:
//   kind R: lane (li, lr) reads [x = 16*t + li][4*lr .. 4*lr+3]            -> element s feeds step s of tile t
//   kind X: lane (li, lr) reads [r = 4*lr + s][4*li .. 4*li+3] per step s  -> element t feeds tile t (tile t holds x = 4*li + t)
// Output mapping (acc[t][v], D row = 4*lr + v, col = li):
//   P kind R: i = i0 + 16*t + 4*lr + v (float4 over v);   P kind X: i = i0 + 16*lr + 4*v + t (float4 over t)
//   Q kind R: j = j0 + 16*wave + li;                      Q kind X: j = j0 + 4*li + wave
// Workgroup = 4 waves, tile 64 x 64, each wave 64 (i) x 16 (j): 4 accumulator tiles, 16 MFMA per slab.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_core.h"

namespace micf {

constexpr int kDmaBR = 16;      // slab depth
constexpr int kDmaNS = 4;       // ring depth

struct DmaOperand {             // plain row-major operand
  const float* p;
  int64_t ld;                   // leading dimension (floats)
  int X;                        // extent along x (rows for kind R, columns for kind X)
};

// one 16-byte-per-lane async copy HBM -> LDS (lane-linear destination at lds_byte + 16*lane)
__device__ __forceinline__ void dma16(const float* gsrc, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_byte)
               : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// PX / QX: operand is kind X (x contiguous) instead of kind R.  R must be a multiple of 16; rows/cols beyond the extents are
// clamped on load (their products are discarded by the epilogue bounds).
template <bool PX, bool QX, class Epi>
__global__ void __launch_bounds__(256) gemm_dma_kernel(DmaOperand P, DmaOperand Q, Epi epi, int I, int J, int R, int r_chunk,
                                                       int tiles_i, float* colsum) {
  __shared__ __attribute__((aligned(1024))) float Ps[kDmaNS][64 * kDmaBR];
  __shared__ __attribute__((aligned(1024))) float Qs[kDmaNS][64 * kDmaBR];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lr = lane >> 4;
  const int bi = blockIdx.x % tiles_i, bj = blockIdx.x / tiles_i;
  const int i0 = bi * 64, j0 = bj * 64;
  const int r_begin = blockIdx.y * r_chunk;
  const int r_end = (r_begin + r_chunk < R) ? r_begin + r_chunk : R;
  const int nslab = (r_end - r_begin) / kDmaBR;

  // per-lane source pointers of this wave's 1 KiB piece of each slab (advance by one slab per issue)
  const int p = wave * 64 + lane;                      // 16-byte position inside the 4 KiB slab image
  const float* psrc;
  int64_t pstep;
  {
    if (PX) { const int r = p >> 4, x4 = (p & 15) * 4; int xx = i0 + x4; if (xx > P.X - 4) xx = P.X - 4 < 0 ? 0 : P.X - 4;
              psrc = P.p + (int64_t)(r_begin + r) * P.ld + xx; pstep = (int64_t)kDmaBR * P.ld; }
    else    { int x = i0 + (p >> 2); if (x > P.X - 1) x = P.X - 1; const int c = p & 3;
              psrc = P.p + (int64_t)x * P.ld + r_begin + 4 * c; pstep = kDmaBR; }
  }
  const float* qsrc;
  int64_t qstep;
  {
    if (QX) { const int r = p >> 4, x4 = (p & 15) * 4; int xx = j0 + x4; if (xx > Q.X - 4) xx = Q.X - 4 < 0 ? 0 : Q.X - 4;
              qsrc = Q.p + (int64_t)(r_begin + r) * Q.ld + xx; qstep = (int64_t)kDmaBR * Q.ld; }
    else    { int x = j0 + (p >> 2); if (x > Q.X - 1) x = Q.X - 1; const int c = p & 3;
              qsrc = Q.p + (int64_t)x * Q.ld + r_begin + 4 * c; qstep = kDmaBR; }
  }
  const unsigned pl = lds_addr(&Ps[0][0]) + wave * 1024, ql = lds_addr(&Qs[0][0]) + wave * 1024;

  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_cs = QX && (colsum != nullptr) && (bi == 0) && (tid < 64);   // column sums of the Q slabs (bias gradient)
  float csum = 0.f;

  // prologue: slabs 0 .. NS-2 in flight
#pragma unroll
  for (int s = 0; s < kDmaNS - 1; ++s) {
    if (s < nslab) {
      dma16(psrc, pl + s * (64 * kDmaBR * 4));
      dma16(qsrc, ql + s * (64 * kDmaBR * 4));
      psrc += pstep; qsrc += qstep;
    }
  }
  for (int it = 0; it < nslab; ++it) {
    // slab `it` has landed when at most the (up to NS-2) younger slabs of THIS wave are still outstanding
    const int younger = (nslab - 1 - it < kDmaNS - 2) ? nslab - 1 - it : kDmaNS - 2;
    if (younger >= 2) wait_vmcnt<4>(); else if (younger == 1) wait_vmcnt<2>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                         // every wave's piece of slab `it` is in LDS; slab it-1 is free
    if (it + kDmaNS - 1 < nslab) {                        // refill the buffer consumed in the previous iteration
      const int buf = (it + kDmaNS - 1) % kDmaNS;
      dma16(psrc, pl + buf * (64 * kDmaBR * 4));
      dma16(qsrc, ql + buf * (64 * kDmaBR * 4));
      psrc += pstep; qsrc += qstep;
    }
    const float* Pb = Ps[it % kDmaNS];
    const float* Qb = Qs[it % kDmaNS];
    if (do_cs) {
#pragma unroll
      for (int r = 0; r < kDmaBR; ++r) csum += Qb[r * 64 + tid];
    }
    float4 pv[4], qv[4];
    if (PX) {
#pragma unroll
      for (int s = 0; s < 4; ++s) pv[s] = *reinterpret_cast<const float4*>(Pb + (4 * lr + s) * 64 + 4 * li);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) pv[t] = *reinterpret_cast<const float4*>(Pb + (16 * t + li) * kDmaBR + 4 * lr);
    }
    if (QX) {
#pragma unroll
      for (int s = 0; s < 4; ++s) qv[s] = *reinterpret_cast<const float4*>(Qb + (4 * lr + s) * 64 + 4 * li);
    } else {
      qv[0] = *reinterpret_cast<const float4*>(Qb + (16 * wave + li) * kDmaBR + 4 * lr);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float b;
      if (QX) {
        const float4 q = qv[s];
        b = wave == 0 ? q.x : (wave == 1 ? q.y : (wave == 2 ? q.z : q.w));
      } else {
        b = s == 0 ? qv[0].x : (s == 1 ? qv[0].y : (s == 2 ? qv[0].z : qv[0].w));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float a;
        if (PX) { const float4 v = pv[s]; a = t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w)); }
        else    { const float4 v = pv[t]; a = s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  if (do_cs && j0 + tid < J) atomicAdd(colsum + j0 + tid, csum * epi.block_scale());
  // epilogue
  const int j = QX ? j0 + 4 * li + wave : j0 + 16 * wave + li;
  if (j < J) {
    if (PX) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = i0 + 16 * lr + 4 * v;
        if (i < I) epi(i, j, f32x4{acc[0][v], acc[1][v], acc[2][v], acc[3][v]}, (I - i < 4) ? I - i : 4);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = i0 + 16 * t + 4 * lr;
        if (i < I) epi(i, j, acc[t], (I - i < 4) ? I - i : 4);
      }
    }
  }
}

// shape gate: 16-byte aligned bases, leading dims multiples of 4, reduction a multiple of the slab depth
inline bool dma_ok(const DmaOperand& P, bool px, const DmaOperand& Q, bool qx, int64_t R, int r_chunk) {
  auto ok = [](const DmaOperand& o, bool x) {
    return ((reinterpret_cast<uintptr_t>(o.p) & 15) == 0) && (o.ld % 4 == 0) && (!x || (o.X >= 4 && o.X % 4 == 0));
  };
  return ok(P, px) && ok(Q, qx) && (R % kDmaBR == 0) && (r_chunk % kDmaBR == 0);
}

template <bool PX, bool QX, class Epi>
inline hipError_t launch_gemm_dma(DmaOperand P, DmaOperand Q, Epi epi, int I, int64_t J, int R, int splits, hipStream_t stream,
                                  float* colsum = nullptr) {
  if (I <= 0 || J <= 0 || R <= 0) return hipSuccess;
  if (splits < 1) splits = 1;
  int r_chunk = ceil_div(ceil_div(R, splits), kDmaBR) * kDmaBR;
  splits = ceil_div(R, r_chunk);
  const int tiles_i = ceil_div(I, 64);
  const int64_t blocks = (int64_t)tiles_i * ceil_div(J, 64);
  hipLaunchKernelGGL((gemm_dma_kernel<PX, QX, Epi>), dim3((unsigned)blocks, splits), dim3(256), 0, stream, P, Q, epi, I, (int)J, R,
                     r_chunk, tiles_i, colsum);
  return hipGetLastError();
}

}  // namespace micf
