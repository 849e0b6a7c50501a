// block_fused.h -- device building blocks of the window-local fused transformer-block kernels (block_fused.hip).
//
// A (Cross)TransformerBlock3D (MS.py:277-524) is local to a 2x2x2 window once the deformable sampling has produced the K/V
// source: LayerNorm, the q/kv/proj/fc1/fc2 linears, GELU, the residual adds are per token and attention is per window.  A
// workgroup therefore takes a tile of TM = 16 * TJ tokens (TM / 8 whole windows), keeps every intermediate of the block in
// LDS, and streams only the WEIGHTS from L2 / HBM:
//   * weights arrive through the gfx950 LDS-DMA engine (`global_load_lds_dwordx4`, gemm_dma.h) as [64][16] slab images in a
//     4-deep ring, one counted `s_waitcnt vmcnt` + one `s_barrier` per slab;
//   * the activation operand of every GEMM is an LDS-resident tile [TM][K + 4] (row = token, natural k order), read as one
//     ds_read_b128 per 4 k-steps with the k-permutation of gemm_dma.h (lane group lr supplies k = 4*lr + s in step s);
//   * results go back to an LDS tile (store / accumulate / scale / GELU'-combine epilogues): the GEMM loop issues NO vector
//     memory operation besides its DMAs, so the counted vmcnt waits stay exact; HBM traffic happens in row-coalesced
//     element-wise passes between the GEMM phases.
// MFMA is v_mfma_f32_16x16x4_f32 (exact fp32: bitwise a k-ordered fmaf chain) or, in bf16 mode, v_mfma_f32_16x16x32_bf16
// on operands rounded to bf16 at the fragment read (fp32 LDS tiles and weights, fp32 accumulation).
#pragma once
#include "common.h"
#include "gemm_dma.h"

namespace micf {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // round-to-nearest-even, a in the low half
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7FFFu + ((ua >> 16) & 1u);
  ub += 0x7FFFu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xFFFF0000u);
}
__device__ __forceinline__ bf16x8 to_bf16x8(const float4& lo, const float4& hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 r = {pack_bf16(lo.x, lo.y), pack_bf16(lo.z, lo.w), pack_bf16(hi.x, hi.y), pack_bf16(hi.z, hi.w)};
  return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ void wait_dma1(int younger) {     // one DMA per slab per wave
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
  }
}

constexpr int kFusedNS = 4;            // weight-slab ring depth
constexpr int kFusedRing = kFusedNS * 1024;   // floats

enum { EPI_STORE = 0, EPI_ACC = 1, EPI_STORE_SCALE = 2, EPI_ACC_SCALE = 3, EPI_GELU_GRAD = 4 };

template <int EPI>
__device__ __forceinline__ void epi_apply(float* o, const float4 v, float rs) {
  float4 r;
  if constexpr (EPI == EPI_STORE) r = v;
  else if constexpr (EPI == EPI_STORE_SCALE) r = make_float4(rs * v.x, rs * v.y, rs * v.z, rs * v.w);
  else {
    const float4 c = *reinterpret_cast<const float4*>(o);
    if constexpr (EPI == EPI_ACC) r = make_float4(c.x + v.x, c.y + v.y, c.z + v.z, c.w + v.w);
    else if constexpr (EPI == EPI_ACC_SCALE) r = make_float4(c.x + rs * v.x, c.y + rs * v.y, c.z + rs * v.z, c.w + rs * v.w);
    else r = make_float4(rs * v.x * gelu_grad_f(c.x), rs * v.y * gelu_grad_f(c.y), rs * v.z * gelu_grad_f(c.z), rs * v.w * gelu_grad_f(c.w));
  }
  *reinterpret_cast<float4*>(o) = r;
}

// One GEMM phase of a fused block kernel:
//   O[t][x] (op)= sum_r A(x, r) * Bs[t][r]      t < 16 * TJ, x < X (X % 16 == 0), r < R (R % 16 == 0)
//   AX == false: A(x, r) = Wg[x * ld + r]   (nn.Linear forward: x = output feature n, r = input feature k)
//   AX == true : A(x, r) = Wg[r * ld + x]   (data gradient: x = input feature k, r = output feature n)
// Bs / Os are LDS tiles with row strides SB / SO (floats, multiples of 4); rowscale (LDS, [16*TJ]) is used by the *_SCALE and
// GELU_GRAD epilogues.  All 256 threads call it together.  On return every DMA has landed, Os is complete and visible to the
// workgroup, and the ring may be reused.
template <int TJ, bool AX, int EPI, bool BF16>
__device__ __forceinline__ void gemm_phase(const float* __restrict__ Wg, int ld, int X, int R, const float* Bs, int SB,
                                           float* Os, int SO, const float* rowscale, float* ring) {
  constexpr int NT = TJ;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lr = lane >> 4;
  const int wj = wave % TJ, wi = wave / TJ;
  const int nxb = (X + 63) >> 6, nsl = R >> 4, total = nxb * nsl;
  const int p = wave * 64 + lane;                       // 16-byte position inside the 4 KiB slab image
  const unsigned ldsbase = lds_addr(ring) + wave * 1024;

  auto issue = [&](int step) {
    const int xb = step / nsl, sl = step - xb * nsl;
    const float* src;
    if constexpr (AX) {
      const int r = p >> 4;
      int xx = xb * 64 + (p & 15) * 4;
      if (xx > X - 4) xx = X - 4;
      src = Wg + (int64_t)(sl * 16 + r) * ld + xx;
    } else {
      int x = xb * 64 + (p >> 2);
      if (x > X - 1) x = X - 1;
      src = Wg + (int64_t)x * ld + sl * 16 + 4 * (p & 3);
    }
    dma16(src, ldsbase + (unsigned)(step % kFusedNS) * 4096u);
  };

#pragma unroll
  for (int s = 0; s < kFusedNS - 1; ++s)
    if (s < total) issue(s);

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int trow = 16 * wj + li;                        // token row of this lane's B fragment and of its outputs
  const float* brow = Bs + trow * SB + 4 * lr;
  float4 qprev = make_float4(0.f, 0.f, 0.f, 0.f);       // bf16: the even slab's fragments wait for the odd slab
  float4 aprev[NT];
  float aprevx[4][NT];

  for (int it = 0; it < total; ++it) {
    const int rem = total - 1 - it;
    wait_dma1(rem < kFusedNS - 2 ? rem : kFusedNS - 2);
    __builtin_amdgcn_s_barrier();                       // every wave's piece of slab `it` is in LDS; slab it-1 is free
    if (it + kFusedNS - 1 < total) issue(it + kFusedNS - 1);
    const int xb = it / nsl, sl = it - xb * nsl;
    const float* Ab = ring + (it % kFusedNS) * 1024;
    const float4 qv = *reinterpret_cast<const float4*>(brow + sl * 16);
    const float qs[4] = {qv.x, qv.y, qv.z, qv.w};
    if constexpr (!AX) {
      float4 pv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) pv[t] = *reinterpret_cast<const float4*>(Ab + (16 * (wi * NT + t) + li) * 16 + 4 * lr);
      if constexpr (!BF16) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float a = s == 0 ? pv[t].x : (s == 1 ? pv[t].y : (s == 2 ? pv[t].z : pv[t].w));
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qs[s], acc[t], 0, 0, 0);
          }
      } else {
        const bool odd = sl & 1, last = sl == nsl - 1;
        if (odd || last) {        // k = 32 per MFMA: (even slab, odd slab) pairs; an unpaired last slab is padded with zeros
          const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          const bf16x8 bq = odd ? to_bf16x8(qprev, qv) : to_bf16x8(qv, z);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const bf16x8 ba = odd ? to_bf16x8(aprev[t], pv[t]) : to_bf16x8(pv[t], z);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bq, acc[t], 0, 0, 0);
          }
        } else {
          qprev = qv;
#pragma unroll
          for (int t = 0; t < NT; ++t) aprev[t] = pv[t];
        }
      }
    } else {
      // lane li holds NT consecutive x of its wave's 16*NT-wide slice: x = 16*NT*wi + NT*li + t  (tile t)
      float av[4][NT];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* ap = Ab + (4 * lr + s) * 64 + 16 * NT * wi + NT * li;
        if constexpr (NT == 4) { const float4 v = *reinterpret_cast<const float4*>(ap); av[s][0] = v.x; av[s][1] = v.y; av[s][2] = v.z; av[s][3] = v.w; }
        else if constexpr (NT == 2) { const float2 v = *reinterpret_cast<const float2*>(ap); av[s][0] = v.x; av[s][1] = v.y; }
        else av[s][0] = *ap;
      }
      if constexpr (!BF16) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], qs[s], acc[t], 0, 0, 0);
      } else {
        const bool odd = sl & 1, last = sl == nsl - 1;
        if (odd || last) {
          const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          const bf16x8 bq = odd ? to_bf16x8(qprev, qv) : to_bf16x8(qv, z);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float4 cur = make_float4(av[0][t], av[1][t], av[2][t], av[3][t]);
            const float4 prv = make_float4(aprevx[0][t], aprevx[1][t], aprevx[2][t], aprevx[3][t]);
            const bf16x8 ba = odd ? to_bf16x8(prv, cur) : to_bf16x8(cur, z);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bq, acc[t], 0, 0, 0);
          }
        } else {
          qprev = qv;
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) aprevx[s][t] = av[s][t];
        }
      }
    }
    if (sl == nsl - 1) {                                // the 64-wide x block `xb` is complete: epilogue into the LDS tile
      const float rs = (EPI >= EPI_STORE_SCALE) ? rowscale[trow] : 1.f;
      float* orow = Os + trow * SO;
      if constexpr (!AX) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int x = xb * 64 + 16 * (wi * NT + t) + 4 * lr;
          if (x < X) epi_apply<EPI>(orow + x, make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]), rs);
          acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      } else {
        if constexpr (NT == 4) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int x = xb * 64 + 16 * lr + 4 * v;
            if (x < X) epi_apply<EPI>(orow + x, make_float4(acc[0][v], acc[1][v], acc[2][v], acc[3][v]), rs);
          }
        } else if constexpr (NT == 2) {
          const int x = xb * 64 + 32 * wi + 8 * lr;
          if (x < X) epi_apply<EPI>(orow + x, make_float4(acc[0][0], acc[1][0], acc[0][1], acc[1][1]), rs);
          if (x + 4 < X) epi_apply<EPI>(orow + x + 4, make_float4(acc[0][2], acc[1][2], acc[0][3], acc[1][3]), rs);
        } else {
          const int x = xb * 64 + 16 * wi + 4 * lr;
          if (x < X) epi_apply<EPI>(orow + x, make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]), rs);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  __syncthreads();
}

// ---- tile geometry: TM = 16 * TJ tokens = TM / 8 whole 2x2x2 windows; rows are window-major (row = 8 * window + 4*id + 2*ih + iw)
struct TileGeo {
  int B, D, H, W;                 // token grid (all even)
  FastDiv f_nww, f_nwh, f_nwd;    // windows per axis
  FastDiv f_rps;                  // tokens per sample
  int64_t T;                      // tokens per group
  int nwin;                       // windows per group
  __device__ __forceinline__ int token(int win, int i) const {
    uint32_t q, xw, xh, xd;
    f_nww.divmod((uint32_t)win, q, xw);
    f_nwh.divmod(q, q, xh);
    f_nwd.divmod(q, q, xd);
    const int b = (int)q;
    return ((b * D + 2 * (int)xd + (i >> 2)) * H + 2 * (int)xh + ((i >> 1) & 1)) * W + 2 * (int)xw + (i & 1);
  }
};
inline TileGeo make_tile_geo(int B, int D, int H, int W) {
  TileGeo g;
  g.B = B; g.D = D; g.H = H; g.W = W;
  g.f_nww = FastDiv((uint32_t)(W / 2)); g.f_nwh = FastDiv((uint32_t)(H / 2)); g.f_nwd = FastDiv((uint32_t)(D / 2));
  g.f_rps = FastDiv((uint32_t)(D * H * W));
  g.T = (int64_t)B * D * H * W;
  g.nwin = (int)(g.T / 8);
  return g;
}

__device__ __forceinline__ float sum16(float v) {      // sum over the 16-lane group (rows are handled by 16 lanes each)
  v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
  return v;
}
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4g(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

}  // namespace micf
