// block_fused.h -- device building blocks of the window-local fused transformer-block kernels (block_fused.hip).
//
// A (Cross)TransformerBlock3D (MS.py:277-524) is local to a 2x2x2 window once the deformable sampling has produced the K/V
// source: LayerNorm, the q/kv/proj/fc1/fc2 linears, GELU, the residual adds are per token and attention is per window.  A
// workgroup therefore takes a tile of TM = 16 * TJ tokens (TM / 8 whole windows), keeps every intermediate of the block in
// LDS, and streams only the WEIGHTS from L2 / HBM:
//   * a wave owns whole 16-wide output-feature tiles, so no weight is shared between waves: each lane loads its MFMA
//     A-fragments straight from L2 / HBM into registers, 8 slabs ahead (no LDS staging, no barrier inside a GEMM phase);
//   * the activation operand of every GEMM is an LDS-resident tile [TM][K + 4] (row = token, natural k order), read as one
//     ds_read_b128 per 4 k-steps with the k-permutation of gemm_dma.h (lane group lr supplies k = 4*lr + s in step s);
//   * results go back to an LDS tile (store / accumulate / scale / GELU'-combine epilogues); HBM traffic of the activations
//     happens in row-coalesced element-wise passes between the GEMM phases.
// MFMA is v_mfma_f32_16x16x4_f32 (exact fp32: bitwise a k-ordered fmaf chain) or, in bf16 mode, v_mfma_f32_16x16x32_bf16
// on operands rounded to bf16 at the fragment read (fp32 LDS tiles and weights, fp32 accumulation).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_dma.h"
#include "sampler_common.h"

namespace micf {

// Workgroup barrier of the fused block kernels.  Everything the threads of a tile exchange goes through LDS; global memory is
// only read (inputs) and written (outputs, each address by one thread), never re-read inside the kernel.  __syncthreads()
// would also drain the vector-memory counter (s_waitcnt vmcnt(0)): every output store and every early-issued input load would
// be waited for at the next barrier, i.e. ~20 exposed HBM round trips per tile.  This barrier orders LDS only; the compiler still
// waits for a load's registers where they are consumed.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Weight warm-up of the tile kernels.  The workgroups of a launch run their GEMM phases in step (128 workgroups at 8^3, all
// resident), so every phase's FIRST weight read is an L2 miss that the whole launch waits for: ~6 (forward) / ~8 (backward) exposed
// round trips per launch, and an XCD's L2 starts every kernel empty.  Instead the first workgroups of each XCD (blockIdx & 7 = the
// XCD, blockIdx >> 3 = its slot there) request the XCD's whole weight set at the top of the kernel, one 64-byte piece per thread
// and round, in the order the phases will read the five matrices (N0 .. N4 = their sizes in units of C * C elements).  The caller
// drops the returned word after its first product -- those loads return behind these, so nothing waits for the warm-up itself.
// Measured (base, 8^3, bf16): block_bwd 30.5 -> 25.2 us, block_fwd 27.4 -> 25.6 us per launch.
template <int C, int NTHR, int N0, int N1, int N2, int N3, int N4, class WT>
__device__ __forceinline__ uint32_t warm_weights(unsigned bid, int tid, const WT* w0, const WT* w1, const WT* w2, const WT* w3, const WT* w4) {
  constexpr int P1 = C * C * (int)sizeof(WT) / 64;      // 64-byte pieces of a [C, C] matrix
  constexpr int E0 = N0 * P1, E1 = E0 + N1 * P1, E2 = E1 + N2 * P1, E3 = E2 + N3 * P1, E4 = E3 + N4 * P1;
  const int nslot = ((int)gridDim.x + 7) >> 3;
  uint32_t word = 0;
  for (int piece = (int)(bid >> 3) * NTHR + tid; piece < E4; piece += nslot * NTHR) {
    const char* base = piece < E0 ? reinterpret_cast<const char*>(w0) : piece < E1 ? reinterpret_cast<const char*>(w1) - (int64_t)E0 * 64
                     : piece < E2 ? reinterpret_cast<const char*>(w2) - (int64_t)E1 * 64
                     : piece < E3 ? reinterpret_cast<const char*>(w3) - (int64_t)E2 * 64 : reinterpret_cast<const char*>(w4) - (int64_t)E3 * 64;
    word ^= *reinterpret_cast<const uint32_t*>(base + (int64_t)piece * 64);
  }
  return word;
}

// ---- epilogues of a GEMM phase: called with the LDS destination of 4 consecutive outputs x..x+3 of token row `trow`.
// They touch LDS only (the GEMM loop must not issue vector memory operations besides its DMAs).
struct EpiStore {
  __device__ __forceinline__ void operator()(float* o, int, int, const float4& v) const { *reinterpret_cast<float4*>(o) = v; }
};
struct EpiAcc {
  __device__ __forceinline__ void operator()(float* o, int, int, const float4& v) const {
    const float4 c = *reinterpret_cast<const float4*>(o);
    *reinterpret_cast<float4*>(o) = make_float4(c.x + v.x, c.y + v.y, c.z + v.z, c.w + v.w);
  }
};
struct EpiBias {              // + bias[x] (bias vector in LDS)
  const float* bias;
  __device__ __forceinline__ void operator()(float* o, int x, int, const float4& v) const {
    const float4 b = *reinterpret_cast<const float4*>(bias + x);
    *reinterpret_cast<float4*>(o) = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
  }
};
struct EpiStoreScale {        // rowscale[trow] * v
  const float* rowscale;
  __device__ __forceinline__ void operator()(float* o, int, int trow, const float4& v) const {
    const float rs = rowscale[trow];
    *reinterpret_cast<float4*>(o) = make_float4(rs * v.x, rs * v.y, rs * v.z, rs * v.w);
  }
};
struct EpiAccScale {          // += rowscale[trow] * v
  const float* rowscale;
  __device__ __forceinline__ void operator()(float* o, int, int trow, const float4& v) const {
    const float rs = rowscale[trow];
    const float4 c = *reinterpret_cast<const float4*>(o);
    *reinterpret_cast<float4*>(o) = make_float4(c.x + rs * v.x, c.y + rs * v.y, c.z + rs * v.z, c.w + rs * v.w);
  }
};
template <bool FAST>
struct EpiGeluGrad {          // rowscale[trow] * v * GELU'(what the tile holds: the saved pre-activation)
  const float* rowscale;
  __device__ __forceinline__ void operator()(float* o, int, int trow, const float4& v) const {
    const float rs = rowscale[trow];
    const float4 c = *reinterpret_cast<const float4*>(o);
    *reinterpret_cast<float4*>(o) = make_float4(rs * v.x * gelu_grad_t<FAST>(c.x), rs * v.y * gelu_grad_t<FAST>(c.y),
                                                 rs * v.z * gelu_grad_t<FAST>(c.z), rs * v.w * gelu_grad_t<FAST>(c.w));
  }
};

// One GEMM phase of a fused block kernel:
//   O[t][x] (epi)= sum_r A(x, r) * B[t][r]      t < 16 * TG, r < R = 16 * NSL * NK
// over one or two weight SEGMENTS (x in [0, X0) from W0 against activation tile Bs0, x in [X0, X0 + X1) from W1 against Bs1;
// X0, X1 multiples of 16, same R and leading dimension LD) -- q | kv in one phase although they are separate tensors:
//   A(x, r) = W[x * LD + r]: rows of W are the outputs.  nn.Linear forward: W = the weight [N, K]; data gradients
//   (dA = dY W): W = the TRANSPOSED weight [K, N] (micf_weight_prep_grouped, once per step), so both directions stream
//   contiguous 64-byte row pieces.
// The 16-wide x tiles are dealt round-robin to the NW waves of the workgroup.  Nothing about a tile's weights is shared between waves, so
// they do not go through LDS at all: every lane loads ITS MFMA A-fragments straight from L2 / HBM into registers (one
// 16-byte load per 16-deep slab = A(x = li, r = 4 lr .. 4 lr + 3)).  The unit of work is
// (tile, k-chunk of NSL slabs); NSL, NK and LD are compile-time, so a unit is ONE base-pointer computation plus NSL loads at
// immediate offsets, fully unrolled, and the loop is software-pipelined in registers: the fragments of unit u + 1 are in
// flight while unit u feeds 4 * TG * NSL MFMAs.  (The first version streamed weights through an LDS-DMA ring with a barrier
// per slab: at 16-32 tokens per workgroup the per-slab bookkeeping -- ~80 instructions for 4-8 MFMAs -- was the bound.)
// The activation operand Bs is an LDS tile [16 TG][SB] read as one ds_read_b128 per token group and slab (k-permutation:
// lane group lr supplies k = 4 lr + s in step s, both operands alike); a wave writes only its own x columns of the LDS output
// tile Os.  Two accumulator chains per token group (even / odd k-steps) keep the fp32 MFMA pipe at its issue rate.
// All 64 * NW threads call it together; on return Os is complete and visible to the workgroup.
template <int TG, int NSL, int NK, int LD, int NW, class Epi>
__device__ __forceinline__ void gemm_phase_f32w(const float* __restrict__ W0, int X0, const float* Bs0, const float* __restrict__ W1,
                                           int X1, const float* Bs1, int SB, float* Os, int SO, const Epi epi) {
  // The units of a phase are dealt to waves 0, 1, ...: with few units per phase the low waves always get the extra one, and
  // wave w of every workgroup on a CU sits on SIMD w.  Rotating the deal by a per-workgroup offset spreads that over the SIMDs.
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (__builtin_amdgcn_readfirstlane(tid >> 6) + (int)((blockIdx.x * 2654435761u) >> 20)) & (NW - 1);
  const int li = lane & 15, lr = lane >> 4;
  const int nt0 = X0 >> 4, ntiles = nt0 + (X1 >> 4);
  const int mytiles = (ntiles - wave + NW - 1) / NW;       // tiles wave, wave + NW, ...
  const int nunits = mytiles * NK;

  struct Frag { float4 v[NSL]; };
  // fragments of unit `u` (clamped to the wave's last unit: loads are unconditional so the pipelined loop stays straight-line)
  auto load_unit = [&](int u, Frag& f) {
    if (u > nunits - 1) u = nunits - 1;
    const int tile = wave + NW * (NK == 1 ? u : u / NK), kc = NK == 1 ? 0 : u % NK;
    const bool seg1 = tile >= nt0;
    const float* Wb = seg1 ? W1 : W0;
    const int xt = (seg1 ? tile - nt0 : tile) * 16;
    // K16-blocked like the bf16 copies (see gemm_phase_bf16w): one load instruction = 16 rows x 16 k = 1 KB of consecutive addresses
    const float* p = Wb + (int64_t)xt * LD + kc * NSL * 256 + li * 16 + 4 * lr;
#pragma unroll
    for (int s = 0; s < NSL; ++s) f.v[s] = *reinterpret_cast<const float4*>(p + 256 * s);
  };

  f32x4 acc0[TG], acc1[TG];
#pragma unroll
  for (int g = 0; g < TG; ++g) { acc0[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[g] = acc0[g]; }
  auto compute_unit = [&](int u, const Frag& f) {
    const int tile = wave + NW * (NK == 1 ? u : u / NK), kc = NK == 1 ? 0 : u % NK;
    const float* brow = ((tile >= nt0) ? Bs1 : Bs0) + li * SB + kc * NSL * 16 + 4 * lr;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const float4 av = f.v[s];
      float4 qv[TG];
#pragma unroll
      for (int g = 0; g < TG; ++g) qv[g] = *reinterpret_cast<const float4*>(brow + g * 16 * SB + 16 * s);
#pragma unroll
      for (int g = 0; g < TG; ++g) {
        acc0[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, qv[g].x, acc0[g], 0, 0, 0);
        acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, qv[g].y, acc1[g], 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < TG; ++g) {
        acc0[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, qv[g].z, acc0[g], 0, 0, 0);
        acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, qv[g].w, acc1[g], 0, 0, 0);
      }
    }
    if (kc == NK - 1) {                                 // tile complete: epilogue into the LDS output tile (own columns only)
      const int x = tile * 16 + 4 * lr;                 // output feature index over both segments
#pragma unroll
      for (int g = 0; g < TG; ++g) {
        const int trow = 16 * g + li;
        epi(Os + trow * SO + x, x, trow, make_float4(acc0[g][0] + acc1[g][0], acc0[g][1] + acc1[g][1], acc0[g][2] + acc1[g][2],
                                                     acc0[g][3] + acc1[g][3]));
        acc0[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[g] = acc0[g];
      }
    }
  };

  if (nunits > 0) {
    Frag fa, fb;
    load_unit(0, fa);
    for (int u = 0; u < nunits; u += 2) {
      load_unit(u + 1, fb);
      compute_unit(u, fa);
      load_unit(u + 2, fa);
      if (u + 1 < nunits) compute_unit(u + 1, fb);
    }
  }
  lds_barrier();
}

// The same phase with bf16 WEIGHTS (shadow copies written once per step by micf_weight_prep_grouped: half the bytes and half
// the load instructions of the weight stream, which is what bounds the small-token stages; no conversion on the A side).
// The copies are K16-BLOCKED: [row / 16][k / 16][row % 16][k % 16], so the 64 lanes of one load instruction (16 rows x 32 k)
// cover 1 KB of consecutive addresses = 8 whole cache lines.  Row-major, the same instruction touched 16 lines (64 bytes of
// each, 2 C bytes apart) and the weight stream ran at a third of this rate (8^3 stage forward: 47.5 -> 31 us).  A weight
// pointer advanced by r rows moves r * LD elements as before; advanced by k columns it moves 16 k elements.
// MFMA 16x16x32: lane group lr supplies k = 8 lr .. 8 lr + 7 of every 32-deep chunk (the natural bf16 mapping): one 16-byte
// load of 8 bf16 for the A fragment, two ds_read_b128 of fp32 activations (rounded to bf16, RNE) for B.  A chunk of K = 16 NSL
// with NSL odd ends in a 16-deep half chunk: lane groups 2, 3 feed zeros on both sides.
template <int TG, int NSL, int NK, int LD, int NW, class Epi>
__device__ __forceinline__ void gemm_phase_bf16w(const uint16_t* __restrict__ W0, int X0, const float* Bs0, const uint16_t* __restrict__ W1,
                                                 int X1, const float* Bs1, int SB, float* Os, int SO, const Epi epi) {
  constexpr int N32 = NSL / 2, REM = NSL & 1, NF = N32 + REM;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // The units of a phase are dealt to waves 0, 1, ...: with few units per phase the low waves always get the extra one, and
  // wave w of every workgroup on a CU sits on SIMD w.  Rotating the deal by a per-workgroup offset spreads that over the SIMDs.
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (__builtin_amdgcn_readfirstlane(tid >> 6) + (int)((blockIdx.x * 2654435761u) >> 20)) & (NW - 1);
  const int li = lane & 15, lr = lane >> 4;
  const int nt0 = X0 >> 4, ntiles = nt0 + (X1 >> 4);
  const int mytiles = (ntiles - wave + NW - 1) / NW;
  const int nunits = mytiles * NK;
  const int lrh = lr & 1;                               // half chunk: lane groups 2, 3 re-read a valid address and drop the value

  struct Frag { u32x4 v[NF]; };
  auto load_unit = [&](int u, Frag& f) {
    if (u > nunits - 1) u = nunits - 1;
    const int tile = wave + NW * (NK == 1 ? u : u / NK), kc = NK == 1 ? 0 : u % NK;
    const bool seg1 = tile >= nt0;
    const uint16_t* Wb = seg1 ? W1 : W0;
    const int xt = (seg1 ? tile - nt0 : tile) * 16;
    // K16-blocked: element (row, k) at (row / 16) * 16 LD + (k / 16) * 256 + (row % 16) * 16 + k % 16
    const uint16_t* p = Wb + (int64_t)xt * LD + kc * NSL * 256 + li * 16 + 8 * lrh;
#pragma unroll
    for (int j = 0; j < N32; ++j) f.v[j] = *reinterpret_cast<const u32x4*>(p + (2 * j + (lr >> 1)) * 256);
    if constexpr (REM) f.v[N32] = *reinterpret_cast<const u32x4*>(p + 2 * N32 * 256);
  };

  f32x4 acc[TG];
#pragma unroll
  for (int g = 0; g < TG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The activation fragments of a (source, k chunk) are read from LDS and rounded to bf16 ONCE per wave and phase and reused
  // by every unit of that chunk (they were re-read per unit: two ds_read_b128 + four conversions per MFMA).
  bf16x8 actf[NF][TG];
  const float* act_src = nullptr;
  int act_kc = -1;
  auto compute_unit = [&](int u, const Frag& f) {
    const int tile = wave + NW * (NK == 1 ? u : u / NK), kc = NK == 1 ? 0 : u % NK;
    const float* src = (tile >= nt0) ? Bs1 : Bs0;
    if (src != act_src || kc != act_kc) {               // (wave-uniform)
      act_src = src; act_kc = kc;
      const float* brow = src + li * SB + kc * NSL * 16;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const bool half = REM && j == N32;
        const int koff = 32 * j + 8 * (half ? lrh : lr);
#pragma unroll
        for (int g = 0; g < TG; ++g) {
          const float* bp = brow + g * 16 * SB + koff;
          float4 lo = *reinterpret_cast<const float4*>(bp), hi = *reinterpret_cast<const float4*>(bp + 4);
          if (half && lr >= 2) { lo = make_float4(0.f, 0.f, 0.f, 0.f); hi = lo; }
          actf[j][g] = to_bf16x8(lo, hi);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const bool half = REM && j == N32;
      bf16x8 ba = __builtin_bit_cast(bf16x8, f.v[j]);
      if (half && lr >= 2) ba = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
#pragma unroll
      for (int g = 0; g < TG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, actf[j][g], acc[g], 0, 0, 0);
    }
    if (kc == NK - 1) {
      const int x = tile * 16 + 4 * lr;
#pragma unroll
      for (int g = 0; g < TG; ++g) {
        const int trow = 16 * g + li;
        epi(Os + trow * SO + x, x, trow, make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]));
        acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  if (nunits > 0 && nunits <= 3) {
    // few units per wave (16 waves at C = 192): request them ALL before the first MFMA -- one L2 round trip per phase instead of
    // one per unit (a unit's 6-12 MFMAs are far too short to hide the next unit's loads behind them)
    Frag fa, fb, fc;
    load_unit(0, fa);
    load_unit(1, fb);
    load_unit(2, fc);
    compute_unit(0, fa);
    if (nunits > 1) compute_unit(1, fb);
    if (nunits > 2) compute_unit(2, fc);
  } else if (nunits > 0) {
    Frag fa, fb;
    load_unit(0, fa);
    for (int u = 0; u < nunits; u += 2) {
      load_unit(u + 1, fb);
      compute_unit(u, fa);
      load_unit(u + 2, fa);
      if (u + 1 < nunits) compute_unit(u + 1, fb);
    }
  }
  lds_barrier();
}

// dispatch on the weight type: float (exact fp32 MFMA) or uint16_t (bf16 shadow weights, bf16 MFMA)
template <int TG, int NSL, int NK, int LD, int NW, bool BF16, class Epi>
__device__ __forceinline__ void gemm_phase(const float* W0, int X0, const float* Bs0, const float* W1, int X1, const float* Bs1, int SB,
                                           float* Os, int SO, const Epi epi) {
  gemm_phase_f32w<TG, NSL, NK, LD, NW, Epi>(W0, X0, Bs0, W1, X1, Bs1, SB, Os, SO, epi);
}
template <int TG, int NSL, int NK, int LD, int NW, bool BF16, class Epi>
__device__ __forceinline__ void gemm_phase(const uint16_t* W0, int X0, const float* Bs0, const uint16_t* W1, int X1, const float* Bs1,
                                           int SB, float* Os, int SO, const Epi epi) {
  gemm_phase_bf16w<TG, NSL, NK, LD, NW, Epi>(W0, X0, Bs0, W1, X1, Bs1, SB, Os, SO, epi);
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
  __device__ __forceinline__ void coords(int win, int i, int& b, int& d, int& h, int& w) const {
    uint32_t q, xw, xh, xd;
    f_nww.divmod((uint32_t)win, q, xw);
    f_nwh.divmod(q, q, xh);
    f_nwd.divmod(q, q, xd);
    b = (int)q; d = 2 * (int)xd + (i >> 2); h = 2 * (int)xh + ((i >> 1) & 1); w = 2 * (int)xw + (i & 1);
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

// floats of the attention backward's P / dS exchange: 16 per thread that takes part (one per attention row and head, whole
// windows per batch)
constexpr int block_bwd_scratch_floats(int TM, int heads, int nthr) { return (TM * heads < nthr ? TM * heads : nthr) * 16; }
// C = 48: the attention backward needs the whole register file (three workgroups per CU), so the LayerNorm-1 inputs requested
// at the top of the kernel wait in LDS meanwhile: [TM][C] rows + mean / rstd slots per thread
constexpr int block_bwd_park_floats(int TM, int C, int nthr) { return C <= 48 ? TM * C + 2 * ((TM + nthr / 16 - 1) / (nthr / 16)) * nthr : 0; }
// LDS floats of a fused block kernel: scratch + A1, A2 [TM][C+4] + U [TM][3C+4] + row scales / token ids + `params` (the
// forward stages its 9C + hidden bias / LayerNorm vectors; the backward keeps none)
// hidden columns of the MLP a tile handles per pass (a chunk of fc1's outputs = fc2's reduction): 2C where several workgroups
// share a CU's LDS (C <= 96), all 4C at C = 192 (one workgroup per CU either way: two GEMM phases and a GELU pass fewer per tile,
// each a barrier-to-barrier round trip in a kernel that is bound by exactly those)
// (fwd: the forward kernel's choice; the backward's LDS also holds the attention exchange and the parked LayerNorm inputs)
constexpr int block_hidden_chunk(int C, bool fwd = false, int TM = 0) { return C == 192 ? 4 * C : 2 * C; }      // (C = 384: 4C columns do not fit LDS)
// columns of the U tile: q | k | v (3C) or a hidden chunk, whichever is wider
constexpr int block_u_cols(int C, bool fwd = false, int TM = 0) {
  return block_hidden_chunk(C, fwd, TM) > 3 * C ? block_hidden_chunk(C, fwd, TM) : 3 * C;
}
inline size_t block_lds_floats(int TM, int C, int scratch, int params, bool fwd = false) {
  return (size_t)scratch + (size_t)TM * (2 * (C + 4) + block_u_cols(C, fwd, TM) + 4) + 3 * TM + params;
}
// waves per workgroup: 8 where a launch has too few tiles to fill the chip and every tile streams megabytes of weights
// (C = 192: 128 tiles of 16 tokens at the base model's 8^3 stage) -- the x tiles of a phase are then dealt to 8 waves
inline int block_waves(int C) { return C >= 192 ? 8 : 4; }
// (sum16: sum over the 16-lane group that handles a row -- sampler_common.h)
// block_wide.hip: the few-token decomposition of the same two entry points (several launches, GEMMs split over features)
int block_wide_tile_tokens(int C, int hd);
// Storage of what the fused kernels save for the backward / leave for the weight gradients.  MICF_DTYPE_BF16 on the
// tile-per-workgroup kernels (this file's users: C <= 192, and C = 384 with head_dim 32) stores every tensor that is only ever consumed as a matrix-core operand or
// by the attention backward as bf16 -- xn, q, kv, o, xn2, g (+ kvs16, a bf16 copy of a cross block's K/V source) in the forward,
// dq, dkv, dh, dx1 (+ dy16, a bf16 copy of dy) in the backward; the residual stream (x1, y, dx, dxs, dx1_copy), the LayerNorm
// statistics and partial sums stay fp32.  The few-token decomposition (block_wide.hip, C = 384 / head_dim 16) re-reads its own intermediates
// between launches and keeps fp32 everywhere.
inline bool block_saves_bf16(int C, int hd, int dtype) { return dtype == MICF_DTYPE_BF16 && !block_wide_tile_tokens(C, hd); }
int block_fwd_wide(const micf_block_fwd_group* groups, int ngroups, int B, int D, int H, int W, int C, int heads, float eps, float scale,
                   int dtype, hipStream_t s);
int block_bwd_wide(const micf_block_bwd_group* groups, int ngroups, int B, int D, int H, int W, int C, int heads, float scale, int dtype,
                   hipStream_t s);
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
// (wave-uniform base) + (32-bit byte offset): the address costs one VGPR and the access takes the scalar-base form
template <class T>
__device__ __forceinline__ T* at32(T* base, uint32_t byte_off) {
  using B = typename std::conditional<std::is_const<T>::value, const char, char>::type;
  return reinterpret_cast<T*>(reinterpret_cast<B*>(base) + byte_off);
}
// Outputs are written once and read by LATER kernels only: non-temporal stores keep them from evicting the block's weights
// (0.9 MB per XCD at C = 192, re-read by every workgroup) out of the 4 MB L2 while the kernel runs.
__device__ __forceinline__ void st4g(float* p, const float4& v) {
  typedef float f32x4_nt __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_nt*>(p));
}
// four values as bf16 (round-to-nearest-even) in 8 bytes and back: the saved fc1 pre-activation of the bf16 mode
__device__ __forceinline__ uint2 pack4_bf16(const float4& v) { return make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)); }
__device__ __forceinline__ float4 unpack4_bf16(const uint2& u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ void st2g(void* p, const uint2& v) {          // non-temporal, like st4g
  typedef unsigned u32x2_nt __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store(u32x2_nt{v.x, v.y}, reinterpret_cast<u32x2_nt*>(p));
}
// store / load 4 consecutive elements of the saved pre-activation at element offset `e` (multiple of 4)
template <bool BF16> __device__ __forceinline__ void st_h4(void* h, int64_t e, const float4& v) {
  if constexpr (BF16) st2g(static_cast<uint16_t*>(h) + e, pack4_bf16(v));
  else st4g(static_cast<float*>(h) + e, v);
}
// ... the same with a (wave-uniform base, 32-bit element offset) address: one offset register, scalar-base store
template <bool BF16> __device__ __forceinline__ void st_h4_32(void* base, uint32_t e, const float4& v) {
  if constexpr (BF16) st2g(at32(static_cast<char*>(base), e * 2u), pack4_bf16(v));
  else st4g(at32(static_cast<float*>(base), e * 4u), v);
}
template <bool BF16> __device__ __forceinline__ float4 ld_h4(const void* h, int64_t e) {
  if constexpr (BF16) return unpack4_bf16(*reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(h) + e));
  else return ld4g(static_cast<const float*>(h) + e);
}

}  // namespace micf
