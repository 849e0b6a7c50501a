// block_bwd.hip -- ONE launch for the data-gradient chain of the window-local part of a (Cross)TransformerBlock3D, both
// modalities (the adjoint of block_fwd.hip; MS.py:277-524 through autograd in the reference):
//
//   dh  = (s2 dy W2) * GELU'(h)            -> HBM (fc1 weight gradient)          dy itself is fc2's output gradient
//         (h = the saved fc1 pre-activation, or -- RECOMP, micf_block_recomputes_h -- rebuilt here as xn2 W1^T + b1 by the forward's
//          own GEMM phase: the forward then never writes h, 8 of its 36 bytes per element, and this kernel reads 2 instead of 8)
//   dx1 = dy + LN2'(dh W1)                 -> HBM (proj weight gradient; the cross block's LN1 backward adds it)
//   do  = s1 dx1 Wp ;  (dq, dk, dv) = attention'(q, k, v, do)   -> HBM dq, dkv (q / kv weight gradients)
//   self : dx  = dx1 + LN1'(dq Wq + dkv Wkv)                 (all "dY W" products read the TRANSPOSED weights W^T [K, N], which
//                                                            micf_weight_prep_grouped refreshes once per step: contiguous rows)
//   cross: dxq = dq Wq  (pre-LayerNorm; the offset-conv path adds its part before LN1')   and   dxs = dkv Wkv
//
// plus per-tile partial sums of the LayerNorm gain / bias gradients ([tiles][2C], summed by micf_layernorm_bwd_finish).
// Weight gradients stay deferred (linear_grouped.hip): this kernel only leaves their operands in HBM.  Same tiling, LDS
// budget (A1, A2 [TM][C+4]; U [TM][3C+4]) and weight streaming as the forward; the attention backward runs in place on the
// q|k|v tile with the 8x8 P / dS rows exchanged through the (idle) ring area.
#include "block_fused.h"
#include "attn_fp8.h"

namespace micf {

struct BlkBwdArgs {
  micf_block_bwd_group g[2];
  TileGeo geo;
  int G, tiles, C, heads, hidden;
  float scale;
  int attn_mfma;        // bf16 mode: the attention adjoint on v_mfma_f32_16x16x16_bf16 (attn_fp8.h::attn16_bwd_bf16); 0 = the VALU form
};

// Rows of a [TM][4 * X4] tile as the element-wise passes distribute them: pass p, wave w, lane group rg -> row p * RPP + 4 w + rg;
// lane l16 covers the float4 columns l16, l16 + 16, ...  All of a tile's global inputs are REQUESTED at the top of the kernel into
// such register sets (one exposed HBM round trip per tile instead of one per phase: the barriers order LDS only, so the loads stay
// in flight across them) and committed to LDS where the phase that needs them starts.
template <int TM, int NW, int X4>
struct RowRegs {
  static constexpr int RPP = 4 * NW, NPASS = (TM + RPP - 1) / RPP, K = (X4 + 15) / 16;
  float4 v[NPASS][K];
  template <class Addr>
  __device__ __forceinline__ void load(const int* tok, int tk0, const Addr addr) {   // addr(row offset, float4 column) -> const float*
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      const int tk = row < TM ? tok[row] : -1;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int c4 = l16 + 16 * k;
        const bool ok = tk >= 0 && c4 < X4;       // (branch-free: out-of-range lanes read a valid address and drop the value)
        const float4 t = ld4g(addr(tk >= 0 ? (uint32_t)(tk - tk0) : 0u, (uint32_t)(c4 < X4 ? c4 : X4 - 1)));
        v[pass][k] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __device__ __forceinline__ void commit(float* dst, int stride) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      if (row >= TM) continue;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int c4 = l16 + 16 * k;
        if (c4 < X4) *reinterpret_cast<float4*>(dst + row * stride + 4 * c4) = v[pass][k];
      }
    }
  }
  // a bf16 copy of the rows (row length 4 * X4) at `base`, advanced to the tile's first token
  __device__ __forceinline__ void store_b16(const int* tok, int tk0, void* base) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      const int tk = row < TM ? tok[row] : -1;
      if (tk < 0) continue;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int c4 = l16 + 16 * k;
        if (c4 < X4) st_h4_32<true>(base, (uint32_t)(tk - tk0) * (4 * X4) + 4 * c4, v[pass][k]);
      }
    }
  }
};
// Rows of the saved fc1 pre-activation: fp32, or bf16 at half the registers (bf16 mode).  base: the tensor advanced to the tile's
// first token and the chunk's first column; ld: its row length in elements.
template <int TM, int NW, int X4, bool BF16>
struct HRegs {
  static constexpr int RPP = 4 * NW, NPASS = (TM + RPP - 1) / RPP, K = (X4 + 15) / 16, ES = BF16 ? 2 : 4;
  using V = typename std::conditional<BF16, uint2, float4>::type;
  V v[NPASS][K];
  __device__ __forceinline__ void load(const int* tok, int tk0, const char* base, uint32_t ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      const int tk = row < TM ? tok[row] : -1;
      const uint32_t rel = tk >= 0 ? (uint32_t)(tk - tk0) : 0u;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int c4 = l16 + 16 * k;
        const bool ok = tk >= 0 && c4 < X4;
        const V t = *reinterpret_cast<const V*>(base + (rel * ld + 4u * (uint32_t)(c4 < X4 ? c4 : X4 - 1)) * (uint32_t)ES);
        v[pass][k] = ok ? t : V{};
      }
    }
  }
  __device__ __forceinline__ void commit(float* dst, int stride) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      if (row >= TM) continue;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int c4 = l16 + 16 * k;
        if (c4 >= X4) continue;
        float4 f;
        if constexpr (BF16) f = unpack4_bf16(v[pass][k]); else f = v[pass][k];
        *reinterpret_cast<float4*>(dst + row * stride + 4 * c4) = f;
      }
    }
  }
};
// ... plus the LayerNorm statistics of the rows and the gain vector: what a LayerNorm backward over the tile reads from HBM
template <int TM, int NW, int C>
struct LnRegs {
  RowRegs<TM, NW, C / 4> x;
  float mu[RowRegs<TM, NW, C / 4>::NPASS], rs[RowRegs<TM, NW, C / 4>::NPASS];
  float4 gm[(C + 63) / 64];
  // xsrc / mean / rstd: already advanced to the tile's first token tk0
  __device__ __forceinline__ void load(const int* tok, int tk0, const float* __restrict__ xsrc, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const float* __restrict__ gamma) {
    constexpr int RPP = 4 * NW, NPASS = RowRegs<TM, NW, C / 4>::NPASS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, rg = lane >> 4;
    x.load(tok, tk0, [&](uint32_t rel, uint32_t c4) { return at32(xsrc, (rel * C + 4 * c4) * 4u); });
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      const int tk = row < TM ? tok[row] : -1;
      const uint32_t rel = tk >= 0 ? (uint32_t)(tk - tk0) : 0u;
      const float m = *at32(mean, rel * 4u), r = *at32(rstd, rel * 4u);
      mu[pass] = tk >= 0 ? m : 0.f; rs[pass] = tk >= 0 ? r : 0.f;
    }
#pragma unroll
    for (int k = 0; k < (C + 63) / 64; ++k) {
      const int c4 = l16 + 16 * k;
      const float4 t = ld4g(gamma + 4 * (c4 < C / 4 ? c4 : C / 4 - 1));
      gm[k] = c4 < C / 4 ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // Set the registers aside in LDS (`stash`: TM * C + 4 * 64 * NW floats, lane-private slots: no barrier needed) across a
  // phase that needs the register file for itself, and take them back afterwards.
  __device__ __forceinline__ void park(float* stash) const {
    constexpr int NPASS = RowRegs<TM, NW, C / 4>::NPASS;
    x.commit(stash, C);
    float* mine = stash + TM * C + threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) { mine[(2 * pass) * 64 * NW] = mu[pass]; mine[(2 * pass + 1) * 64 * NW] = rs[pass]; }
  }
  __device__ __forceinline__ void unpark(const float* stash, const float* __restrict__ gamma) {
    constexpr int RPP = 4 * NW, NPASS = RowRegs<TM, NW, C / 4>::NPASS, K = RowRegs<TM, NW, C / 4>::K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, rg = lane >> 4;
    const float* mine = stash + TM * C + threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      mu[pass] = mine[(2 * pass) * 64 * NW]; rs[pass] = mine[(2 * pass + 1) * 64 * NW];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int c4 = l16 + 16 * k;
        x.v[pass][k] = (row < TM && c4 < C / 4) ? *reinterpret_cast<const float4*>(stash + row * C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < (C + 63) / 64; ++k) {
      const int c4 = l16 + 16 * k;
      const float4 t = ld4g(gamma + 4 * (c4 < C / 4 ? c4 : C / 4 - 1));
      gm[k] = c4 < C / 4 ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
};

// LayerNorm backward of the rows held in LDS tile `D` (gradient w.r.t. the normalised output, pre-gain) against the preloaded
// LayerNorm input rows `in`: out = addt + rs * (g d - mean(g d) - xh mean(g d xh)).  addt / out are LDS tile A (in place) and
// HBM `hout` / `hout2` (both advanced to the tile's first token tk0).  The per-tile column sums of d * xh and d go through `scratch` (LDS, >= 16 * 2C floats) to part[2C].
template <int TJ, int VPL, int NW, int C, bool H16 = false>
__device__ __forceinline__ void ln_bwd_tile(const float* D, float* A, int S, const LnRegs<16 * TJ, NW, C>& in, const int* tok,
                                            int tk0, void* __restrict__ hout, float* __restrict__ hout2, float* scratch,
                                            float* __restrict__ part) {
  constexpr int TM = 16 * TJ, NTHR = 64 * NW, RPP = 4 * NW, NPASS = (TM + RPP - 1) / RPP, NPR = RPP < TM ? RPP : TM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, rg = lane >> 4;
  constexpr int C4 = C >> 2;
  const float invC = 1.0f / (float)C;
  float4 ag[VPL], ab[VPL], gm[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    ag[k] = make_float4(0.f, 0.f, 0.f, 0.f); ab[k] = ag[k];
    gm[k] = in.gm[k];
  }
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const int row = pass * RPP + wave * 4 + rg;
    if (row >= TM) continue;
    const int tk = tok[row];
    const float mu = in.mu[pass], rs = in.rs[pass];
    float4 xh[VPL], d[VPL];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int c4 = l16 + 16 * k;
      xh[k] = make_float4(0.f, 0.f, 0.f, 0.f); d[k] = xh[k];
      if (c4 < C4 && tk >= 0) {
        const float4 v = in.x.v[pass][k];
        d[k] = *reinterpret_cast<const float4*>(D + row * S + 4 * c4);
        xh[k] = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
        const float g0 = gm[k].x * d[k].x, g1 = gm[k].y * d[k].y, g2 = gm[k].z * d[k].z, g3 = gm[k].w * d[k].w;
        sa += (g0 + g1) + (g2 + g3);
        sb += (g0 * xh[k].x + g1 * xh[k].y) + (g2 * xh[k].z + g3 * xh[k].w);
        ag[k].x += d[k].x * xh[k].x; ag[k].y += d[k].y * xh[k].y; ag[k].z += d[k].z * xh[k].z; ag[k].w += d[k].w * xh[k].w;
        ab[k].x += d[k].x; ab[k].y += d[k].y; ab[k].z += d[k].z; ab[k].w += d[k].w;
      }
    }
  const float Am = sum16(sa) * invC, Bm = sum16(sb) * invC;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int c4 = l16 + 16 * k;
      if (c4 < C4) {
        float* ap = A + row * S + 4 * c4;
        float4 o = *reinterpret_cast<const float4*>(ap);
        o.x += rs * (gm[k].x * d[k].x - Am - xh[k].x * Bm); o.y += rs * (gm[k].y * d[k].y - Am - xh[k].y * Bm);
        o.z += rs * (gm[k].z * d[k].z - Am - xh[k].z * Bm); o.w += rs * (gm[k].w * d[k].w - Am - xh[k].w * Bm);
        if (tk < 0) o = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(ap) = o;
        if (tk >= 0) {
          const uint32_t eoff = (uint32_t)(tk - tk0) * C + 4 * c4;
          st_h4_32<H16>(hout, eoff, o);                     // (H16: the weight-gradient operand copy, bf16)
          if (hout2) st4g(at32(hout2, eoff * 4u), o);
        }
      }
    }
  }
  // column sums over the tile: 16 (wave, lane group) partial rows -> scratch -> part
  float* mine = scratch + (wave * 4 + rg) * 2 * C;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int c4 = l16 + 16 * k;
    if (c4 < C4 && wave * 4 + rg < NPR) {
      *reinterpret_cast<float4*>(mine + 4 * c4) = ag[k];
      *reinterpret_cast<float4*>(mine + C + 4 * c4) = ab[k];
    }
  }
  lds_barrier();
  if (part) {
    for (int c = tid; c < 2 * C; c += NTHR) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < NPR; ++r) s += scratch[r * 2 * C + c];
      part[c] = s;
    }
  }
  lds_barrier();
}

// (three resident workgroups per CU at C = 48 -- four spilled 7 registers and measured slower; two at C = 96 / 192 with 4-8 waves)
template <int C, int HD, int TJ, int NW, bool BF16, bool RECOMP>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : (C <= 48 ? 3 : 2)) block_bwd_kernel(const BlkBwdArgs a) {
  constexpr int TM = 16 * TJ, VPL = (C + 63) / 64, NSL = C / 16, NTHR = 64 * NW, RPP = 4 * NW, NPASS = (TM + RPP - 1) / RPP;
  constexpr bool PARK = block_bwd_park_floats(TM, C, NTHR) != 0;
  constexpr bool LATE = C >= 384;                       // (see the MLP backward: inputs requested in two waves)
  // C = 384: a K = 384 product as two k chunks of 12 slabs -- a wave's weight fragments of a unit are 24 registers instead of 48 (two
  // units in flight) and the activation fragments 24 instead of 48
  constexpr int NSLK = C >= 384 ? NSL / 2 : NSL, NKK = C >= 384 ? 2 : 1;
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  constexpr int C4 = C >> 2, S = C + 4, SU = block_u_cols(C, false, TM) + 4, Hd = 4 * C;
  float* ring = lds;
  float* A1 = ring + block_bwd_scratch_floats(TM, C / HD, NTHR);
  float* A2 = A1 + TM * S;
  float* U = A2 + TM * S;
  float* sc1 = U + TM * SU;
  float* sc2 = sc1 + TM;
  int* tok = reinterpret_cast<int*>(sc2 + TM);
  float* stash = sc2 + 2 * TM;                          // (PARK only) LayerNorm-1 inputs across the attention backward
  float* pb1 = stash + block_bwd_park_floats(TM, C, NTHR);   // (RECOMP only) fc1 bias [4C]
  float* XN2 = block_hidden_chunk(C, false, TM) == 4 * C ? A2 : U + 2 * C;      // (RECOMP only) where the xn2 rows wait, and their row stride
  constexpr int SXN2 = block_hidden_chunk(C, false, TM) == 4 * C ? S : SU;

  int grp, tile;
  if (a.G == 2) { const int xcd = blockIdx.x & 7; grp = xcd >> 2; tile = (int)(blockIdx.x >> 3) * 4 + (xcd & 3); }
  else { grp = 0; tile = blockIdx.x; }
  if (tile >= a.tiles) return;
  const micf_block_bwd_group& g = a.g[grp];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, rg = lane >> 4;
  const int64_t T = a.geo.T;
  using WT = typename std::conditional<BF16, uint16_t, float>::type;      // bf16 mode streams bf16 shadow weights
  const WT* wqt = static_cast<const WT*>(g.wqt), *wkvt = static_cast<const WT*>(g.wkvt), *wpt = static_cast<const WT*>(g.wpt),
           *w1t = static_cast<const WT*>(g.w1t), *w2t = static_cast<const WT*>(g.w2t), *w1f = static_cast<const WT*>(g.w1);

  if (tid < TM) {
    const int win = tile * (TM / 8) + (tid >> 3);
    int tk = -1;
    float v1 = 1.f, v2 = 1.f;
    if (win < a.geo.nwin) {
      tk = a.geo.token(win, tid & 7);
      const int b = (int)a.geo.f_rps.div((uint32_t)tk);
      if (g.s1) v1 = g.s1[b];
      if (g.s2) v2 = g.s2[b];
    }
    tok[tid] = tk; sc1[tid] = v1; sc2[tid] = v2;
  }
  lds_barrier();

  // Row addresses are (scalar base advanced to the tile's first token) + (32-bit byte offset): one register per address and
  // the `saddr` form of the global loads / stores.  tok[0] is the smallest token of the tile (windows are raster ordered).
  const int tk0 = __builtin_amdgcn_readfirstlane(tok[0]);
  const float* dy0 = g.dy + (int64_t)tk0 * C;
  constexpr int ES = BF16 ? 2 : 4;                      // element size of the tensors the bf16 mode stores as bf16
  const char* h0 = static_cast<const char*>(g.h) + (int64_t)tk0 * Hd * ES;
  const char* q0 = reinterpret_cast<const char*>(g.q) + (int64_t)tk0 * C * ES;
  const char* kv0 = reinterpret_cast<const char*>(g.kv) + (int64_t)tk0 * 2 * C * ES;
  char* dh0 = reinterpret_cast<char*>(g.dh) + (int64_t)tk0 * Hd * ES;
  char* dq0 = reinterpret_cast<char*>(g.dq) + (int64_t)tk0 * C * ES;
  char* dkv0 = reinterpret_cast<char*>(g.dkv) + (int64_t)tk0 * 2 * C * ES;
  float* dx0 = g.dx + (int64_t)tk0 * C;

  // ---- (pre) the LayerNorm backward that produces this block's output gradient runs as a prologue: dy = add + LN'(pre_d) -> A1 and
  // the bf16 copy dy16.  Its inputs are requested FIRST and consumed while the tile's other inputs are still in flight (loads return
  // in order); the LayerNorm-1 inputs of the block itself -- needed last -- are requested after it, so the registers the prologue
  // holds are never live together with them.
  const bool pre = BF16 && g.pre_d != nullptr;          // (workgroup-uniform)
  RowRegs<TM, NW, C4> r_add, r_d;
  LnRegs<TM, NW, C> r_lc;
  if constexpr (BF16) {
    if (pre) {
      r_add.load(tok, tk0, [&](uint32_t rel, uint32_t c4) { return at32(dy0, (rel * C + 4 * c4) * 4u); });
      const float* pd0 = g.pre_d + (int64_t)tk0 * C;
      r_d.load(tok, tk0, [&](uint32_t rel, uint32_t c4) { return at32(pd0, (rel * C + 4 * c4) * 4u); });
      r_lc.load(tok, tk0, g.pre_x + (int64_t)tk0 * C, g.pre_mean + tk0, g.pre_rstd + tk0, g.pre_g);
    }
  }

  // ---- request every global input of the tile (see RowRegs)
  constexpr int HC = block_hidden_chunk(C, false, TM);
  RowRegs<TM, NW, C4> r_dy;
  HRegs<TM, NW, HC / 4, BF16> r_h[RECOMP ? 1 : Hd / HC];
  HRegs<TM, NW, C4, BF16> r_xn2;
  float4 r_b1[(Hd / 4 + NTHR - 1) / NTHR];
  HRegs<TM, NW, C4, BF16> r_q;
  HRegs<TM, NW, 2 * C4, BF16> r_kv;
  LnRegs<TM, NW, C> r_ln2, r_ln1;
  if (!pre) r_dy.load(tok, tk0, [&](uint32_t rel, uint32_t c4) { return at32(dy0, (rel * C + 4 * c4) * 4u); });
  if constexpr (RECOMP) {
    // (requested right behind dy: loads return in order, and both are committed to LDS first)
#pragma unroll
    for (int k = 0; k < (Hd / 4 + NTHR - 1) / NTHR; ++k) {
      const int e4 = tid + k * NTHR;
      r_b1[k] = ld4g(g.b1 + 4 * (e4 < Hd / 4 ? e4 : Hd / 4 - 1));
    }
    r_xn2.load(tok, tk0, reinterpret_cast<const char*>(g.xn2) + (int64_t)tk0 * C * ES, (uint32_t)C);
  } else {
#pragma unroll
    for (int ch = 0; ch < (LATE ? 1 : Hd / HC); ++ch)
      r_h[ch].load(tok, tk0, h0 + ch * HC * (BF16 ? 2 : 4), (uint32_t)Hd);
  }
  r_ln2.load(tok, tk0, g.x1 + (int64_t)tk0 * C, g.stats + 2 * T + tk0, g.stats + 3 * T + tk0, g.ln2_g);
  if constexpr (!LATE) {
    if (!g.dxs && !pre) r_ln1.load(tok, tk0, g.x + (int64_t)tk0 * C, g.stats + tk0, g.stats + T + tk0, g.ln1_g);
    r_q.load(tok, tk0, q0, (uint32_t)C);
    r_kv.load(tok, tk0, kv0, (uint32_t)(2 * C));
  }
  // (block_fused.h warm_weights: the XCD's weight set, in the order the phases below read it)
  uint32_t warm = 0;
  if constexpr (C >= 96) warm = warm_weights<C, NTHR, 4, 4, 1, 1, 2>(blockIdx.x, tid, w2t, w1t, wpt, wqt, wkvt);

  // ---- dy rows -> A1 (+ the bf16 copy fc2's weight gradient reads)
  if (!pre) {
    r_dy.commit(A1, S);
    if (BF16 && g.dy16) r_dy.store_b16(tok, tk0, reinterpret_cast<char*>(g.dy16) + (int64_t)tk0 * C * 2);
  } else if constexpr (BF16) {
    r_add.commit(A1, S);
    r_d.commit(A2, S);
    lds_barrier();
    ln_bwd_tile<TJ, VPL, NW, C, true>(A2, A1, S, r_lc, tok, tk0, reinterpret_cast<char*>(g.dy16) + (int64_t)tk0 * C * 2, nullptr, U,
                                      g.pre_part + (int64_t)tile * 2 * C);
    if constexpr (!LATE) r_ln1.load(tok, tk0, g.x + (int64_t)tk0 * C, g.stats + tk0, g.stats + T + tk0, g.ln1_g);      // (a self block: see the entry point)
  }

  // ---- MLP backward in hidden chunks: h chunk -> U;  U <- s2 (dy W2) GELU'(h) = dh (saved);  A2 (+)= dh W1
#pragma unroll
  for (int ch = 0; ch < Hd / HC; ++ch) {
    const int c0 = ch * HC;
    constexpr int hc = HC;
    constexpr int X4 = hc >> 2;
    if constexpr (RECOMP) {
      // h chunk = xn2 W1[c0 .. c0 + HC)^T + b1 -> U[:, 0 .. HC): the forward's phase (same units, same k order); the xn2 rows wait in
      // the third (idle) column block of U -- or, with a single chunk, in A2 (free until the chunk's dh W1 product overwrites it)
      if (ch == 0) {
        r_xn2.commit(XN2, SXN2);
#pragma unroll
        for (int k = 0; k < (Hd / 4 + NTHR - 1) / NTHR; ++k) {
          const int e4 = tid + k * NTHR;
          if (e4 < Hd / 4) *reinterpret_cast<float4*>(pb1 + 4 * e4) = r_b1[k];
        }
        lds_barrier();
      }
      gemm_phase<TJ, NSLK, NKK, C, NW, BF16>(w1f + (int64_t)c0 * C, hc, XN2, nullptr, 0, nullptr, SXN2, U, SU, EpiBias{pb1 + c0});
    } else {
      r_h[ch].commit(U, SU);
      if constexpr (LATE) {       // (the next chunk's rows are requested once this chunk's registers are free)
        if (ch + 1 < Hd / HC) r_h[ch + 1].load(tok, tk0, h0 + (ch + 1) * HC * (BF16 ? 2 : 4), (uint32_t)Hd);
      }
      lds_barrier();
    }
    gemm_phase<TJ, NSLK, NKK, C, NW, BF16>(w2t + (int64_t)c0 * C, hc, A1, nullptr, 0, nullptr, S, U, SU, EpiGeluGrad<BF16>{sc2});
    if (ch == 0) asm volatile("" :: "v"(warm));
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      if (row >= TM) continue;
      const int tk = tok[row];
      if (tk < 0) continue;
      const uint32_t rel = (uint32_t)(tk - tk0);
      for (int c4 = l16; c4 < X4; c4 += 16)
        st_h4_32<BF16>(dh0, rel * Hd + c0 + 4 * c4, *reinterpret_cast<const float4*>(U + row * SU + 4 * c4));
    }
    if (ch == 0) gemm_phase<TJ, NSLK, NKK * HC / C, Hd, NW, BF16>(w1t + 16 * c0, C, U, nullptr, 0, nullptr, SU, A2, S, EpiStore{});
    else gemm_phase<TJ, NSLK, NKK * HC / C, Hd, NW, BF16>(w1t + 16 * c0, C, U, nullptr, 0, nullptr, SU, A2, S, EpiAcc{});
  }

  if constexpr (LATE) {
    // C = 384: a lane's share of ALL the tile's inputs is ~180 registers; what the attention backward and LayerNorm 1 need is
    // requested here instead of at the top and arrives under LayerNorm 2's backward and the proj product
    if (!g.dxs) r_ln1.load(tok, tk0, g.x + (int64_t)tk0 * C, g.stats + tk0, g.stats + T + tk0, g.ln1_g);
    r_q.load(tok, tk0, q0, (uint32_t)C);
    r_kv.load(tok, tk0, kv0, (uint32_t)(2 * C));
  }
  // ---- dx1 = dy + LN2'(A2) -> A1 + HBM; LN2 gain / bias partials
  ln_bwd_tile<TJ, VPL, NW, C, BF16>(A2, A1, S, r_ln2, tok, tk0, reinterpret_cast<char*>(g.dx1) + (int64_t)tk0 * C * ES,
                                    g.dx1_copy ? g.dx1_copy + (int64_t)tk0 * C : nullptr, U,
                                    g.ln2_part ? g.ln2_part + (int64_t)tile * 2 * C : nullptr);

  // ---- do = s1 dx1 Wp -> A2;  q | k | v rows -> U
  gemm_phase<TJ, NSLK, NKK, C, NW, BF16>(wpt, C, A1, nullptr, 0, nullptr, S, A2, S, EpiStoreScale{sc1});
  r_q.commit(U, SU);
  r_kv.commit(U + C, SU);
  if (PARK && !g.dxs) r_ln1.park(stash);
  lds_barrier();

  // ---- attention backward in place on U.
  // bf16 mode (round 5): on the matrix cores, unit = (16-token group = 2 windows, head) dealt to the waves -- seven products per 16
  // channels of the head, the score quads feed the next product directly, no exchange through LDS and no barrier between units: a
  // unit reads and overwrites only its own (rows, head columns) of U (LDS operations of one wave retire in order).
  if (BF16 && a.attn_mfma) {
    constexpr int heads = C / HD;
    for (int unit = wave; unit < TJ * heads; unit += NW) {
      const int gq = unit / heads, hh = unit - gq * heads, hoff = hh * HD;
      float* base = U + gq * 16 * SU + hoff;
      float4 dq[HD / 16], dk[HD / 16], dv[HD / 16];
      attn16_bwd_bf16<HD>(base, base + C, base + 2 * C, SU, A2 + gq * 16 * S + hoff, S, a.scale, dq, dk, dv);
      float* out = base + l16 * SU + 4 * rg;
#pragma unroll
      for (int cb = 0; cb < HD / 16; ++cb) {
        *reinterpret_cast<float4*>(out + 16 * cb) = dq[cb];
        *reinterpret_cast<float4*>(out + C + 16 * cb) = dk[cb];
        *reinterpret_cast<float4*>(out + 2 * C + 16 * cb) = dv[cb];
      }
    }
    lds_barrier();
  } else
  // fp32 (parity) mode: thread = (window, row i, head, half of the head's channels); batches of whole
  // windows.  Where a tile has fewer (row, head) pairs than half the workgroup (C = 48: 96 of 256 threads), 2 or 4 adjacent
  // lanes share a pair: each owns HD / 2 or HD / 4 channels and the partial dot products meet in cross-lane adds.
  {
    // (C = 96 at 32 tokens: 192 pairs on 256 threads would run whole 16-channel head rows per lane -- 80 registers of dq / dk / dv /
    //  q / do rows on top of the parked LayerNorm-1 inputs: 24 spilled.  Two lanes per pair in two batches of 2 windows instead.)
    constexpr bool kSplit96 = C == 96 && TM == 32 && HD == 16;
    constexpr int heads = C / HD, SP = (TM * heads * 4 <= NTHR && HD >= 16) ? 4 : (TM * heads * 2 <= NTHR || kSplit96) ? 2 : 1,
                  HP = HD / SP, per = 8 * heads * SP, wpb = NTHR / per;
    float* PS = ring;                                   // [(row, head) pair][16]: P row | dS row
    // (whole waves rotated per workgroup like the GEMM units: with fewer pairs than threads the last waves = SIMDs stay idle)
    const int vt = (tid + 64 * (int)((blockIdx.x * 2654435761u) >> 20)) & (NTHR - 1);
    for (int w0 = 0; w0 < TM / 8; w0 += wpb) {
      const int wl = vt / per, rem = vt - wl * per;
      const bool active = wl < wpb && (w0 + wl) < TM / 8;
      const int sub = rem & (SP - 1), pr = rem / SP;    // pr = i * heads + hh
      const int i = pr / heads, hh = pr - i * heads;
      const int row = (w0 + wl) * 8 + i, r0 = row - i, hoff = hh * HD + sub * HP;
      const int pair = vt / SP;                         // = (wl * 8 + i) * heads + hh
      float dq[HP], dk[HP], dv[HP];
      if (active) {
        float qr[HP], dor[HP];
        const float* qp = U + row * SU + hoff;
        const float* dop = A2 + row * S + hoff;
#pragma unroll
        for (int d = 0; d < HP; d += 4) {
          const float4 t = *reinterpret_cast<const float4*>(qp + d);
          qr[d] = t.x * a.scale; qr[d + 1] = t.y * a.scale; qr[d + 2] = t.z * a.scale; qr[d + 3] = t.w * a.scale;
          const float4 u = *reinterpret_cast<const float4*>(dop + d);
          dor[d] = u.x; dor[d + 1] = u.y; dor[d + 2] = u.z; dor[d + 3] = u.w;
        }
        float p[8], dp[8], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float* kp = U + (r0 + j) * SU + C + hoff;
          const float* vp = U + (r0 + j) * SU + 2 * C + hoff;
          float sacc = 0.f, dacc = 0.f;
#pragma unroll
          for (int d = 0; d < HP; d += 4) {
            const float4 t = *reinterpret_cast<const float4*>(kp + d);
            sacc += qr[d] * t.x; sacc += qr[d + 1] * t.y; sacc += qr[d + 2] * t.z; sacc += qr[d + 3] * t.w;
            const float4 u = *reinterpret_cast<const float4*>(vp + d);
            dacc += dor[d] * u.x; dacc += dor[d + 1] * u.y; dacc += dor[d + 2] * u.z; dacc += dor[d + 3] * u.w;
          }
          if (SP >= 2) { sacc += __shfl_xor(sacc, 1, 64); dacc += __shfl_xor(dacc, 1, 64); }
          if (SP == 4) { sacc += __shfl_xor(sacc, 2, 64); dacc += __shfl_xor(dacc, 2, 64); }
          p[j] = sacc; dp[j] = dacc;
          mx = fmaxf(mx, sacc);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { p[j] = expf(p[j] - mx); den += p[j]; }
        const float inv = 1.0f / den;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { p[j] *= inv; dot += p[j] * dp[j]; }
#pragma unroll
        for (int d = 0; d < HP; ++d) dq[d] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float ds = p[j] * (dp[j] - dot);
          if (j / (8 / SP) == sub) {                      // (the lanes of a pair hold identical rows: each stores its share)
            PS[pair * 16 + j] = p[j];
            PS[pair * 16 + 8 + j] = ds;
          }
          const float* kp = U + (r0 + j) * SU + C + hoff;
          const float dss = ds * a.scale;
#pragma unroll
          for (int d = 0; d < HP; d += 4) {
            const float4 t = *reinterpret_cast<const float4*>(kp + d);
            dq[d] += dss * t.x; dq[d + 1] += dss * t.y; dq[d + 2] += dss * t.z; dq[d + 3] += dss * t.w;
          }
        }
      }
      lds_barrier();
      if (active) {                                     // now as key / value row j = i: gather column j of P and dS from the mates
#pragma unroll
        for (int d = 0; d < HP; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        const int base = pair - i * heads;              // pair of (window, row 0, head hh)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float pm = PS[(base + m * heads) * 16 + i];
          const float dsm = PS[(base + m * heads) * 16 + 8 + i] * a.scale;
          const float* qp = U + (r0 + m) * SU + hoff;
          const float* dop = A2 + (r0 + m) * S + hoff;
#pragma unroll
          for (int d = 0; d < HP; d += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qp + d);
            dk[d] += dsm * t.x; dk[d + 1] += dsm * t.y; dk[d + 2] += dsm * t.z; dk[d + 3] += dsm * t.w;
            const float4 u = *reinterpret_cast<const float4*>(dop + d);
            dv[d] += pm * u.x; dv[d + 1] += pm * u.y; dv[d + 2] += pm * u.z; dv[d + 3] += pm * u.w;
          }
        }
      }
      lds_barrier();                                  // every read of this batch's q / k / v rows is done: overwrite in place
      if (active) {
#pragma unroll
        for (int d = 0; d < HP; d += 4) {
          *reinterpret_cast<float4*>(U + row * SU + hoff + d) = make_float4(dq[d], dq[d + 1], dq[d + 2], dq[d + 3]);
          *reinterpret_cast<float4*>(U + row * SU + C + hoff + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
          *reinterpret_cast<float4*>(U + row * SU + 2 * C + hoff + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
        }
      }
      lds_barrier();
    }
  }

  // ---- dq | dk | dv rows -> HBM (operands of the q / kv weight gradients)
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    const int row = pass * RPP + wave * 4 + rg;
    if (row >= TM) continue;
    const int tk = tok[row];
    if (tk < 0) continue;
    const uint32_t rel = (uint32_t)(tk - tk0);
    for (int c4 = l16; c4 < 3 * C4; c4 += 16) {
      const float4 v = *reinterpret_cast<const float4*>(U + row * SU + 4 * c4);
      if (c4 < C4) st_h4_32<BF16>(dq0, rel * C + 4 * c4, v);
      else st_h4_32<BF16>(dkv0, rel * 2 * C + 4 * (c4 - C4), v);
    }
  }

  if (!g.dxs) {
    // ---- self: dxn = dq Wq + dkv Wkv -> A2;  dx = dx1 + LN1'(dxn) -> HBM; LN1 partials
    if (PARK) r_ln1.unpark(stash, g.ln1_g);
    gemm_phase<TJ, NSLK, NKK, C, NW, BF16>(wqt, C, U, nullptr, 0, nullptr, SU, A2, S, EpiStore{});
    gemm_phase<TJ, NSLK, 2 * NKK, 2 * C, NW, BF16>(wkvt, C, U + C, nullptr, 0, nullptr, SU, A2, S, EpiAcc{});
    ln_bwd_tile<TJ, VPL, NW, C>(A2, A1, S, r_ln1, tok, tk0, dx0, nullptr, U, g.ln1_part ? g.ln1_part + (int64_t)tile * 2 * C : nullptr);
  } else {
    // ---- cross: the q path's pre-LayerNorm gradient and the sampled K/V source's gradient leave separately
    gemm_phase<TJ, NSLK, NKK, C, NW, BF16>(wqt, C, U, nullptr, 0, nullptr, SU, A2, S, EpiStore{});
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      if (row >= TM) continue;
      const int tk = tok[row];
      if (tk < 0) continue;
      for (int c4 = l16; c4 < C4; c4 += 16)
        st4g(at32(dx0, ((uint32_t)(tk - tk0) * C + 4 * c4) * 4u), *reinterpret_cast<const float4*>(A2 + row * S + 4 * c4));
    }
    lds_barrier();
    gemm_phase<TJ, NSLK, 2 * NKK, 2 * C, NW, BF16>(wkvt, C, U + C, nullptr, 0, nullptr, SU, A2, S, EpiStore{});
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      if (row >= TM) continue;
      const int tk = tok[row];
      if (tk < 0) continue;
      for (int c4 = l16; c4 < C4; c4 += 16)
        st4g(at32(g.dxs + (int64_t)tk0 * C, ((uint32_t)(tk - tk0) * C + 4 * c4) * 4u), *reinterpret_cast<const float4*>(A2 + row * S + 4 * c4));
    }
  }
}

template <int C, int HD, int TJ>
static int launch_bwd(const BlkBwdArgs& a, int dtype, hipStream_t s) {
  constexpr int TM = 16 * TJ, NW = C >= 192 ? 8 : 4;
  const bool recomp = a.g[0].h == nullptr;              // (both groups alike: checked by the entry point)
  const size_t lds = block_lds_floats(TM, C, block_bwd_scratch_floats(TM, C / HD, 64 * NW),
                                      block_bwd_park_floats(TM, C, 64 * NW) + (recomp ? 4 * C : 0)) * sizeof(float);
  if (lds > 160 * 1024) return MICF_EUNSUPPORTED;
  const unsigned grid = a.G == 2 ? (unsigned)((a.tiles + 3) / 4 * 8) : (unsigned)a.tiles;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_bwd_kernel<C, HD, TJ, NW, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_bwd_kernel<C, HD, TJ, NW, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_bwd_kernel<C, HD, TJ, NW, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_bwd_kernel<C, HD, TJ, NW, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (dtype == MICF_DTYPE_BF16) {
    if (recomp) hipLaunchKernelGGL((block_bwd_kernel<C, HD, TJ, NW, true, true>), dim3(grid), dim3(64 * NW), lds, s, a);
    else hipLaunchKernelGGL((block_bwd_kernel<C, HD, TJ, NW, true, false>), dim3(grid), dim3(64 * NW), lds, s, a);
  } else {
    if (recomp) hipLaunchKernelGGL((block_bwd_kernel<C, HD, TJ, NW, false, true>), dim3(grid), dim3(64 * NW), lds, s, a);
    else hipLaunchKernelGGL((block_bwd_kernel<C, HD, TJ, NW, false, false>), dim3(grid), dim3(64 * NW), lds, s, a);
  }
  MICF_RETURN_LAUNCH();
}

}  // namespace micf

#include "block_wave_bwd.h"

using namespace micf;

extern "C" int micf_block_bwd(const micf_block_bwd_group* groups, int ngroups, int B, int D, int H, int W, int C, int heads,
                              int hidden, float scale, int dtype, micf_stream_t stream) {
  if (!groups || ngroups < 1 || ngroups > 2) return MICF_EINVAL;
  const int TM = micf_block_tile_tokens(B, D, H, W, C, heads, hidden, 1);
  if (TM == 0) return MICF_EUNSUPPORTED;
  if (dtype != MICF_DTYPE_F32 && dtype != MICF_DTYPE_BF16) return MICF_EINVAL;
  BlkBwdArgs a;
  for (int i = 0; i < ngroups; ++i) {
    const micf_block_bwd_group& g = groups[i];
    const void* need[] = {g.dy, g.x1, g.stats, g.q, g.kv, g.ln2_g, g.wqt, g.wkvt, g.wpt, g.w1t, g.w2t, g.dx, g.dx1, g.dh, g.dq, g.dkv};
    for (const void* p : need)
      if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) return MICF_EINVAL;
    if (g.h) {                                          // the saved pre-activation ...
      if (reinterpret_cast<uintptr_t>(g.h) & 15) return MICF_EINVAL;
    } else {                                            // ... or what the kernel rebuilds it from (tile kernels only, both groups alike)
      const void* re[] = {g.xn2, g.w1, g.b1};
      for (const void* p : re)
        if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) return MICF_EINVAL;
      if (!micf_block_recomputes_h(C, heads)) return MICF_EINVAL;
    }
    if (i > 0 && (g.h == nullptr) != (groups[0].h == nullptr)) return MICF_EINVAL;
    if (g.pre_d) {                                      // LayerNorm-backward prologue: self block, tile kernels, bf16 storage
      const void* pr[] = {g.pre_d, g.pre_x, g.pre_mean, g.pre_rstd, g.pre_g, g.pre_part, g.dy16};
      for (const void* p : pr)
        if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) return MICF_EINVAL;
      if (g.dxs || dtype != MICF_DTYPE_BF16 || block_wide_tile_tokens(C, C / heads)) return MICF_EINVAL;
    }
    if (!g.dxs && (!g.x || !g.ln1_g)) return MICF_EINVAL;          // self: LayerNorm-1 backward runs in the kernel
    const void* opt[] = {g.x, g.ln1_g, g.dxs, g.ln1_part, g.ln2_part, g.dx1_copy, g.dy16};
    for (const void* p : opt)
      if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return MICF_EINVAL;
    a.g[i] = g;
  }
  if (ngroups == 1) a.g[1] = a.g[0];
  a.geo = make_tile_geo(B, D, H, W);
  a.G = ngroups; a.C = C; a.heads = heads; a.hidden = hidden; a.scale = scale;
  a.tiles = (a.geo.nwin + TM / 8 - 1) / (TM / 8);
  a.attn_mfma = dtype == MICF_DTYPE_BF16 ? 1 : 0;          // (the attention adjoint on the matrix cores in the bf16 modes; VALU in the fp32 parity mode)
  hipStream_t s = (hipStream_t)stream;
  const int hd = C / heads, tj = TM / 16;
  // rows are addressed relative to the tile's first token with 32-bit byte offsets: a tile of TM / 8 consecutive windows spans
  // fewer than 2 H W (TM / 8 + 1) tokens
  if ((int64_t)2 * H * W * (TM / 8 + 1) * hidden * (int64_t)sizeof(float) >= (int64_t)1 << 32) return MICF_EUNSUPPORTED;
  if (block_wide_tile_tokens(C, hd)) return block_bwd_wide(groups, ngroups, B, D, H, W, C, heads, scale, dtype, s);
  // the C = 48 stages in bf16 mode: one wave per 32-token tile, nothing exchanged through LDS (block_wave_bwd.h); the test hook
  // "block_wave" = 0 keeps the tile-per-workgroup kernel (which also serves the fp32 mode and the recomputed h)
  {
    if (options().block_wave != 0 && C == 48 && hd == 16 && tj == 2 && dtype == MICF_DTYPE_BF16 && a.attn_mfma && a.g[0].h && a.g[1].h && (a.g[0].dxs != nullptr) == (a.g[1].dxs != nullptr) &&
        (a.g[0].pre_d != nullptr) == (a.g[1].pre_d != nullptr) && !(a.g[0].dxs && a.g[0].pre_d) &&
        a.geo.T * (int64_t)hidden * 2 < ((int64_t)1 << 31))
    {
      if (options().block_debug & 1) a.attn_mfma |= 2;         // (probe: the wave kernel stores nothing)
      return wave48::launch_bwd_wave48(a, s);
    }
  }
#define MICF_BB(C_, HD_, TJ_) if (C == C_ && hd == HD_ && tj == TJ_) return launch_bwd<C_, HD_, TJ_>(a, dtype, s)
  MICF_BB(48, 16, 2); MICF_BB(96, 16, 2); MICF_BB(192, 16, 1);
  MICF_BB(96, 32, 1); MICF_BB(192, 32, 1); MICF_BB(384, 32, 1);
#undef MICF_BB
  return MICF_EUNSUPPORTED;
}
