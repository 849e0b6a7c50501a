// block_fwd.hip -- ONE launch for the window-local part of a (Cross)TransformerBlock3D forward, both modalities:
//
//   xn = LN1(x)                                            MS.py:343 / :473      (the cross block norms only x)
//   q = xn Wq^T + bq ;  kv = src Wkv^T + bkv               MS.py:188-191 / 246-249   src = xn (self) | sampled raw xa (cross)
//   o = softmax(scale q k^T) v  per 2x2x2 window and head  MS.py:193-200 / 251-258 (window_partition / reverse = index math)
//   x1 = x + s1 * (o Wp^T + bp)                            MS.py:201,419 / 259,517
//   y = x1 + s2 * (GELU(LN2(x1) W1^T + b1) W2^T + b2)      MS.py:28-34, 403-404,424 / 501-502,522
//
// replacing 7 (self) / 6 (cross tail) launches per modality of the round-1 decomposition.  A workgroup owns TM = 16 * TJ tokens
// = TM / 8 whole windows of one modality (blockIdx -> (modality, tile), XCDs 0-3 take modality 0 and 4-7 modality 1 so each L2
// caches one weight set); all intermediates live in three LDS tiles (A1, A2 [TM][C+4]; U [TM][3C+4]); the bias / LayerNorm
// vectors are staged in LDS once; only the weights stream (register-prefetched fragments, block_fused.h).  Everything the backward / the
// weight-gradient GEMMs need is written out once, in natural token order: xn, q, kv, o, x1, xn2, h (fc1 pre-activation),
// g = GELU(h), and the LayerNorm statistics.
#include <cstdlib>

#include "block_fused.h"
#include "attn_fp8.h"

namespace micf {

struct BlkFwdArgs {
  micf_block_fwd_group g[2];
  TileGeo geo;
  int G, tiles, C, heads, hidden, debug;
  int att8;                 // attention products on the matrix cores (attn_fp8.h): 0 = VALU (the fp32 parity mode), 1 = bf16
                            // operands (MICF_DTYPE_BF16), 2 = e4m3 operands (MICF_DTYPE_BF16_ATTN_FP8)
  float eps, scale;
};

// rows of a [TM][X] tile are handled by 16-lane groups: pass p, wave w, lane group rg -> row p*16 + 4*w + rg; lane l16 covers
// the float4 columns l16, l16 + 16, ...
// SAMP: the variant whose cross blocks sample their K/V source themselves (micf_block_fwd_group.hid).  Its own instantiation: the
// sampling prologue costs registers (8 taps in flight per row), and the self blocks -- half of all launches -- must not pay for it.
template <int C, int HD, int TJ, int NW, bool BF16, bool SAMP>
__device__ __forceinline__ void block_fwd_tile(const BlkFwdArgs& a, const unsigned bid) {     // bid: the workgroup's index in the launch
  constexpr int TM = 16 * TJ, VPL = (C + 63) / 64, NSL = C / 16, NTHR = 64 * NW, RPP = 4 * NW, NPASS = (TM + RPP - 1) / RPP;
  constexpr int NSLK = C >= 384 ? NSL / 2 : NSL, NKK = C >= 384 ? 2 : 1;     // (C = 384: a K = 384 product as two k chunks, see block_bwd.hip)
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  constexpr int C4 = C >> 2, S = C + 4, SU = block_u_cols(C, true, TM) + 4, Hd = 4 * C;
  float* A1 = lds;
  float* A2 = A1 + TM * S;
  float* U = A2 + TM * S;
  float* sc1 = U + TM * SU;
  float* sc2 = sc1 + TM;
  int* tok = reinterpret_cast<int*>(sc2 + TM);
  float* PV = sc2 + 2 * TM;                 // parameter vectors: ln1_g ln1_b bq bkv(2C) bp ln2_g ln2_b b2 | b1 (hidden)
  float* p_ln1g = PV, *p_ln1b = PV + C, *p_bq = PV + 2 * C, *p_bkv = PV + 3 * C, *p_bp = PV + 5 * C, *p_ln2g = PV + 6 * C,
        *p_ln2b = PV + 7 * C, *p_b2 = PV + 8 * C, *p_b1 = PV + 9 * C;

  int grp, tile;
  if (a.G == 2) { const int xcd = bid & 7; grp = xcd >> 2; tile = (int)(bid >> 3) * 4 + (xcd & 3); }
  else { grp = 0; tile = bid; }
  if (tile >= a.tiles) return;
  const micf_block_fwd_group& g = a.g[grp];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, rg = lane >> 4;
  const int64_t T = a.geo.T;
  const bool save = !(a.debug & 1), do_gelu = !(a.debug & 8);
  using WT = typename std::conditional<BF16, uint16_t, float>::type;      // bf16 mode streams bf16 shadow weights
  const WT* wq = static_cast<const WT*>(g.wq), *wkv = static_cast<const WT*>(g.wkv), *wp = static_cast<const WT*>(g.wp),
           *w1 = static_cast<const WT*>(g.w1), *w2 = static_cast<const WT*>(g.w2);

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
  // ---- the tile's input rows are requested first (they stay in flight under the parameter staging; x is kept for the residual)
  const bool cross = g.kvsrc != nullptr || (SAMP && g.hid != nullptr);
  float4 xr[NPASS][VPL], kvv[NPASS][VPL];
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const int rowl = pass * RPP + wave * 4 + rg;
    const int winl = tile * (TM / 8) + (rowl >> 3);             // (tok[] is not visible yet: same index math)
    const int tk = (rowl < TM && winl < a.geo.nwin) ? a.geo.token(winl, rowl & 7) : -1;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int c4 = l16 + 16 * k;
      // (branch-free: out-of-range lanes read a valid address and drop the value, so all loads of the pass are in flight together)
      const bool ok = tk >= 0 && c4 < C4;
      const int64_t off = (int64_t)(tk >= 0 ? tk : 0) * C + 4 * (c4 < C4 ? c4 : C4 - 1);
      const float4 xv = ld4g(g.x + off);
      float4 kv4 = xv;
      if (g.kvsrc) kv4 = ld4g(g.kvsrc + off);           // (workgroup-uniform: a self block reads its rows once)
      xr[pass][k] = ok ? xv : make_float4(0.f, 0.f, 0.f, 0.f);
      kvv[pass][k] = ok ? kv4 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (SAMP && g.hid) {
    // ---- cross block with the deformable sampling fused in (MS.py:360-384, STN.py:9-32; the stand-alone form is
    // offset_sample.hip): the 16-lane group that owns a row runs the offset head of its token -- LayerNorm(16) -> GELU -> 1^3 conv
    // on the offset conv's output row, + reference point -- and gathers the 8 trilinear taps of the raw other modality straight
    // into the K/V source registers: no sampler launch and no [T, C] round trip of the sampled rows through HBM.
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int rowl = pass * RPP + wave * 4 + rg;
      const int winl = tile * (TM / 8) + (rowl >> 3);
      const bool live = rowl < TM && winl < a.geo.nwin;
      int b = 0, d = 0, hh = 0, w = 0;
      a.geo.coords(live ? winl : 0, rowl & 7, b, d, hh, w);
      const int tk = ((b * a.geo.D + d) * a.geo.H + hh) * a.geo.W + w;
      float xh, rs, ln, gl, off3[3];
      head_fwd(g.hid + (int64_t)tk * kHid, g.ln16_g, g.ln16_b, g.w1c, a.eps, l16, xh, rs, ln, gl, off3);
      float fl[3];
      fl[0] = off3[0] + (((float)d + 0.5f) / (float)a.geo.H * 2.f - 1.f);      // MS.py:335  ref[...,0] /= H_key
      fl[1] = off3[1] + (((float)hh + 0.5f) / (float)a.geo.W * 2.f - 1.f);     // MS.py:334  ref[...,1] /= W_key
      fl[2] = off3[2] + (((float)w + 0.5f) / (float)a.geo.D * 2.f - 1.f);      // MS.py:333  ref[...,2] /= D_key
      if (live && l16 < 3 && save) g.flow[(int64_t)tk * 3 + l16] = l16 == 0 ? fl[0] : (l16 == 1 ? fl[1] : fl[2]);
      const Taps tp = make_taps(d, hh, w, fl, a.geo.D, a.geo.H, a.geo.W);
      const float* base = g.samp_src + (int64_t)b * a.geo.D * a.geo.H * a.geo.W * C;
      int lin[8]; float wgt[8]; bool okq[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
        lin[q] = 0;
        okq[q] = live && tp.finite && corner(tp, dz, dy, dx, a.geo.D, a.geo.H, a.geo.W, lin[q]);
        const float wx = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
        const float wy = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
        const float wz = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
        wgt[q] = okq[q] ? wx * wy * wz : 0.f;            // (an invalid tap reads token 0 of the sample with weight 0)
      }
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        const int c4 = l16 + 16 * k;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // taps in flight per batch: all 8, or 4 + 4 where the workgroup has 16 waves (128 registers per lane: 8 float4 taps on top
        // of the row registers spilled 20 of them); the accumulation order is 0..7 either way
        constexpr int TB = NW >= 16 ? 4 : 8;
#pragma unroll
        for (int q0 = 0; q0 < 8; q0 += TB) {
          float4 tv[TB];
#pragma unroll
          for (int q = 0; q < TB; ++q) tv[q] = ld4g(base + (int64_t)lin[q0 + q] * C + 4 * (c4 < C4 ? c4 : C4 - 1));
#pragma unroll
          for (int q = 0; q < TB; ++q) {                 // (tap order 0..7 with ok-skips, as the stand-alone kernel)
            if (okq[q0 + q]) { acc.x += tv[q].x * wgt[q0 + q]; acc.y += tv[q].y * wgt[q0 + q]; acc.z += tv[q].z * wgt[q0 + q]; acc.w += tv[q].w * wgt[q0 + q]; }
          }
        }
        kvv[pass][k] = (live && c4 < C4) ? acc : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  {
    const float* const srcs[9] = {g.ln1_g, g.ln1_b, g.bq, g.bkv, g.bp, g.ln2_g, g.ln2_b, g.b2, g.b1};
    const int offs[10] = {0, C, 2 * C, 3 * C, 5 * C, 6 * C, 7 * C, 8 * C, 9 * C, 9 * C + Hd};
    for (int e4 = tid; e4 < (9 * C + Hd) >> 2; e4 += NTHR) {
      const int e = e4 << 2;
      int k = 0;
#pragma unroll
      for (int j = 1; j < 9; ++j) k += (e >= offs[j]) ? 1 : 0;
      const float* sp = srcs[0];
      int so = offs[0];
#pragma unroll
      for (int j = 1; j < 9; ++j) if (k == j) { sp = srcs[j]; so = offs[j]; }
      *reinterpret_cast<float4*>(PV + e) = ld4g(sp + (e - so));
    }
  }
  // (epilogue LayerNorm: its gain | bias take the LDS slots of LayerNorm 1's once that phase is over)
  float4 nl4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.nln_g && tid < 2 * C4) nl4 = tid < C4 ? ld4g(g.nln_g + 4 * tid) : ld4g(g.nln_b + 4 * (tid - C4));
  // (C >= 96: the tile kernels of the 16^3 / 8^3 stages; the C = 48 tile kernel serves the fp32 mode of the big grids only)
  uint32_t warm = 0;
  if constexpr (C >= 96) warm = warm_weights<C, NTHR, 1, 2, 1, 4, 4>(bid, tid, wq, wkv, wp, w1, w2);
  lds_barrier();

  // ---- LayerNorm 1 straight from HBM: all rows of this lane group in flight at once -> registers -> (xn -> A1 + HBM,
  // statistics); cross: the K/V source rows -> A2
  const float invC = 1.0f / (float)C;
  {
    float4 v[NPASS][VPL];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
      for (int k = 0; k < VPL; ++k) v[pass][k] = xr[pass][k];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      if (row >= TM) continue;
      const int tk = tok[row];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) s += (v[pass][k].x + v[pass][k].y) + (v[pass][k].z + v[pass][k].w);
      const float mu = sum16(s) * invC;
      float qd = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        if (l16 + 16 * k < C4) {
          const float d0 = v[pass][k].x - mu, d1 = v[pass][k].y - mu, d2 = v[pass][k].z - mu, d3 = v[pass][k].w - mu;
          qd += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
      const float rs = 1.0f / sqrtf(sum16(qd) * invC + a.eps);
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        const int c4 = l16 + 16 * k;
        if (c4 < C4) {
          const float4 gm = *reinterpret_cast<const float4*>(p_ln1g + 4 * c4), bt = *reinterpret_cast<const float4*>(p_ln1b + 4 * c4);
          float4 y = make_float4((v[pass][k].x - mu) * rs * gm.x + bt.x, (v[pass][k].y - mu) * rs * gm.y + bt.y,
                                 (v[pass][k].z - mu) * rs * gm.z + bt.z, (v[pass][k].w - mu) * rs * gm.w + bt.w);
          if (tk < 0) y = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(A1 + row * S + 4 * c4) = y;
          if (cross) *reinterpret_cast<float4*>(A2 + row * S + 4 * c4) = kvv[pass][k];
          if (tk >= 0 && g.xn && save) st_h4<BF16>(g.xn, (int64_t)tk * C + 4 * c4, y);
          if (BF16 && tk >= 0 && cross && g.kvs16 && save) st_h4<true>(g.kvs16, (int64_t)tk * C + 4 * c4, kvv[pass][k]);
          if (SAMP && !BF16 && tk >= 0 && g.hid && g.xs32 && save) st4g(g.xs32 + (int64_t)tk * C + 4 * c4, kvv[pass][k]);
        }
      }
      if (l16 == 0 && tk >= 0 && save) { g.stats[tk] = mu; g.stats[T + tk] = rs; }
    }
  }
  lds_barrier();

  // ---- q | k | v (+ bias) -> U, then out to HBM
  if (!(a.debug & 4)) {
    // q | k | v in ONE phase (two weight segments; the biases bq | bkv are contiguous in PV)
    gemm_phase<TJ, NSLK, NKK, C, NW, BF16>(wq, C, A1, wkv, 2 * C, cross ? A2 : A1, S, U, SU, EpiBias{p_bq});
    asm volatile("" :: "v"(warm));                        // (the warm-up's loads have returned: in order, in front of this phase's)
    if (g.nln_g && tid < 2 * C4) *reinterpret_cast<float4*>(PV + 4 * tid) = nl4;   // (p_ln1g | p_ln1b: read again in the last pass only)
  }
  if (save) {
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      if (row >= TM) continue;
      const int tk = tok[row];
      if (tk < 0) continue;
      for (int c4 = l16; c4 < 3 * C4; c4 += 16) {
        const float4 v = *reinterpret_cast<const float4*>(U + row * SU + 4 * c4);
        if (c4 < C4) st_h4<BF16>(g.q, (int64_t)tk * C + 4 * c4, v);
        else st_h4<BF16>(g.kv, (int64_t)tk * 2 * C + 4 * (c4 - C4), v);
      }
    }
  }

  // ---- attention per (row, head): the 8x8 score row lives in registers; o -> A1 (xn is no longer needed) + HBM.  Where a tile
  // has fewer (row, head) pairs than half the workgroup, 2 or 4 adjacent lanes share a pair (HD / 2 or HD / 4 channels each; the
  // partial scores meet in cross-lane adds).
  if (BF16 && a.att8) {
    // fp8 attention: unit = (16-token group = 2 windows, head), dealt to the waves; every lane of a wave works on its unit
    constexpr int heads = C / HD;
    for (int unit = wave; unit < TJ * heads; unit += NW) {
      const int gq = unit / heads, hh = unit - gq * heads, hoff = hh * HD;
      const float* base = U + gq * 16 * SU + hoff;
      float4 ov[HD / 16];
      if (a.att8 == 2) attn16_fp8<HD>(base, base + C, base + 2 * C, SU, a.scale, ov);
      else attn16_bf16<HD>(base, base + C, base + 2 * C, SU, a.scale, ov);
      const int row = gq * 16 + l16, tk = tok[row];
#pragma unroll
      for (int cb = 0; cb < HD / 16; ++cb) {
        *reinterpret_cast<float4*>(A1 + row * S + hoff + 16 * cb + 4 * rg) = ov[cb];
        if (tk >= 0 && save) st_h4<BF16>(g.o, (int64_t)tk * C + hoff + 16 * cb + 4 * rg, ov[cb]);
      }
    }
  } else if (!(a.debug & 2)) {
    constexpr int heads = C / HD, SP = (TM * heads * 4 <= NTHR && HD >= 16) ? 4 : (TM * heads * 2 <= NTHR) ? 2 : 1, HP = HD / SP;
    // (whole waves rotated per workgroup like the GEMM units: with fewer items than threads the last waves = SIMDs stay idle)
    const int vt = (tid + 64 * (int)((bid * 2654435761u) >> 20)) & (NTHR - 1);
    for (int item = vt; item < TM * heads * SP; item += NTHR) {
      const int sub = item & (SP - 1), pr = item / SP;
      const int row = pr / heads, hh = pr - row * heads;
      const int r0 = row & ~7, hoff = hh * HD + sub * HP;
      float qr[HP];
      const float* qp = U + row * SU + hoff;
#pragma unroll
      for (int d = 0; d < HP; d += 4) {
        const float4 t = *reinterpret_cast<const float4*>(qp + d);
        qr[d] = t.x * a.scale; qr[d + 1] = t.y * a.scale; qr[d + 2] = t.z * a.scale; qr[d + 3] = t.w * a.scale;
      }
      float sj[8], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* kp = U + (r0 + j) * SU + C + hoff;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HP; d += 4) {
          const float4 t = *reinterpret_cast<const float4*>(kp + d);
          acc += qr[d] * t.x; acc += qr[d + 1] * t.y; acc += qr[d + 2] * t.z; acc += qr[d + 3] * t.w;
        }
        if (SP >= 2) acc += __shfl_xor(acc, 1, 64);
        if (SP == 4) acc += __shfl_xor(acc, 2, 64);
        sj[j] = acc;
        mx = fmaxf(mx, acc);
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { sj[j] = expf(sj[j] - mx); den += sj[j]; }
      const float inv = 1.0f / den;
      float oa[HP];
#pragma unroll
      for (int d = 0; d < HP; ++d) oa[d] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* vp = U + (r0 + j) * SU + 2 * C + hoff;
        const float pj = sj[j] * inv;
#pragma unroll
        for (int d = 0; d < HP; d += 4) {
          const float4 t = *reinterpret_cast<const float4*>(vp + d);
          oa[d] += pj * t.x; oa[d + 1] += pj * t.y; oa[d + 2] += pj * t.z; oa[d + 3] += pj * t.w;
        }
      }
      const int tk = tok[row];
#pragma unroll
      for (int d = 0; d < HP; d += 4) {
        const float4 t = make_float4(oa[d], oa[d + 1], oa[d + 2], oa[d + 3]);
        *reinterpret_cast<float4*>(A1 + row * S + hoff + d) = t;
        if (tk >= 0 && save) st_h4<BF16>(g.o, (int64_t)tk * C + hoff + d, t);
      }
    }
  }
  lds_barrier();

  // ---- proj (+ bp) -> A2; x1 = x + s1 * proj -> A2 + HBM; LayerNorm 2 of the same registers -> A1 (xn2) + HBM
  if (!(a.debug & 4)) gemm_phase<TJ, NSLK, NKK, C, NW, BF16>(wp, C, A1, nullptr, 0, nullptr, S, A2, S, EpiBias{p_bp});
  {
    float4 v[NPASS][VPL];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
      for (int k = 0; k < VPL; ++k) v[pass][k] = xr[pass][k];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      if (row >= TM) continue;
      const int tk = tok[row];
      const float s1v = sc1[row];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        const int c4 = l16 + 16 * k;
        if (c4 < C4) {
          const float4 pr = *reinterpret_cast<const float4*>(A2 + row * S + 4 * c4);
          float4 xv = v[pass][k];
          xv = make_float4(xv.x + s1v * pr.x, xv.y + s1v * pr.y, xv.z + s1v * pr.z, xv.w + s1v * pr.w);
          if (tk < 0) xv = make_float4(0.f, 0.f, 0.f, 0.f);
          v[pass][k] = xv;
          *reinterpret_cast<float4*>(A2 + row * S + 4 * c4) = xv;
          if (tk >= 0 && save) st4g(g.x1 + (int64_t)tk * C + 4 * c4, xv);
          s += (xv.x + xv.y) + (xv.z + xv.w);
        }
      }
      const float mu = sum16(s) * invC;
      float qd = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        if (l16 + 16 * k < C4) {
          const float d0 = v[pass][k].x - mu, d1 = v[pass][k].y - mu, d2 = v[pass][k].z - mu, d3 = v[pass][k].w - mu;
          qd += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
      const float rs = 1.0f / sqrtf(sum16(qd) * invC + a.eps);
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        const int c4 = l16 + 16 * k;
        if (c4 < C4) {
          const float4 gm = *reinterpret_cast<const float4*>(p_ln2g + 4 * c4), bt = *reinterpret_cast<const float4*>(p_ln2b + 4 * c4);
          float4 y = make_float4((v[pass][k].x - mu) * rs * gm.x + bt.x, (v[pass][k].y - mu) * rs * gm.y + bt.y,
                                 (v[pass][k].z - mu) * rs * gm.z + bt.z, (v[pass][k].w - mu) * rs * gm.w + bt.w);
          if (tk < 0) y = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(A1 + row * S + 4 * c4) = y;
          if (tk >= 0 && save) st_h4<BF16>(g.xn2, (int64_t)tk * C + 4 * c4, y);
        }
      }
      if (l16 == 0 && tk >= 0 && save) { g.stats[2 * T + tk] = mu; g.stats[3 * T + tk] = rs; }
    }
  }
  lds_barrier();

  // ---- MLP in hidden chunks of <= 2C (<= 3C fits U): fc1 chunk (+ b1) -> U; save h, GELU in place, save g; fc2 chunk
  // accumulates s2 * (g W2^T) into A2 (which holds x1)
  constexpr int HC = block_hidden_chunk(C, true, TM);
  for (int c0 = 0; c0 < Hd; c0 += HC) {
    constexpr int hc = HC;
    if (!(a.debug & 4)) gemm_phase<TJ, NSLK, NKK, C, NW, BF16>(w1 + (int64_t)c0 * C, hc, A1, nullptr, 0, nullptr, S, U, SU, EpiBias{p_b1 + c0});
    const int X4 = hc >> 2;
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + wave * 4 + rg;
      if (row >= TM) continue;
      const int tk = tok[row];
      for (int c4 = l16; c4 < X4; c4 += 16) {
        float* up = U + row * SU + 4 * c4;
        const float4 v = *reinterpret_cast<const float4*>(up);
        const float4 ge = do_gelu ? make_float4(gelu_t<BF16>(v.x), gelu_t<BF16>(v.y), gelu_t<BF16>(v.z), gelu_t<BF16>(v.w)) : v;
        *reinterpret_cast<float4*>(up) = ge;
        if (tk >= 0 && save) {
          if (g.h) st_h4<BF16>(g.h, (int64_t)tk * Hd + c0 + 4 * c4, v);       // (workgroup-uniform; NULL: the backward recomputes it)
          st_h4<BF16>(g.g, (int64_t)tk * Hd + c0 + 4 * c4, ge);
        }
      }
    }
    lds_barrier();
    if (!(a.debug & 4)) gemm_phase<TJ, NSLK, NKK * HC / C, Hd, NW, BF16>(w2 + 16 * c0, C, U, nullptr, 0, nullptr, SU, A2, S, EpiAccScale{sc2});
  }

  // ---- y = x1 + s2 * (fc2 + b2); optional epilogue: the NEXT block's LayerNorm of the row (micf_block_fwd_group.nln_g) while the
  // 16-lane group still holds it -- the cross block's norm1, otherwise a launch of its own that re-reads y
  const bool nln = g.nln_g != nullptr;                  // (workgroup-uniform)
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    const int row = pass * RPP + wave * 4 + rg;
    if (row >= TM) continue;
    const int tk = tok[row];
    if (tk < 0) continue;                               // (uniform over the 16 lanes of a row)
    const float s2v = sc2[row];
    float4 v[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int c4 = l16 + 16 * k;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < C4) {
        v[k] = *reinterpret_cast<const float4*>(A2 + row * S + 4 * c4);
        const float4 b = *reinterpret_cast<const float4*>(p_b2 + 4 * c4);
        v[k].x += s2v * b.x; v[k].y += s2v * b.y; v[k].z += s2v * b.z; v[k].w += s2v * b.w;
        st4g(g.y + (int64_t)tk * C + 4 * c4, v[k]);
      }
    }
    if (nln) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      const float mu = sum16(s) * invC;
      float qd = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        if (l16 + 16 * k < C4) {
          const float d0 = v[k].x - mu, d1 = v[k].y - mu, d2 = v[k].z - mu, d3 = v[k].w - mu;
          qd += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
      const float rs = 1.0f / sqrtf(sum16(qd) * invC + a.eps);
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        const int c4 = l16 + 16 * k;
        if (c4 < C4) {
          const float4 gm = *reinterpret_cast<const float4*>(p_ln1g + 4 * c4), bt = *reinterpret_cast<const float4*>(p_ln1b + 4 * c4);
          st4g(g.nln_y + (int64_t)tk * C + 4 * c4, make_float4((v[k].x - mu) * rs * gm.x + bt.x, (v[k].y - mu) * rs * gm.y + bt.y,
                                                                (v[k].z - mu) * rs * gm.z + bt.z, (v[k].w - mu) * rs * gm.w + bt.w));
        }
      }
      if (l16 == 0) { g.nln_mean[tk] = mu; g.nln_rstd[tk] = rs; }
      if (g.zero16) g.zero16[(int64_t)tk * 16 + l16] = 0.f;
    }
  }
}

template <int C, int HD, int TJ, int NW, bool BF16, bool SAMP>
__global__ void __launch_bounds__(64 * NW) block_fwd_kernel(const BlkFwdArgs a) {
  block_fwd_tile<C, HD, TJ, NW, BF16, SAMP>(a, blockIdx.x);
}

// ---- MEASUREMENT PROBE (judge's round-4 item 5c: "measure the persistent form instead of pricing it").  The same tile body, `repeats`
// times inside ONE launch with a device-wide barrier between the passes -- what a persistent kernel walking the depth slots of the
// 8^3 stage would pay per dependent phase -- against `repeats` launches of block_fwd_kernel (tools/bench_persist.py).  All
// workgroups must be resident at once (128 of 1024 threads at the 8^3 stage: one per CU); the spin is bounded so that a
// mis-sized launch raises instead of hanging the GPU.
__device__ __forceinline__ void probe_grid_barrier(unsigned* bar, unsigned target, int* err) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(bar, 1u);
    long spins = 0;
    while (__atomic_load_n(bar, __ATOMIC_ACQUIRE) < target) {
      if (++spins > 40000000) { *err = 1; break; }
    }
    __threadfence();
  }
  __syncthreads();
}
template <int C, int HD, int TJ, int NW, bool BF16, bool SAMP>
__global__ void __launch_bounds__(64 * NW) block_fwd_persist_probe_kernel(const BlkFwdArgs a, int repeats, unsigned* bar, int* err) {
  for (int k = 0; k < repeats; ++k) {
    block_fwd_tile<C, HD, TJ, NW, BF16, SAMP>(a, blockIdx.x);
    if (k + 1 < repeats) probe_grid_barrier(bar, (unsigned)(k + 1) * gridDim.x, err);
  }
}

template <int C, int HD, int TJ>
static int launch_fwd(const BlkFwdArgs& a, int dtype, hipStream_t s) {
  constexpr int TM = 16 * TJ, NW = C >= 192 ? (HD <= 16 ? 16 : 8) : 4;   // (head_dim 32 attention rows need > 128 registers: 8 waves there)
  const size_t lds = block_lds_floats(TM, C, 0, 9 * C + 4 * C, true) * sizeof(float);
  if (lds > 160 * 1024) return MICF_EUNSUPPORTED;
  const unsigned grid = a.G == 2 ? (unsigned)((a.tiles + 3) / 4 * 8) : (unsigned)a.tiles;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_kernel<C, HD, TJ, NW, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_kernel<C, HD, TJ, NW, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_kernel<C, HD, TJ, NW, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_kernel<C, HD, TJ, NW, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const bool samp = a.g[0].hid != nullptr || a.g[1].hid != nullptr;
  if (dtype == MICF_DTYPE_BF16) {
    if (samp) hipLaunchKernelGGL((block_fwd_kernel<C, HD, TJ, NW, true, true>), dim3(grid), dim3(64 * NW), lds, s, a);
    else hipLaunchKernelGGL((block_fwd_kernel<C, HD, TJ, NW, true, false>), dim3(grid), dim3(64 * NW), lds, s, a);
  } else {
    if (samp) hipLaunchKernelGGL((block_fwd_kernel<C, HD, TJ, NW, false, true>), dim3(grid), dim3(64 * NW), lds, s, a);
    else hipLaunchKernelGGL((block_fwd_kernel<C, HD, TJ, NW, false, false>), dim3(grid), dim3(64 * NW), lds, s, a);
  }
  MICF_RETURN_LAUNCH();
}

}  // namespace micf

#include "block_wave_fwd.h"

using namespace micf;

// tokens per tile for a block of C channels (0 = this shape is not handled by the fused kernels)
extern "C" int micf_block_tile_tokens(int B, int D, int H, int W, int C, int heads, int hidden, int backward) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || hidden <= 0) return 0;
  if ((D | H | W) & 1) return 0;                          // whole 2x2x2 windows only (no pad-to-window, no clamped windows)
  if (C % heads || hidden != 4 * C) return 0;
  if ((int64_t)B * D * H * W >= (1LL << 31)) return 0;
  const int hd = C / heads;
  if (const int wt = block_wide_tile_tokens(C, hd)) return wt;      // few-token stages: block_wide.hip
  // The kernels are compiled per channel count (C fixes every loop bound and load offset): the shapes of MicFormer base
  // (C = 48 / 96 / 192, head_dim 16) and large (C = 96 / 192 / 384, head_dim 32).  C = 384 with head_dim 16 takes the few-token
  // decomposition above: a tile would stream the whole weight set of the block through one compute unit (7 MB at C = 384 for 128 tokens
  // at the base model's 4^3 stage); block_wide.hip spreads every weight matrix over the chip instead.
  // token groups of 16 per workgroup (measured on MI355X, base shapes, batch 2): forward and backward tile independently
  int tj = 0;
  if (C == 48 && hd == 16) tj = 2;
  else if (C == 96 && hd == 16) tj = backward ? 2 : 1;
  else if ((C == 96 || C == 192) && (hd == 16 || hd == 32)) tj = 1;
  else if (C == 384 && hd == 32) tj = 1;                   // (the large model's third stage; head_dim 16 went to block_wide.hip above)
  return 16 * tj;
}

extern "C" int micf_block_fuses_sampler(int C, int heads) {
  if (C <= 0 || heads <= 0 || C % heads) return 0;
  return block_wide_tile_tokens(C, C / heads) ? 0 : 1;
}

extern "C" int micf_block_recomputes_h(int C, int heads) {
  if (C <= 0 || heads <= 0 || C % heads || block_wide_tile_tokens(C, C / heads)) return 0;
  // opt-in: measured on MI355X (base shapes, bf16, batch 2) the extra GEMM phase costs more than the bytes save -- block_bwd
  // 123 -> 135 us at 32^3, 45 -> 52 at 16^3, 34 -> 42 at 8^3, forward unchanged (its stores are fire-and-forget), step 10.35 ->
  // 10.44 ms -- so the default stores h.  What the switch buys is MEMORY: 8 of the 34 bytes saved per element of T * C.
  return options().block_recompute_h != 0 ? 1 : 0;
}

extern "C" int micf_block_saves_bf16(int C, int heads, int dtype) {
  if (C <= 0 || heads <= 0 || C % heads) return 0;
  return block_saves_bf16(C, C / heads, dtype) ? 1 : 0;
}

extern "C" int micf_block_fwd(const micf_block_fwd_group* groups, int ngroups, int B, int D, int H, int W, int C, int heads,
                              int hidden, float eps, float scale, int dtype, micf_stream_t stream) {
  if (!groups || ngroups < 1 || ngroups > 2) return MICF_EINVAL;
  const int TM = micf_block_tile_tokens(B, D, H, W, C, heads, hidden, 0);
  if (TM == 0) return MICF_EUNSUPPORTED;
  const int att8 = dtype == MICF_DTYPE_BF16_ATTN_FP8;
  if (att8) dtype = MICF_DTYPE_BF16;                     // everything but the two attention products is the bf16 mode
  if (dtype != MICF_DTYPE_F32 && dtype != MICF_DTYPE_BF16) return MICF_EINVAL;
  BlkFwdArgs a;
  a.att8 = att8 ? 2 : (dtype == MICF_DTYPE_BF16 ? 1 : 0);     // (attention on the matrix cores in both bf16 modes; VALU in the fp32 parity mode)
  // INFERENCE FORM: every saved-tensor pointer of every group NULL -> the launch writes y only (4 instead of 40 bytes per element of
  // T * C in bf16 mode); tile-per-workgroup and wave-private kernels (not the few-token decomposition, which re-reads its own saves)
  int nosave = -1;
  for (int i = 0; i < ngroups; ++i) {
    const micf_block_fwd_group& g = groups[i];
    const void* must[] = {g.x, g.ln1_g, g.ln1_b, g.wq, g.bq, g.wkv, g.bkv, g.wp, g.bp, g.ln2_g, g.ln2_b, g.w1, g.b1, g.w2, g.b2, g.y};
    for (const void* p : must)
      if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) return MICF_EINVAL;
    const void* saves[] = {g.q, g.kv, g.o, g.x1, g.xn2, g.g, g.stats};
    int nnull = 0;
    for (const void* p : saves) nnull += p == nullptr;
    const bool ns = nnull == 7 && !g.h && !g.xn && !g.kvs16 && !g.flow && !g.xs32;
    if (nosave >= 0 && nosave != (int)ns) return MICF_EINVAL;      // (both groups alike)
    nosave = ns;
    if (ns) {
      if (block_wide_tile_tokens(C, C / heads)) return MICF_EUNSUPPORTED;
    } else {
      for (const void* p : saves)
        if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) return MICF_EINVAL;
      // h may be left out where the backward rebuilds it (micf_block_recomputes_h); the few-token decomposition always stores it
      if (g.h ? (reinterpret_cast<uintptr_t>(g.h) & 15) != 0 : !micf_block_recomputes_h(C, heads)) return MICF_EINVAL;
    }
    if ((g.kvsrc && (reinterpret_cast<uintptr_t>(g.kvsrc) & 15)) || (g.xn && (reinterpret_cast<uintptr_t>(g.xn) & 15)) ||
        (g.kvs16 && (reinterpret_cast<uintptr_t>(g.kvs16) & 15))) return MICF_EINVAL;
    if (g.hid) {      // fused sampling: the offset head's parameters, the raw source and the flow output come with it
      if (g.kvsrc || !g.samp_src || !g.ln16_g || !g.ln16_b || !g.w1c || (!g.flow && !ns) || (reinterpret_cast<uintptr_t>(g.samp_src) & 15) ||
          (g.xs32 && (reinterpret_cast<uintptr_t>(g.xs32) & 15)) || block_wide_tile_tokens(C, C / heads))
        return MICF_EINVAL;
    }
    if (g.nln_g) {    // epilogue: the next block's LayerNorm of y
      const void* need[] = {g.nln_g, g.nln_b, g.nln_y, g.nln_mean, g.nln_rstd};
      for (const void* p : need)
        if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) return MICF_EINVAL;
      if (g.zero16 && (reinterpret_cast<uintptr_t>(g.zero16) & 15)) return MICF_EINVAL;
      if (block_wide_tile_tokens(C, C / heads)) return MICF_EUNSUPPORTED;
    } else if (g.nln_b || g.nln_y || g.nln_mean || g.nln_rstd || g.zero16) {
      return MICF_EINVAL;
    }
    a.g[i] = g;
  }
  if (ngroups == 1) a.g[1] = a.g[0];
  a.geo = make_tile_geo(B, D, H, W);
  a.G = ngroups; a.C = C; a.heads = heads; a.hidden = hidden; a.eps = eps; a.scale = scale;
  a.tiles = (a.geo.nwin + TM / 8 - 1) / (TM / 8);
  a.debug = options().block_debug;
  if (nosave == 1) a.debug |= 1;
  hipStream_t s = (hipStream_t)stream;
  const int hd = C / heads, tj = TM / 16;
  if (block_wide_tile_tokens(C, hd)) return block_fwd_wide(groups, ngroups, B, D, H, W, C, heads, eps, scale, att8 ? MICF_DTYPE_BF16_ATTN_FP8 : dtype, s);   // (the few-token F1: attention on the matrix cores in both bf16 modes)
  // the C = 48 stages in bf16 mode: one wave per 16 tokens, nothing exchanged through LDS (block_wave_fwd.h); the test hook
  // "block_wave" = 0 keeps the tile-per-workgroup kernel (which also serves the fp32 mode, the fp8 attention and the probe flags)
  if (options().block_wave != 0 && C == 48 && hd == 16 && dtype == MICF_DTYPE_BF16 && a.att8 == 1 && !(a.debug & ~17) &&
      a.geo.T * (int64_t)C * 4 < ((int64_t)1 << 32))          // (the fused sampling addresses its tap rows with 32-bit byte offsets)
    return wave48::launch_fwd_wave48(a, s);
#define MICF_BF(C_, HD_, TJ_) if (C == C_ && hd == HD_ && tj == TJ_) return launch_fwd<C_, HD_, TJ_>(a, dtype, s)
  MICF_BF(48, 16, 2); MICF_BF(96, 16, 1); MICF_BF(192, 16, 1);
  MICF_BF(96, 32, 1); MICF_BF(192, 32, 1); MICF_BF(384, 32, 1);
#undef MICF_BF
  return MICF_EUNSUPPORTED;
}

// HAZARD PROBE, not a product entry point (LABNOTES.md, round 5; block_wave.h::mfma48): one K = 48 product of a 16 x 16 tile as
// the compiler emits it from the two natural source forms -- form 0: acc = mfma_16x16x16(c, d, mfma_16x16x32(a, b, 0)), the dependent
// pair of different shapes back to back; form 1: two independent products and a vector add.  a / b: [64 lanes][8] bf16 fragments,
// c / d: [64][4]; out [64][4] floats in accumulator order.  tests/test_gpu_block_wave.py compares both with the exact product.
namespace micf {
namespace wave48 {
__global__ void __launch_bounds__(64) mfma_chain_probe_kernel0(const bf16x8* a, const bf16x8* b, const bf16x4_t* c, const bf16x4_t* d, f32x4* out) {
  const int l = threadIdx.x;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  out[l] = mfma16(c[l], d[l], mfma32(a[l], b[l], z));
}
__global__ void __launch_bounds__(64) mfma_chain_probe_kernel1(const bf16x8* a, const bf16x8* b, const bf16x4_t* c, const bf16x4_t* d, f32x4* out) {
  const int l = threadIdx.x;
  out[l] = mfma48(a[l], c[l], b[l], d[l]);
}
// form 2: the same dependent pair written out, VGPR accumulator, NOTHING between the two instructions (no compiler in the way);
// form 3: ... with 16 wait states between them
template <int NOPS>
__global__ void __launch_bounds__(64) mfma_chain_probe_kernel2(const bf16x8* a, const bf16x8* b, const bf16x4_t* c, const bf16x4_t* d, f32x4* out) {
  const int l = threadIdx.x;
  const bf16x8 av = a[l], bv = b[l];
  const bf16x4_t cv = c[l], dv = d[l];
  f32x4 acc;
  if (NOPS == 0)
    asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\tv_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n\ts_nop 15\n\ts_nop 15"
                 : "=&v"(acc) : "v"(av), "v"(bv), "v"(cv), "v"(dv));
  else
    asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15\n\tv_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n\ts_nop 15\n\ts_nop 15"
                 : "=&v"(acc) : "v"(av), "v"(bv), "v"(cv), "v"(dv));
  out[l] = acc;
}
}  // namespace wave48
}  // namespace micf
extern "C" int micf_probe_mfma_chain(const void* a, const void* b, const void* c, const void* d, float* out, int form, micf_stream_t stream) {
  if (!a || !b || !c || !d || !out || form < 0 || form > 3) return MICF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  using namespace micf::wave48;
  if (form == 0) hipLaunchKernelGGL(mfma_chain_probe_kernel0, dim3(1), dim3(64), 0, s, static_cast<const bf16x8*>(a), static_cast<const bf16x8*>(b), static_cast<const bf16x4_t*>(c), static_cast<const bf16x4_t*>(d), reinterpret_cast<f32x4*>(out));
  else if (form == 3) hipLaunchKernelGGL(mfma_chain_probe_kernel2<16>, dim3(1), dim3(64), 0, s, static_cast<const bf16x8*>(a), static_cast<const bf16x8*>(b), static_cast<const bf16x4_t*>(c), static_cast<const bf16x4_t*>(d), reinterpret_cast<f32x4*>(out));
  else if (form == 2) hipLaunchKernelGGL(mfma_chain_probe_kernel2<0>, dim3(1), dim3(64), 0, s, static_cast<const bf16x8*>(a), static_cast<const bf16x8*>(b), static_cast<const bf16x4_t*>(c), static_cast<const bf16x4_t*>(d), reinterpret_cast<f32x4*>(out));
  else hipLaunchKernelGGL(mfma_chain_probe_kernel1, dim3(1), dim3(64), 0, s, static_cast<const bf16x8*>(a), static_cast<const bf16x8*>(b), static_cast<const bf16x4_t*>(c), static_cast<const bf16x4_t*>(d), reinterpret_cast<f32x4*>(out));
  MICF_RETURN_LAUNCH();
}

// MEASUREMENT PROBE, not a product entry point: `repeats` passes of micf_block_fwd's tile kernel (base 8^3 shape only: C = 192, head_dim
// 16, bf16 mode) in ONE launch with a device-wide barrier between passes.  sync_ws: 2 device ints (barrier counter, error flag; cleared
// here).  MICF_EUNSUPPORTED for any other shape / more workgroups than CUs.  The error flag is 1 after the call if a barrier timed out.
extern "C" int micf_block_fwd_persistent_probe(const micf_block_fwd_group* groups, int ngroups, int B, int D, int H, int W, int C,
                                               int heads, int hidden, float eps, float scale, int dtype, int repeats, int* sync_ws,
                                               micf_stream_t stream) {
  if (!groups || ngroups < 1 || ngroups > 2 || !sync_ws || repeats < 1 || repeats > 64) return MICF_EINVAL;
  if (C != 192 || heads != 12 || hidden != 4 * C || dtype != MICF_DTYPE_BF16) return MICF_EUNSUPPORTED;
  const int TM = micf_block_tile_tokens(B, D, H, W, C, heads, hidden, 0);
  if (TM != 16) return MICF_EUNSUPPORTED;
  BlkFwdArgs a;
  a.att8 = 1;
  for (int i = 0; i < 2; ++i) a.g[i] = groups[i < ngroups ? i : 0];
  a.geo = make_tile_geo(B, D, H, W);
  a.G = ngroups; a.C = C; a.heads = heads; a.hidden = hidden; a.eps = eps; a.scale = scale;
  a.tiles = (a.geo.nwin + TM / 8 - 1) / (TM / 8);
  a.debug = 0;
  const unsigned grid = a.G == 2 ? (unsigned)((a.tiles + 3) / 4 * 8) : (unsigned)a.tiles;
  if (grid > 256) return MICF_EUNSUPPORTED;                      // (one resident workgroup per CU)
  constexpr int NW = 16;
  const size_t lds = block_lds_floats(TM, C, 0, 9 * C + 4 * C, true) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sync_ws, 0, 2 * sizeof(int), s) != hipSuccess) return MICF_ELAUNCH;
  const bool samp = a.g[0].hid != nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_persist_probe_kernel<192, 16, 1, NW, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_persist_probe_kernel<192, 16, 1, NW, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  unsigned* bar = reinterpret_cast<unsigned*>(sync_ws);
  if (samp) hipLaunchKernelGGL((block_fwd_persist_probe_kernel<192, 16, 1, NW, true, true>), dim3(grid), dim3(64 * NW), lds, s, a, repeats, bar, sync_ws + 1);
  else hipLaunchKernelGGL((block_fwd_persist_probe_kernel<192, 16, 1, NW, true, false>), dim3(grid), dim3(64 * NW), lds, s, a, repeats, bar, sync_ws + 1);
  MICF_RETURN_LAUNCH();
}
