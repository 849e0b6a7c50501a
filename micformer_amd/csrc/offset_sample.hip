// offset_sample.hip -- the deformable re-sampling tail of CrossTransformerBlock3D.forward_part1, fused per token:
//   LayerNormProxy(16) -> GELU -> Conv3d(16->3, k1, no bias)          (MS.py:263-273, 315-317)
//   + _get_ref_points (divisors permuted exactly as MS.py:333-335)       (MS.py:326-337, 360-364)
//   + SpatialTransformer: idx + flow -> 2*(new/(S-1) - .5) -> grid_sample(trilinear, zeros, align_corners=False)
//                                                                         (STN.py:9-32, MS.py:379)
// One 64-lane wave per token: the 16-wide offset head is computed redundantly in each 16-lane group, then the lanes
// stride over the C channels of the 8 tap rows (each a contiguous channels-last token row: coalesced).
// Backward reduces d(flow) across the wave and runs the 16-wide head backwards, accumulating the tiny parameter gradients per
// wave before one atomic flush.  d(xa) is a scatter (taps of neighbouring tokens collide).  Device-scope fp32 atomics cost a
// fabric transaction each (8 taps x C channels per token: 317 us for the 32^3 x 2 stage), so on big grids the scatter is
// turned into a GATHER: every token registers itself in the list of its base cell floor(coord) (one int atomic per token,
// capacity kCellCap, overflow -> a small atomic fallback pass), and a second kernel walks, for every voxel, the lists of
// the 8 cells that have it as a corner, accumulating w * dxs[token] rows with coalesced 16-byte loads and ONE plain
// read-modify-write per output element.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "sampler_common.h"

namespace micf {

// tokens walked by one wave: 8 amortises the per-wave flush of the head's parameter gradients on big grids; small grids
// (8^3, 4^3 stages) take 1 so that every CU gets a wave
static inline int tok_per_wave(int64_t T) { return T >= 32768 ? 8 : (T >= 4096 ? 2 : 1); }
// quad kernels: iterations of 4 tokens per wave
static inline int quads_per_wave(int64_t T) { return T >= 32768 ? 2 : 1; }
static inline int quad_waves_per_block(int64_t T) { return T >= 16384 ? 4 : 1; }

__device__ __forceinline__ void offset_sample_fwd_body(const float* __restrict__ h, const float* __restrict__ ln_g,
                                                                const float* __restrict__ ln_b, const float* __restrict__ w1,
                                                                const float* __restrict__ xa, float* __restrict__ flow_out,
                                                                float* __restrict__ xs, Geo g, int C, float eps, int tpw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane & 15;
  const int64_t T = g.tokens();
  for (int it = 0; it < tpw; ++it) {
    const int64_t t = ((int64_t)blockIdx.x * 4 + wave) * tpw + it;
    if (t >= T) return;
    float xh, rs, ln, gl, off[3];
    head_fwd(h + t * kHid, ln_g, ln_b, w1, eps, k, xh, rs, ln, gl, off);
    int b, d, hh, w; g.decode((int)t, b, d, hh, w);
    float fl[3];
    fl[0] = off[0] + (((float)d + 0.5f) / (float)g.H * 2.f - 1.f);      // MS.py:335  ref[...,0] /= H_key
    fl[1] = off[1] + (((float)hh + 0.5f) / (float)g.W * 2.f - 1.f);     // MS.py:334  ref[...,1] /= W_key
    fl[2] = off[2] + (((float)w + 0.5f) / (float)g.D * 2.f - 1.f);      // MS.py:333  ref[...,2] /= D_key
    if (lane < 3) flow_out[t * 3 + lane] = fl[lane];
    const Taps tp = make_taps(d, hh, w, fl, g.D, g.H, g.W);
    const float* base = xa + (int64_t)b * g.D * g.H * g.W * C;
    int lin[8]; float wgt[8]; bool ok[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
      ok[q] = tp.finite && corner(tp, dz, dy, dx, g.D, g.H, g.W, lin[q]);
      const float wx = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
      const float wy = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
      const float wz = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
      wgt[q] = wx * wy * wz;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {                            // out-of-volume taps: row 0 with weight 0 -- all 8 loads issue back to back
      lin[q] = ok[q] ? lin[q] : 0;
      wgt[q] = ok[q] ? wgt[q] : 0.f;
    }
    for (int c = lane; c < C; c += 64) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = base[(int64_t)lin[q] * C + c];
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += ok[q] ? v[q] * wgt[q] : 0.f;
      xs[t * C + c] = acc;
    }
  }
}


// ---- quad variants: FOUR tokens per wave, one per 16-lane group.  The 16-wide head is computed once per group (not 4x
// redundantly), the lanes of a group cover the C channels with 16-byte loads, and four tokens' dependent chains
// (h row -> LN/GELU/1^3 conv -> taps -> gather) are in flight per wave instead of one: these kernels are latency-bound.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ void offset_sample_fwd4_body(const float* __restrict__ h, const float* __restrict__ ln_g,
                                                                 const float* __restrict__ ln_b, const float* __restrict__ w1,
                                                                 const float* __restrict__ xa, float* __restrict__ flow_out,
                                                                 float* __restrict__ xs, Geo g, int C, float eps, int tpw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane & 15, grp = lane >> 4;
  const int64_t T = g.tokens();
  for (int it = 0; it < tpw; ++it) {
    const int64_t t0 = (((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * tpw + it) * 4;
    if (t0 >= T) return;                                     // wave-uniform
    const bool live = t0 + grp < T;
    const int64_t t = live ? t0 + grp : T - 1;               // idle groups shadow the last token (no stores)
    float xh, rs, ln, gl, off[3];
    head_fwd(h + t * kHid, ln_g, ln_b, w1, eps, k, xh, rs, ln, gl, off);
    int b, d, hh, w; g.decode((int)t, b, d, hh, w);
    float fl[3];
    fl[0] = off[0] + (((float)d + 0.5f) / (float)g.H * 2.f - 1.f);      // MS.py:335  ref[...,0] /= H_key
    fl[1] = off[1] + (((float)hh + 0.5f) / (float)g.W * 2.f - 1.f);     // MS.py:334  ref[...,1] /= W_key
    fl[2] = off[2] + (((float)w + 0.5f) / (float)g.D * 2.f - 1.f);      // MS.py:333  ref[...,2] /= D_key
    if (live && k < 3) flow_out[t * 3 + k] = k == 0 ? fl[0] : (k == 1 ? fl[1] : fl[2]);
    const Taps tp = make_taps(d, hh, w, fl, g.D, g.H, g.W);
    const float* base = xa + (int64_t)b * g.D * g.H * g.W * C;
    int lin[8]; float wgt[8]; bool ok[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
      ok[q] = tp.finite && corner(tp, dz, dy, dx, g.D, g.H, g.W, lin[q]);
      const float wx = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
      const float wy = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
      const float wz = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
      wgt[q] = wx * wy * wz;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) lin[q] = ok[q] ? lin[q] : 0;  // out-of-volume taps: row 0, masked -- all 8 loads issue back to back
    for (int c = 4 * k; c < C; c += 64) {
      float4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = ld4(base + (int64_t)lin[q] * C + c);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (ok[q]) { acc.x += v[q].x * wgt[q]; acc.y += v[q].y * wgt[q]; acc.z += v[q].z * wgt[q]; acc.w += v[q].w * wgt[q]; }
      if (live) *reinterpret_cast<float4*>(xs + t * C + c) = acc;
    }
  }
}

constexpr int kCellCap = 8;      // tokens per cell list


// cell of a token: base corner floor(coord) + 1 per axis, or -1 when no corner of the cell lies inside the volume
__device__ __forceinline__ int cell_of(const Taps& tp, int b, int D, int H, int W) {
  if (!tp.finite) return -1;
  if (!(tp.z0 >= -1.f && tp.z0 <= (float)(D - 1) && tp.y0 >= -1.f && tp.y0 <= (float)(H - 1) && tp.x0 >= -1.f && tp.x0 <= (float)(W - 1)))
    return -1;
  return ((b * (D + 1) + (int)tp.z0 + 1) * (H + 1) + (int)tp.y0 + 1) * (W + 1) + (int)tp.x0 + 1;
}

constexpr int kBwd4TapBatch = 8;     // tap rows in flight per batch of the 4-tokens-per-wave adjoint
// How d(xa) -- a scatter: the taps of neighbouring tokens collide -- is produced (template parameter MODE of the backward kernels):
//   kScatter  every (tap, channel) term is a device-scope atomic add                     (no workspace; odd channel counts)
//   kCells    tokens register in per-cell lists, sample_gather_kernel walks them         (round 1-4 path for >= 4096 tokens; test hook "sample_tile" = 0)
//   kTile     NEAR tokens (base cell within `near_e` voxels of the token's own position on every axis -- in this network the
//             displacement is ref in (-1, 1) plus a small learned offset) contribute nothing here: sample_gather_tile_kernel
//             finds them again by scanning a bounded neighbourhood of every output tile and sums their terms in LDS;
//             FAR tokens (normally none) take the atomic path right here, before that kernel's plain read-modify-write of d(xa)
enum { kScatter = 0, kCells = 1, kTile = 2 };

// near <=> -E <= floor(coord) - index <= E - 1 on every axis: then BOTH corners of the cell lie within E voxels of the token on
// that axis, i.e. the token is inside the candidate box (tile grown by E) of every output tile its corners fall into
__device__ __forceinline__ bool near_token(const Taps& tp, int d, int h, int w, int E) {
  if (!tp.finite) return false;
  const float lo = -(float)E, hi = (float)(E - 1);
  const float dz = tp.z0 - (float)d, dy = tp.y0 - (float)h, dx = tp.x0 - (float)w;
  return dz >= lo && dz <= hi && dy >= lo && dy <= hi && dx >= lo && dx <= hi;
}

template <int MODE>
__device__ __forceinline__ void offset_sample_bwd_body(
    const float* __restrict__ dxs, const float* __restrict__ h, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
    const float* __restrict__ w1, const float* __restrict__ xa, const float* __restrict__ flow, float* __restrict__ dxa,
    float* __restrict__ dh, float* __restrict__ dln_g, float* __restrict__ dln_b, float* __restrict__ dw1, Geo g, int C, float eps,
    int tpw, CellLists cl, float* __restrict__ partials, int nwaves, int near_e, int bid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane & 15;
  const int64_t T = g.tokens();
  float acc_w[3] = {0.f, 0.f, 0.f}, acc_g = 0.f, acc_b = 0.f;       // per-lane (channel k) partials, lanes 0..15 flush
  for (int it = 0; it < tpw; ++it) {
    const int64_t t = ((int64_t)bid * 4 + wave) * tpw + it;
    if (t >= T) break;
    float xh, rs, ln, gl, off[3];
    head_fwd(h + t * kHid, ln_g, ln_b, w1, eps, k, xh, rs, ln, gl, off);
    int b, d, hh, w; g.decode((int)t, b, d, hh, w);
    const float fl[3] = {flow[t * 3 + 0], flow[t * 3 + 1], flow[t * 3 + 2]};
    const Taps tp = make_taps(d, hh, w, fl, g.D, g.H, g.W);
    const int64_t boff = (int64_t)b * g.D * g.H * g.W * C;
    int lin[8]; bool ok[8]; float wx[8], wy[8], wz[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
      ok[q] = tp.finite && corner(tp, dz, dy, dx, g.D, g.H, g.W, lin[q]);
      wx[q] = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
      wy[q] = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
      wz[q] = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
    }
    float gz = 0.f, gy = 0.f, gx = 0.f;
    const bool scatter = MODE == kScatter || (MODE == kTile && !near_token(tp, d, hh, w, near_e));
    int64_t arow[8];                                         // (unconditional tap loads: see the quad body)
#pragma unroll
    for (int q = 0; q < 8; ++q) arow[q] = boff + (int64_t)(ok[q] ? lin[q] : 0) * C;
    for (int c = lane; c < C; c += 64) {
      const float go = dxs[t * C + c];
      float xv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) xv[q] = xa[arow[q] + c];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
        if (scatter && ok[q]) atomicAdd(dxa + arow[q] + c, wx[q] * wy[q] * wz[q] * go);
        const float val = xv[q] * go;
        // (selects, not multiplications by 0: the weights of a non-finite coordinate are NaN and must not reach the sums)
        gx += ok[q] ? (dx ? val : -val) * wy[q] * wz[q] : 0.f;
        gy += ok[q] ? (dy ? val : -val) * wx[q] * wz[q] : 0.f;
        gz += ok[q] ? (dz ? val : -val) * wx[q] * wy[q] : 0.f;
      }
    }
    if (MODE == kCells && lane == 0) {
      const int cell = cell_of(tp, b, g.D, g.H, g.W);
      if (cell >= 0) {
        const int slot = atomicAdd(cl.count + cell, 1);
        if (slot < cl.cap) cl.list[(int64_t)cell * kCellCap + slot] = (int)t;
        else cl.ovf[atomicAdd(cl.ovf_count, 1)] = (int)t;
      }
    }
    gz = wave_sum(gz); gy = wave_sum(gy); gx = wave_sum(gx);
    // grid_sample: d/dn = d/dcoord * S/2 ; STN.py:24: d/dnew = 2 * d/dn / (S-1)      (S == 1 -> 0/0 = NaN, as the reference)
    float go3[3];
    go3[0] = (2.f * ((float)g.D / 2.f * gz)) / (float)(g.D - 1);
    go3[1] = (2.f * ((float)g.H / 2.f * gy)) / (float)(g.H - 1);
    go3[2] = (2.f * ((float)g.W / 2.f * gx)) / (float)(g.W - 1);
    // 16-wide head backward (lane k)
    float dgl = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { acc_w[a] += go3[a] * gl; dgl += w1[a * kHid + k] * go3[a]; }
    const float dln = dgl * gelu_grad_f(ln);
    acc_g += dln * xh;
    acc_b += dln;
    const float gd = ln_g[k] * dln;
    const float ma = sum16(gd) * (1.f / kHid);
    const float mb = sum16(gd * xh) * (1.f / kHid);
    if (lane < kHid) dh[t * kHid + k] = rs * (gd - ma - xh * mb);
  }
  if (lane < kHid) {
    if (partials) {      // partials[address][wave]: summed by sample_finish_kernel (thousands of waves hitting the same 80 addresses
                         // with device-scope atomics serialise: that, not the scatter, was most of this kernel's time)
      const int wg = bid * 4 + wave;
#pragma unroll
      for (int a = 0; a < 3; ++a) partials[(int64_t)(a * kHid + k) * nwaves + wg] = acc_w[a];
      partials[(int64_t)(3 * kHid + k) * nwaves + wg] = acc_g;
      partials[(int64_t)(4 * kHid + k) * nwaves + wg] = acc_b;
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) atomicAdd(dw1 + a * kHid + k, acc_w[a]);
      atomicAdd(dln_g + k, acc_g);
      atomicAdd(dln_b + k, acc_b);
    }
  }
}



template <int MODE>
__device__ __forceinline__ void offset_sample_bwd4_body(
    const float* __restrict__ dxs, const float* __restrict__ h, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
    const float* __restrict__ w1, const float* __restrict__ xa, const float* __restrict__ flow, float* __restrict__ dxa,
    float* __restrict__ dh, float* __restrict__ dln_g, float* __restrict__ dln_b, float* __restrict__ dw1, Geo g, int C, float eps,
    int tpw, CellLists cl, float* __restrict__ partials, int nwaves, int near_e, int bid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane & 15, grp = lane >> 4;
  const int64_t T = g.tokens();
  float acc_w[3] = {0.f, 0.f, 0.f}, acc_g = 0.f, acc_b = 0.f;       // per-lane (group, channel k) partials
  for (int it = 0; it < tpw; ++it) {
    const int64_t t0 = (((int64_t)bid * (blockDim.x >> 6) + wave) * tpw + it) * 4;
    if (t0 >= T) break;                                      // wave-uniform
    const bool live = t0 + grp < T;
    const int64_t t = live ? t0 + grp : T - 1;
    float xh, rs, ln, gl, off[3];
    head_fwd(h + t * kHid, ln_g, ln_b, w1, eps, k, xh, rs, ln, gl, off);
    int b, d, hh, w; g.decode((int)t, b, d, hh, w);
    const float fl[3] = {flow[t * 3 + 0], flow[t * 3 + 1], flow[t * 3 + 2]};
    const Taps tp = make_taps(d, hh, w, fl, g.D, g.H, g.W);
    const int64_t boff = (int64_t)b * g.D * g.H * g.W * C;
    int lin[8]; bool ok[8]; float wx[8], wy[8], wz[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
      ok[q] = tp.finite && corner(tp, dz, dy, dx, g.D, g.H, g.W, lin[q]);
      wx[q] = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
      wy[q] = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
      wz[q] = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
    }
    float gz = 0.f, gy = 0.f, gx = 0.f;
    // The 8 tap rows are loaded UNCONDITIONALLY (an out-of-volume tap reads row 0 of the sample and is masked afterwards): with a
    // branch per tap the compiler emitted load -> s_waitcnt vmcnt(0) -> FMAs inside each branch, i.e. eight dependent L2 round
    // trips per token (round 5: 95 -> 60 us at 32^3 together with the 4th resident wave).
    const float* xb = xa + boff;
    int rowo[8];                                             // (a sample has < 2^31 elements: 32-bit row offsets)
#pragma unroll
    for (int q = 0; q < 8; ++q) rowo[q] = (ok[q] ? lin[q] : 0) * C;
    for (int c = 4 * k; c < C; c += 64) {
      const float4 go = ld4(dxs + t * C + c);
#pragma unroll
      for (int q0 = 0; q0 < 8; q0 += kBwd4TapBatch) {         // taps in flight per batch
        float4 xv[kBwd4TapBatch];
#pragma unroll
        for (int u = 0; u < kBwd4TapBatch; ++u) xv[u] = ld4(xb + rowo[q0 + u] + c);
#pragma unroll
        for (int u = 0; u < kBwd4TapBatch; ++u) {
          const int q = q0 + u;
          const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
          if (MODE == kScatter && live && ok[q]) {
            const float wq = wx[q] * wy[q] * wz[q];
            float* a = dxa + boff + rowo[q] + c;
            atomicAdd(a, wq * go.x); atomicAdd(a + 1, wq * go.y); atomicAdd(a + 2, wq * go.z); atomicAdd(a + 3, wq * go.w);
          }
          const float val = xv[u].x * go.x + xv[u].y * go.y + xv[u].z * go.z + xv[u].w * go.w;
          // (selects, not multiplications by 0: the weights of a non-finite coordinate are NaN and must not reach the sums)
          gx += ok[q] ? (dx ? val : -val) * wy[q] * wz[q] : 0.f;
          gy += ok[q] ? (dy ? val : -val) * wx[q] * wz[q] : 0.f;
          gz += ok[q] ? (dz ? val : -val) * wx[q] * wy[q] : 0.f;
        }
      }
    }
    if (MODE == kCells || MODE == kTile) {
      // kCells: a token whose cell list is full (normally none) scatters its d(xa) contributions atomically right here -- before the
      // gather launch reads / writes d(xa), so the finishing launch has no overflow pass left and only sums head-parameter partials;
      // kTile: a FAR token (normally none) does the same -- out of the hot loop above, which carries no atomic in either mode
      int ovf = 0;
      if (MODE == kTile) ovf = (live && tp.finite && !near_token(tp, d, hh, w, near_e)) ? 1 : 0;
      if (MODE == kCells && live && k == 0) {
        const int cell = cell_of(tp, b, g.D, g.H, g.W);
        if (cell >= 0) {
          const int slot = atomicAdd(cl.count + cell, 1);
          if (slot < cl.cap) cl.list[(int64_t)cell * kCellCap + slot] = (int)t;
          else ovf = 1;
        }
      }
      if (MODE == kCells) ovf = __shfl(ovf, lane & 48, 64);
      if (ovf) {
        for (int c = 4 * k; c < C; c += 64) {
          const float4 go = ld4(dxs + t * C + c);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (!ok[q]) continue;
            const int64_t a = boff + (int64_t)lin[q] * C + c;
            const float wq = wx[q] * wy[q] * wz[q];
            atomicAdd(dxa + a, wq * go.x); atomicAdd(dxa + a + 1, wq * go.y);
            atomicAdd(dxa + a + 2, wq * go.z); atomicAdd(dxa + a + 3, wq * go.w);
          }
        }
      }
    }
    if (MODE == kCells && live && k < 8 && cl.w8) {           // lane k < 8: the weight of corner k (dz, dy, dx = bits of k)
      const float ux = (k & 1) ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
      const float uy = (k & 2) ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
      const float uz = (k & 4) ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
      cl.w8[t * 8 + k] = ux * uy * uz;
    }
    gz = sum16(gz); gy = sum16(gy); gx = sum16(gx);
    // grid_sample: d/dn = d/dcoord * S/2 ; STN.py:24: d/dnew = 2 * d/dn / (S-1)      (S == 1 -> 0/0 = NaN, as the reference)
    float go3[3];
    go3[0] = (2.f * ((float)g.D / 2.f * gz)) / (float)(g.D - 1);
    go3[1] = (2.f * ((float)g.H / 2.f * gy)) / (float)(g.H - 1);
    go3[2] = (2.f * ((float)g.W / 2.f * gx)) / (float)(g.W - 1);
    float dgl = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) dgl += w1[a * kHid + k] * go3[a];
    const float dln = dgl * gelu_grad_f(ln);
    if (live) {
#pragma unroll
      for (int a = 0; a < 3; ++a) acc_w[a] += go3[a] * gl;
      acc_g += dln * xh;
      acc_b += dln;
    }
    const float gd = ln_g[k] * dln;
    const float ma = sum16(gd) * (1.f / kHid);
    const float mb = sum16(gd * xh) * (1.f / kHid);
    if (live) dh[t * kHid + k] = rs * (gd - ma - xh * mb);
  }
  // fold the four groups, then lanes 0..15 flush
#pragma unroll
  for (int a = 0; a < 3; ++a) { acc_w[a] += __shfl_xor(acc_w[a], 16, 64); acc_w[a] += __shfl_xor(acc_w[a], 32, 64); }
  acc_g += __shfl_xor(acc_g, 16, 64); acc_g += __shfl_xor(acc_g, 32, 64);
  acc_b += __shfl_xor(acc_b, 16, 64); acc_b += __shfl_xor(acc_b, 32, 64);
  if (lane < kHid) {
    if (partials) {
      const int wg = bid * (blockDim.x >> 6) + wave;
#pragma unroll
      for (int a = 0; a < 3; ++a) partials[(int64_t)(a * kHid + k) * nwaves + wg] = acc_w[a];
      partials[(int64_t)(3 * kHid + k) * nwaves + wg] = acc_g;
      partials[(int64_t)(4 * kHid + k) * nwaves + wg] = acc_b;
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) atomicAdd(dw1 + a * kHid + k, acc_w[a]);
      atomicAdd(dln_g + k, acc_g);
      atomicAdd(dln_b + k, acc_b);
    }
  }
}

// d(xa)[v, :] += sum over the tokens registered in the 8 cells that have voxel v as a corner.  Thread = (voxel, 4 channels).
__device__ __forceinline__ void sample_gather_body(const float* __restrict__ dxs, const float* __restrict__ flow,
                                                            float* __restrict__ dxa, Geo g, int C, CellLists cl) {
  const int q4 = C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t v = idx / q4;
  if (v >= g.tokens()) return;
  const int c4 = (int)(idx % q4) * 4;
  int b, z, y, x; g.decode((int)v, b, z, y, x);
  float4 acc = *reinterpret_cast<const float4*>(dxa + v * C + c4);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
    const int cell = ((b * (g.D + 1) + z - dz + 1) * (g.H + 1) + y - dy + 1) * (g.W + 1) + x - dx + 1;
    int n = cl.count[cell];
    n = n < cl.cap ? n : cl.cap;
    for (int i = 0; i < n; ++i) {
      const int t = cl.list[(int64_t)cell * kCellCap + i];
      float wgt;
      if (cl.w8) {
        wgt = cl.w8[(int64_t)t * 8 + q];
      } else {
        int tb, td, th, tw; g.decode(t, tb, td, th, tw);
        const float fl[3] = {flow[(int64_t)t * 3 + 0], flow[(int64_t)t * 3 + 1], flow[(int64_t)t * 3 + 2]};
        const Taps tp = make_taps(td, th, tw, fl, g.D, g.H, g.W);
        const float wx = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
        const float wy = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
        const float wz = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
        wgt = wx * wy * wz;
      }
      const float4 go = *reinterpret_cast<const float4*>(dxs + (int64_t)t * C + c4);
      acc.x += wgt * go.x; acc.y += wgt * go.y; acc.z += wgt * go.z; acc.w += wgt * go.w;
    }
  }
  *reinterpret_cast<float4*>(dxa + v * C + c4) = acc;
}

// Second pass of the backward.  (1) blocks 0..79: sum the per-wave partials of one head-parameter gradient element and add it
// to dw1 / dln_g / dln_b (one writer per element).  (2) all blocks: tokens whose cell list was full -- the atomic scatter, one
// wave per token (normally zero tokens).
__device__ __forceinline__ void sample_finish_body(const float* __restrict__ dxs, const float* __restrict__ flow,
                                                            float* __restrict__ dxa, Geo g, int C, CellLists cl,
                                                            const float* __restrict__ partials, int nwaves, float* __restrict__ dw1,
                                                            float* __restrict__ dln_g, float* __restrict__ dln_b) {
  const int lane = threadIdx.x & 63;
  if (partials && blockIdx.x < 5 * kHid) {
    __shared__ float red[4];
    const float* p = partials + (int64_t)blockIdx.x * nwaves;
    float acc = 0.f;
    for (int i = threadIdx.x; i < nwaves; i += 256) acc += p[i];
    acc = wave_sum(acc);
    if (lane == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float tot = red[0] + red[1] + red[2] + red[3];
      const int a = blockIdx.x;
      if (a < 3 * kHid) dw1[a] += tot;
      else if (a < 4 * kHid) dln_g[a - 3 * kHid] += tot;
      else dln_b[a - 4 * kHid] += tot;
    }
  }
  if (!cl.count) return;
  const int n = *cl.ovf_count;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
    const int t = cl.ovf[i];
    int b, d, hh, w; g.decode(t, b, d, hh, w);
    const float fl[3] = {flow[(int64_t)t * 3 + 0], flow[(int64_t)t * 3 + 1], flow[(int64_t)t * 3 + 2]};
    const Taps tp = make_taps(d, hh, w, fl, g.D, g.H, g.W);
    const int64_t boff = (int64_t)b * g.D * g.H * g.W * C;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
      int lin;
      if (!(tp.finite && corner(tp, dz, dy, dx, g.D, g.H, g.W, lin))) continue;
      const float wx = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
      const float wy = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
      const float wz = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
      for (int c = lane; c < C; c += 64) atomicAdd(dxa + boff + (int64_t)lin * C + c, wx * wy * wz * dxs[(int64_t)t * C + c]);
    }
  }
}


// ---- launch form of the kernels above: up to two independent pointer sets of the same shape (the two modalities' offset
// heads of a cross pair) per launch, selected by blockIdx.y
struct SampleFwdSets { SampleFwdSet s[2]; };
struct SampleBwdSets { SampleBwdSet s[2]; };

__global__ void __launch_bounds__(256) offset_sample_fwd_kernel(const SampleFwdSets p, Geo g, int C, float eps, int tpw) {
  const SampleFwdSet& q = p.s[blockIdx.y];
  offset_sample_fwd_body(q.h, q.ln_g, q.ln_b, q.w1, q.xa, q.flow, q.xs, g, C, eps, tpw);
}
__global__ void __launch_bounds__(256) offset_sample_fwd4_kernel(const SampleFwdSets p, Geo g, int C, float eps, int tpw) {
  const SampleFwdSet& q = p.s[blockIdx.y];
  offset_sample_fwd4_body(q.h, q.ln_g, q.ln_b, q.w1, q.xa, q.flow, q.xs, g, C, eps, tpw);
}
template <int MODE>
__global__ void __launch_bounds__(256) offset_sample_bwd_kernel(const SampleBwdSets p, Geo g, int C, float eps, int tpw, int nwaves, int near_e) {
  const SampleBwdSet& q = p.s[blockIdx.y];
  offset_sample_bwd_body<MODE>(q.dxs, q.h, q.ln_g, q.ln_b, q.w1, q.xa, q.flow, q.dxa, q.dh, q.dln_g, q.dln_b, q.dw1, g, C, eps, tpw, q.cl,
                               q.partials, nwaves, near_e, blockIdx.x);
}
// (3 waves per SIMD; 4 = a 128-register cap: 12 spills, same time once the tap loads are unconditional)
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) offset_sample_bwd4_kernel(const SampleBwdSets p, Geo g, int C, float eps, int tpw, int nwaves, int near_e) {
  const SampleBwdSet& q = p.s[blockIdx.y];
  offset_sample_bwd4_body<MODE>(q.dxs, q.h, q.ln_g, q.ln_b, q.w1, q.xa, q.flow, q.dxa, q.dh, q.dln_g, q.dln_b, q.dw1, g, C, eps, tpw, q.cl,
                                q.partials, nwaves, near_e, blockIdx.x);
}

// ---- d(xa) of the NEAR tokens, output-tile-centric (kTile).  A workgroup owns a box of td x th x tw output voxels of one sample:
//   scan     every token of the box grown by E (clipped to the volume) re-derives its base cell from the saved flow (12 B per
//            candidate, all of a lane's candidates requested before the first is looked at); a near token whose cell has a corner
//            inside the box is appended to the box's HIT list in LDS (record = token + source coordinates) and to the short list
//            of its base cell (two LDS integer atomics per hit);
//   lists    two threads per voxel walk the lists of the 4 + 4 cells that have the voxel as a corner (LDS only) and write the
//            voxel's own flat list {token, weight}: every list segment has one writer;
//   sum      thread = (voxel, 4 channels), two voxels at a time: the lists 8 entries at a time -- 16 independent 16-byte loads of
//            d(xs) rows in flight (the rows of a box stay in L1 / L2), then the FMAs -- and ONE plain 16-byte read-modify-write of
//            d(xa) per element: every voxel has exactly one owner in the whole launch.
// Overflow (strongly compressive fields, normally never): a full hit list makes the scanning lane scatter that token's in-box
// terms atomically itself (fenced before the sums); a full cell list or list segment makes the voxel's threads walk the hit list.
// No global lists, counters, memset or weight table (the cell-list path: 4 MB of weights + 2.4 M list entries per 32^3 launch):
// the only inputs are flow and d(xs).
// Measured on the way (MI355X, 32^3 x 2 samples x 2 modalities per launch; the global cell-list gather it replaces: 45 us):
//   box accumulator in LDS, ds_add_f32 per term                       210 us  (LDS float atomics: ~85 cycles per wave instruction)
//   per-cell lists of records, each sum thread walking its 8 cells     46 us  (a branchy loop per cell: the d(xs) loads issue one at a time)
//   per-voxel flat lists built by brute force over the hit list        42 us  (scan 9 + lists 18: 150 hits x 25 VALU instructions per thread + sums 15)
constexpr int kHitCap = 512;       // hits per box (typical: 250-300 for 4 x 4 x 8)
constexpr int kCellCap2 = 6;       // hit indices per cell of the box (typical: 1)
constexpr int kVoxSeg = 12;        // entries per list segment; a voxel has two segments (typical: 8 entries per voxel)
struct TileShape { int td, th, tw, e, nz, ny, nx, ntiles, hit_cap, seg_cap, cell_cap, whole; };   // *_cap: the constants above, less under the test hooks "tile_cap_*"

__device__ __forceinline__ float corner_weight(const float4& hr, int dz, int dy, int dx) {
  const float z0 = floorf(hr.y), y0 = floorf(hr.z), x0 = floorf(hr.w);
  const float wx = dx ? hr.w - x0 : (x0 + 1.f) - hr.w;
  const float wy = dy ? hr.z - y0 : (y0 + 1.f) - hr.z;
  const float wz = dz ? hr.y - z0 : (z0 + 1.f) - hr.y;
  return wx * wy * wz;
}

__device__ __forceinline__ void gather_tile_body(const SampleBwdSet& q, const Geo& g, int C, const TileShape& ts, int bid) {
  extern __shared__ float4 smem4[];
  // XCD-contiguous tile ranges: neighbouring boxes share candidate rows of d(xs) / flow in one L2
  const int per = (ts.ntiles + 7) >> 3;
  int tile = (bid & 7) * per + (bid >> 3);
  if ((bid >> 3) >= per || tile >= ts.ntiles) return;
  const int tx = tile % ts.nx; tile /= ts.nx;
  const int ty = tile % ts.ny; tile /= ts.ny;
  const int tz = tile % ts.nz;
  const int b = tile / ts.nz;
  const int oz = tz * ts.td, oy = ty * ts.th, ox = tx * ts.tw;
  const int ez = min(ts.td, g.D - oz), ey = min(ts.th, g.H - oy), ex = min(ts.tw, g.W - ox);
  const int nvox = ez * ey * ex, q4 = C >> 2, maxvox = ts.td * ts.th * ts.tw;
  const int cy_n = ey + 1, cx_n = ex + 1, ncell = (ez + 1) * cy_n * cx_n, maxcell = (ts.td + 1) * (ts.th + 1) * (ts.tw + 1);
  float4* hits = smem4;                                                        // [kHitCap] {token, cz, cy, cx}
  float2* vlist = reinterpret_cast<float2*>(smem4 + kHitCap);                  // [maxvox][2][kVoxSeg] {token, weight}
  int* vcnt = reinterpret_cast<int*>(vlist + maxvox * 2 * kVoxSeg);            // [maxvox][2]  (-1: walk the hit list instead)
  int* ccnt = vcnt + maxvox * 2;                                               // [maxcell]
  int* cnt = ccnt + maxcell;                                                   // [1] (+ 3 pad)
  unsigned short* clist = reinterpret_cast<unsigned short*>(cnt + 4);          // [maxcell][kCellCap2] hit indices
  const int tid = threadIdx.x;
  for (int i = tid; i < ncell; i += 256) ccnt[i] = 0;
  if (tid == 0) cnt[0] = 0;
  __syncthreads();
  const int E = ts.e;
  const int z_lo = max(oz - E, 0), y_lo = max(oy - E, 0), x_lo = max(ox - E, 0);
  const int nzc = min(oz + ez - 1 + E, g.D - 1) - z_lo + 1, nyc = min(oy + ey - 1 + E, g.H - 1) - y_lo + 1,
            nxc = min(ox + ex - 1 + E, g.W - 1) - x_lo + 1, ncand = nzc * nyc * nxc;
  const float fz0 = (float)oz, fz1 = (float)(oz + ez - 1), fy0 = (float)oy, fy1 = (float)(oy + ey - 1), fx0 = (float)ox,
              fx1 = (float)(ox + ex - 1);
  bool spilled = false;
  for (int i0 = tid; i0 < ncand; i0 += 256 * 4) {
    int tk[4], cd[4], ch[4], cw[4];
    float fl[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                      // all loads of the batch first
      const int i = i0 + 256 * u, ic = i < ncand ? i : i0;
      const int x = ic % nxc, r = ic / nxc;
      cd[u] = z_lo + r / nyc; ch[u] = y_lo + r % nyc; cw[u] = x_lo + x;
      tk[u] = g.token(b, cd[u], ch[u], cw[u]);
      const float* f = q.flow + (int64_t)tk[u] * 3;
      fl[u][0] = f[0]; fl[u][1] = f[1]; fl[u][2] = f[2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + 256 * u >= ncand) continue;
      const Taps tp = make_taps(cd[u], ch[u], cw[u], fl[u], g.D, g.H, g.W);
      if (!near_token(tp, cd[u], ch[u], cw[u], E)) continue;
      if (!(tp.z0 + 1.f >= fz0 && tp.z0 <= fz1 && tp.y0 + 1.f >= fy0 && tp.y0 <= fy1 && tp.x0 + 1.f >= fx0 && tp.x0 <= fx1)) continue;
      const float4 hr = make_float4(__int_as_float(tk[u]), tp.cz, tp.cy, tp.cx);
      const int slot = atomicAdd(cnt, 1);
      if (slot < ts.hit_cap) {
        hits[slot] = hr;
        const int cell = (((int)tp.z0 - oz + 1) * cy_n + ((int)tp.y0 - oy + 1)) * cx_n + ((int)tp.x0 - ox + 1);
        const int cs = atomicAdd(ccnt + cell, 1);
        if (cs < ts.cell_cap) clist[cell * kCellCap2 + cs] = (unsigned short)slot;
        continue;
      }
      // the hit list is full: this lane adds the token's in-box terms to d(xa) itself (fenced before the owners read d(xa))
      spilled = true;
      const int64_t boff = (int64_t)b * g.D * g.H * g.W * C;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int dz = c >> 2, dy = (c >> 1) & 1, dx = c & 1;
        const float z = tp.z0 + dz, y = tp.y0 + dy, xx = tp.x0 + dx;
        if (!(z >= fz0 && z <= fz1 && y >= fy0 && y <= fy1 && xx >= fx0 && xx <= fx1)) continue;
        const float wq = corner_weight(hr, dz, dy, dx);
        float* dst = q.dxa + boff + (int64_t)(((int)z * g.H + (int)y) * g.W + (int)xx) * C;
        const float* src = q.dxs + (int64_t)tk[u] * C;
        for (int cc = 0; cc < C; ++cc) atomicAdd(dst + cc, wq * src[cc]);
      }
    }
  }
  if (__syncthreads_or(spilled)) __threadfence();
  const int nh = min(cnt[0], ts.hit_cap);
  // per-voxel flat lists: thread (voxel, dz) walks the four cells with base z = voxel z - dz
  for (int i = tid; i < 2 * nvox; i += 256) {
    const int v = i >> 1, dz = i & 1;
    const int lx = v % ex, r = v / ex;
    const int ly = r % ey, lz = r / ey;
    float2* seg = vlist + (v * 2 + dz) * kVoxSeg;
    int n = 0;
    bool over = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int dy = c >> 1, dx = c & 1;
      const int cell = ((lz - dz + 1) * cy_n + (ly - dy + 1)) * cx_n + (lx - dx + 1);
      const int cn = ccnt[cell];
      over = over || cn > ts.cell_cap;
      for (int j = 0; j < min(cn, ts.cell_cap); ++j) {
        const float4 hr = hits[clist[cell * kCellCap2 + j]];
        if (n < ts.seg_cap) seg[n] = make_float2(hr.x, corner_weight(hr, dz, dy, dx));
        ++n;
      }
    }
    vcnt[v * 2 + dz] = (over || n > ts.seg_cap) ? -1 : n;
  }
  __syncthreads();
  const int nitem = nvox * q4;
  for (int i0 = tid; i0 < nitem; i0 += 512) {
    float4 acc[2];
    int vv[2], cc4[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int i = i0 + 256 * w < nitem ? i0 + 256 * w : i0;
      vv[w] = i / q4; cc4[w] = (i - vv[w] * q4) * 4;
      acc[w] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int n00 = vcnt[2 * vv[0]], n01 = vcnt[2 * vv[0] + 1], n10 = vcnt[2 * vv[1]], n11 = vcnt[2 * vv[1] + 1];
    if (n00 >= 0 && n01 >= 0 && n10 >= 0 && n11 >= 0) {
      const int na = n00 + n01, nb = n10 + n11;
      for (int base = 0; base < max(na, nb); base += 8) {
        float4 go[2][8]; float wq[2][8];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const float2* l0 = vlist + vv[w] * 2 * kVoxSeg;
          const int n0 = w ? n10 : n00, n = w ? nb : na;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int idx = base + u;
            const bool on = idx < n;
            const float2 e = l0[on ? (idx < n0 ? idx : kVoxSeg + idx - n0) : 0];
            wq[w][u] = on ? e.y : 0.f;
            go[w][u] = ld4(q.dxs + (int64_t)(on ? __float_as_int(e.x) : 0) * C + cc4[w]);
          }
        }
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            acc[w].x += wq[w][u] * go[w][u].x; acc[w].y += wq[w][u] * go[w][u].y;
            acc[w].z += wq[w][u] * go[w][u].z; acc[w].w += wq[w][u] * go[w][u].w;
          }
      }
    } else {
      // a list of one of these voxels overflowed: walk the hit list
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int lx = vv[w] % ex, r = vv[w] / ex;
        const float vz = (float)(oz + r / ey), vy = (float)(oy + r % ey), vx = (float)(ox + lx);
        for (int j = 0; j < nh; ++j) {
          const float4 hr = hits[j];
          const float dz = vz - floorf(hr.y), dy = vy - floorf(hr.z), dx = vx - floorf(hr.w);
          if (!(dz >= 0.f && dz <= 1.f && dy >= 0.f && dy <= 1.f && dx >= 0.f && dx <= 1.f)) continue;
          const float wq = corner_weight(hr, (int)dz, (int)dy, (int)dx);
          const float4 go = ld4(q.dxs + (int64_t)__float_as_int(hr.x) * C + cc4[w]);
          acc[w].x += wq * go.x; acc[w].y += wq * go.y; acc[w].z += wq * go.z; acc[w].w += wq * go.w;
        }
      }
    }
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      if (i0 + 256 * w >= nitem) continue;
      const int lx = vv[w] % ex, r = vv[w] / ex;
      const int64_t tok = g.token(b, oz + r / ey, oy + r % ey, ox + lx);
      float4* dst = reinterpret_cast<float4*>(q.dxa + tok * C + cc4[w]);
      float4 o = *dst;
      o.x += acc[w].x; o.y += acc[w].y; o.z += acc[w].z; o.w += acc[w].w;
      *dst = o;
    }
  }
}

__global__ void __launch_bounds__(256) sample_gather_tile_kernel(const SampleBwdSets p, Geo g, int C, TileShape ts) {
  gather_tile_body(p.s[blockIdx.y], g, C, ts, (int)blockIdx.x);
}

// Small grids (a whole sample <= kHitCap tokens: the 8^3 / 4^3 stages): the candidate box IS the sample, so no token is far, no
// atomic ever touches d(xa), and the two jobs are independent -- ONE launch: workgroups [0, token_blocks) run the per-token
// adjoint (d(flow), the 16-wide head backwards, dh, parameter-gradient partials), the rest one output box each.
template <bool QUAD>
__global__ void __launch_bounds__(256) sample_bwd_fused_kernel(const SampleBwdSets p, Geo g, int C, float eps, int tpw, int nwaves,
                                                               TileShape ts, int token_blocks) {
  const SampleBwdSet& q = p.s[blockIdx.y];
  if ((int)blockIdx.x >= token_blocks)
    gather_tile_body(q, g, C, ts, (int)blockIdx.x - token_blocks);
  else if (QUAD)
    offset_sample_bwd4_body<kTile>(q.dxs, q.h, q.ln_g, q.ln_b, q.w1, q.xa, q.flow, q.dxa, q.dh, q.dln_g, q.dln_b, q.dw1, g, C, eps, tpw, q.cl,
                                   q.partials, nwaves, ts.e, (int)blockIdx.x);
  else
    offset_sample_bwd_body<kTile>(q.dxs, q.h, q.ln_g, q.ln_b, q.w1, q.xa, q.flow, q.dxa, q.dh, q.dln_g, q.dln_b, q.dw1, g, C, eps, tpw, q.cl,
                                  q.partials, nwaves, ts.e, (int)blockIdx.x);
}

__global__ void __launch_bounds__(256) sample_gather_kernel(const SampleBwdSets p, Geo g, int C) {
  const SampleBwdSet& q = p.s[blockIdx.y];
  sample_gather_body(q.dxs, q.flow, q.dxa, g, C, q.cl);
}
__global__ void __launch_bounds__(256) sample_finish_kernel(const SampleBwdSets p, Geo g, int C, int nwaves) {
  const SampleBwdSet& q = p.s[blockIdx.y];
  sample_finish_body(q.dxs, q.flow, q.dxa, g, C, q.cl, q.partials, nwaves, q.dw1, q.dln_g, q.dln_b);
}

// The finishing sums of SEVERAL deferred backward calls (different grids / layers) in ONE launch: blockIdx.y = set.
constexpr int kFinishMany = 32;
struct FinishSets { const float* partials[kFinishMany]; float* dw1[kFinishMany]; float* dln_g[kFinishMany]; float* dln_b[kFinishMany]; int nwaves[kFinishMany]; };
__global__ void __launch_bounds__(256) sample_finish_many_kernel(const FinishSets f, const Geo unit) {
  const int k = blockIdx.y;
  const CellLists none{nullptr, nullptr, nullptr, nullptr, 0, nullptr};
  sample_finish_body(nullptr, nullptr, nullptr, unit, 0, none, f.partials[k], f.nwaves[k], f.dw1[k], f.dln_g[k], f.dln_b[k]);
}

// ---- standalone SpatialTransformer (STN.py:9-32) on channels-last src with a GIVEN flow [T,3] (voxel units, z,y,x)
__global__ void __launch_bounds__(256) stn_fwd_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                      float* __restrict__ out, Geo g, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * 4 + wave;
  if (t >= g.tokens()) return;
  int b, d, hh, w; g.decode((int)t, b, d, hh, w);
  const float fl[3] = {flow[t * 3 + 0], flow[t * 3 + 1], flow[t * 3 + 2]};
  const Taps tp = make_taps(d, hh, w, fl, g.D, g.H, g.W);
  const float* base = src + (int64_t)b * g.D * g.H * g.W * C;
  int lin[8]; float wgt[8]; bool ok[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
    ok[q] = tp.finite && corner(tp, dz, dy, dx, g.D, g.H, g.W, lin[q]);
    const float wx = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
    const float wy = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
    const float wz = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
    wgt[q] = wx * wy * wz;
  }
  for (int c = lane; c < C; c += 64) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (ok[q]) acc += base[(int64_t)lin[q] * C + c] * wgt[q];
    out[t * C + c] = acc;
  }
}

__global__ void __launch_bounds__(256) stn_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ src,
                                                      const float* __restrict__ flow, float* __restrict__ dsrc,
                                                      float* __restrict__ dflow, Geo g, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * 4 + wave;
  if (t >= g.tokens()) return;
  int b, d, hh, w; g.decode((int)t, b, d, hh, w);
  const float fl[3] = {flow[t * 3 + 0], flow[t * 3 + 1], flow[t * 3 + 2]};
  const Taps tp = make_taps(d, hh, w, fl, g.D, g.H, g.W);
  const int64_t boff = (int64_t)b * g.D * g.H * g.W * C;
  int lin[8]; bool ok[8]; float wx[8], wy[8], wz[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
    ok[q] = tp.finite && corner(tp, dz, dy, dx, g.D, g.H, g.W, lin[q]);
    wx[q] = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
    wy[q] = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
    wz[q] = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
  }
  float gz = 0.f, gy = 0.f, gx = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float go = dout[t * C + c];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (!ok[q]) continue;
      const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
      const int64_t a = boff + (int64_t)lin[q] * C + c;
      if (dsrc) atomicAdd(dsrc + a, wx[q] * wy[q] * wz[q] * go);
      const float val = src[a] * go;
      gx += (dx ? val : -val) * wy[q] * wz[q];
      gy += (dy ? val : -val) * wx[q] * wz[q];
      gz += (dz ? val : -val) * wx[q] * wy[q];
    }
  }
  gz = wave_sum(gz); gy = wave_sum(gy); gx = wave_sum(gx);
  if (dflow && lane == 0) {
    dflow[t * 3 + 0] = (2.f * ((float)g.D / 2.f * gz)) / (float)(g.D - 1);
    dflow[t * 3 + 1] = (2.f * ((float)g.H / 2.f * gy)) / (float)(g.H - 1);
    dflow[t * 3 + 2] = (2.f * ((float)g.W / 2.f * gx)) / (float)(g.W - 1);
  }
}

}  // namespace micf
using namespace micf;

// forward of 1 or 2 offset heads of the same shape: ONE launch
int micf::offset_sample_fwd_groups(const SampleFwdSet* sets, int n, int B, int D, int H, int W, int C, float eps, hipStream_t stream) {
  if (!sets || n < 1 || n > 2 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MICF_EINVAL;
  SampleFwdSets p;
  bool al = true;
  for (int i = 0; i < 2; ++i) {
    p.s[i] = sets[i < n ? i : 0];
    const SampleFwdSet& q = p.s[i];
    if (!q.h || !q.ln_g || !q.ln_b || !q.w1 || !q.xa || !q.flow || !q.xs) return MICF_EINVAL;
    al = al && aligned16(q.xa) && aligned16(q.xs);
  }
  const Geo g{B, D, H, W};
  if (g.tokens() >= (1LL << 31)) return MICF_EUNSUPPORTED;
  if (g.tokens() >= 4096 && C % 4 == 0 && al) {   // tiny grids: one token per wave spreads wider
    const int qpw = quads_per_wave(g.tokens());
    const int wpb = quad_waves_per_block(g.tokens());      // small grids: one-wave workgroups, so every CU gets one
    hipLaunchKernelGGL(offset_sample_fwd4_kernel, dim3(ceil_div(g.tokens(), 4 * wpb * qpw), n), dim3(64 * wpb), 0, stream, p, g, C, eps, qpw);
    MICF_RETURN_LAUNCH();
  }
  const int tpw = tok_per_wave(g.tokens());
  const int blocks = ceil_div(g.tokens(), 4 * tpw);
  hipLaunchKernelGGL(offset_sample_fwd_kernel, dim3(blocks, n), dim3(256), 0, stream, p, g, C, eps, tpw);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_offset_sample_fwd(const float* h, const float* ln_g, const float* ln_b, const float* w1,
                                      const float* xa, float* flow, float* xs, int B, int D, int H, int W, int C,
                                      float eps, micf_stream_t stream) {
  const SampleFwdSet one{h, ln_g, ln_b, w1, xa, flow, xs};
  return offset_sample_fwd_groups(&one, 1, B, D, H, W, C, eps, (hipStream_t)stream);
}

static int64_t cell_count(int B, int D, int H, int W) { return (int64_t)B * (D + 1) * (H + 1) * (W + 1); }
static int64_t partial_floats(int64_t T) {           // [80][waves], rounded up to a 16-byte multiple
  const int64_t waves = ceil_div(T, (int64_t)4 * tok_per_wave(T)) * 4;
  return (5 * kHid * waves + 3) / 4 * 4;
}
static bool use_cells(int64_t T) { return T >= 4096; }

// Output-tile shape of sample_gather_tile_kernel for a grid: big grids take 4 x 4 x 8 voxels (the candidate box grown by E = 3 is
// 10 x 10 x 14 tokens, 11 x the tile, 12 B each), small ones smaller boxes so that the launch still spreads over the chip.
// false: test hook "sample_tile" = 0 (then the cell lists / atomics take the grid).
static bool tile_shape(const Geo& g, int C, TileShape& ts, size_t& lds_bytes) {
  const Options& opt = options();
  if (opt.sample_tile == 0) return false;
  const bool e_hook = opt.sample_e >= 0;                    // hook: radius of the NEAR neighbourhood; 0 = every token takes the far path
  const int e_env = e_hook ? opt.sample_e : 3;
  const int64_t T = g.tokens();
  int td = 4, th = 4, tw = 8;
  if (T < 32768) td = 2;
  if (T < 4096) { th = 2; tw = 4; }
  ts.td = td < g.D ? td : g.D; ts.th = th < g.H ? th : g.H; ts.tw = tw < g.W ? tw : g.W;
  ts.e = e_env < 0 ? 0 : (e_env > 8 ? 8 : e_env);
  // a small sample (<= 512 tokens: the 8^3 / 4^3 stages): the candidate box is the whole sample (E = its largest extent) -- no far
  // tokens exist, the hit list cannot overflow.  (Taking the 16^3 stage too -- 4096 candidates per box, a full hit list handled
  // inside the workgroup -- measured 66 -> 61 us per pair alone but 66 -> 85 us inside the step: not done.)
  ts.whole = (!e_hook && (int64_t)g.D * g.H * g.W <= (int64_t)kHitCap) ? 1 : 0;
  if (ts.whole) ts.e = g.D > g.H ? (g.D > g.W ? g.D : g.W) : (g.H > g.W ? g.H : g.W);
  ts.nz = ceil_div(g.D, ts.td); ts.ny = ceil_div(g.H, ts.th); ts.nx = ceil_div(g.W, ts.tw);
  const int64_t nt = (int64_t)g.B * ts.nz * ts.ny * ts.nx;
  if (nt >= (1LL << 28)) return false;
  ts.ntiles = (int)nt;
  // test hooks "tile_cap_hits / _voxel / _cell" shrink the LDS lists to force their overflow paths (0: every hit / every voxel)
  auto cap = [](int hook, int full) { return hook < 0 ? full : (hook > full ? full : hook); };
  ts.hit_cap = cap(opt.tile_cap_hits, kHitCap); ts.seg_cap = cap(opt.tile_cap_voxel, kVoxSeg); ts.cell_cap = cap(opt.tile_cap_cell, kCellCap2);
  const size_t maxvox = (size_t)ts.td * ts.th * ts.tw, maxcell = (size_t)(ts.td + 1) * (ts.th + 1) * (ts.tw + 1);
  // 4 x 4 x 8: hits 8 KB + voxel lists 24 KB + counters 2 KB + cell lists 2.7 KB
  lds_bytes = (size_t)kHitCap * 16 + maxvox * 2 * kVoxSeg * 8 + maxvox * 2 * 4 + maxcell * 4 + 16 + maxcell * kCellCap2 * 2 + 16;
  return true;
}

// launch shape of the backward's per-token kernel (the finishing launch sums `blocks * wpb` partial rows: one formula for both)
struct BwdShape { bool quad, tiles, fused; int tpw, wpb, blocks; TileShape ts; size_t tile_lds; };
static void bwd_shape(const Geo& g, int C, bool al, BwdShape& sh) {
  const int64_t T = g.tokens();
  // quad kernels (4 tokens per wave) when the channel rows allow 16-byte accesses
  sh.quad = T >= 4096 && (C % 4 == 0) && al;              // tiny grids: 1 token per wave
  // d(xa): output boxes summed from LDS lists (kTile, every grid) | cell lists (hook "sample_tile" = 0, >= 4096 tokens) | atomics
  sh.tile_lds = 0;
  sh.tiles = (C % 4 == 0) && al && tile_shape(g, C, sh.ts, sh.tile_lds);
  sh.fused = sh.tiles && sh.ts.whole;                      // (one launch: 256-thread workgroups in both roles)
  sh.tpw = sh.quad ? quads_per_wave(T) : tok_per_wave(T);
  sh.wpb = sh.quad ? (sh.fused ? 4 : quad_waves_per_block(T)) : 4;
  sh.blocks = ceil_div(T, (sh.quad ? 4 * sh.wpb : 4) * sh.tpw);
}

extern "C" int64_t micf_offset_sample_bwd_workspace(int B, int D, int H, int W) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  const int64_t T = (int64_t)B * D * H * W;
  int64_t need = partial_floats(T);
  if (use_cells(T)) {
    const int64_t nc = (cell_count(B, D, H, W) + 3) / 4 * 4;
    need += nc + 4 + nc * kCellCap + T + 8 * T;
  }
  return need;
}

// backward of 1 or 2 offset heads of the same shape.  `sets[i]` carries the tensors (cl / partials are filled in here);
// workspace: n * micf_offset_sample_bwd_workspace floats (or NULL: atomic scatter, atomic parameter gradients).  The
// parameter-gradient partial sums are finished by the last launch; with `defer_finish` (allowed when no cell lists are in
// use, i.e. small grids) that launch is left to the caller: offset_sample_bwd_finish_groups with the same arguments.
// phase: 0 = everything; 1 = leave the finishing launch (head-parameter partial sums) to a later call when no cell lists are in
// use (small grids: nothing on the data path waits for it); 2 = that finishing launch only (same workspace, same arguments).
int micf::offset_sample_bwd_groups(SampleBwdSet* sets, int n, int B, int D, int H, int W, int C, float eps, float* workspace,
                             int64_t workspace_floats, hipStream_t s, int phase) {
  if (!sets || n < 1 || n > 2 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MICF_EINVAL;
  const Geo g{B, D, H, W};
  const int64_t T = g.tokens();
  if (T >= (1LL << 31) / 4) return MICF_EUNSUPPORTED;
  // the kernels address the tap rows of ONE sample with 32-bit element offsets (offset_sample_bwd4: rowo = lin * C)
  if ((int64_t)D * H * W * C >= (1LL << 31)) return MICF_EUNSUPPORTED;
  bool al = true;
  for (int i = 0; i < n; ++i) {
    const SampleBwdSet& q = sets[i];
    if (!q.dxs || !q.h || !q.ln_g || !q.ln_b || !q.w1 || !q.xa || !q.flow || !q.dxa || !q.dh || !q.dln_g || !q.dln_b || !q.dw1)
      return MICF_EINVAL;
    al = al && aligned16(q.dxs) && aligned16(q.xa) && aligned16(q.dxa);
  }
  BwdShape sh;
  bwd_shape(g, C, al, sh);
  const bool quad = sh.quad, tiles = sh.tiles, fused = sh.fused;
  const int tpw = sh.tpw, wpb = sh.wpb, blocks = sh.blocks, nwaves = sh.blocks * sh.wpb;
  const TileShape& ts = sh.ts;
  const size_t tile_lds = sh.tile_lds;
  const CellLists none{nullptr, nullptr, nullptr, nullptr, 0, nullptr};
  const int64_t per = micf_offset_sample_bwd_workspace(B, D, H, W);
  const bool have_ws = workspace && workspace_floats >= per * n && aligned16(workspace);
  const bool cells = !tiles && have_ws && use_cells(T) && quad && cell_count(B, D, H, W) * kCellCap < (1LL << 31);
  SampleBwdSets p;
  const int64_t nc = (cell_count(B, D, H, W) + 3) / 4 * 4;
  // workspace: [partials g0 | partials g1 | counters g0 | counters g1 | lists + overflow g0 | lists + overflow g1]
  // (the counters of both groups are contiguous: one memset)
  int* counters = have_ws ? reinterpret_cast<int*>(workspace + n * partial_floats(T)) : nullptr;
  int* lists = counters ? counters + n * (nc + 4) : nullptr;
  int cap = options().cell_cap;                             // test hook: a smaller list forces the overflow pass
  cap = cap < 0 ? kCellCap : (cap > kCellCap ? kCellCap : cap);
  for (int i = 0; i < 2; ++i) {
    const int k = i < n ? i : 0;
    p.s[i] = sets[k];
    p.s[i].partials = have_ws ? workspace + k * partial_floats(T) : nullptr;
    p.s[i].cl = none;
    if (cells) {
      int* cnt = counters + k * (nc + 4);
      int* ls = lists + k * (nc * kCellCap + T);
      float* w8 = reinterpret_cast<float*>(lists + n * (nc * kCellCap + T)) + (int64_t)k * 8 * T;
      p.s[i].cl = CellLists{cnt, cnt + nc, ls, ls + nc * kCellCap, cap, w8};
    }
  }
  if (phase == 2) {
    if (!have_ws) return MICF_EINVAL;
    hipLaunchKernelGGL(sample_finish_kernel, dim3(5 * kHid, n), dim3(256), 0, s, p, g, C, nwaves);
    MICF_RETURN_LAUNCH();
  }
  if (cells && hipMemsetAsync(counters, 0, sizeof(int) * (size_t)(n * (nc + 4)), s) != hipSuccess) return MICF_ELAUNCH;
  const int mode = tiles ? kTile : (cells ? kCells : kScatter);
  const dim3 grid(blocks, n), blk(64 * wpb);
  if (fused) {
    const int tb = 8 * ((ts.ntiles + 7) / 8);
    if (quad) hipLaunchKernelGGL(sample_bwd_fused_kernel<true>, dim3(blocks + tb, n), dim3(256), tile_lds, s, p, g, C, eps, tpw, nwaves, ts, blocks);
    else hipLaunchKernelGGL(sample_bwd_fused_kernel<false>, dim3(blocks + tb, n), dim3(256), tile_lds, s, p, g, C, eps, tpw, nwaves, ts, blocks);
    if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
    if (!have_ws || phase == 1) return MICF_OK;
    hipLaunchKernelGGL(sample_finish_kernel, dim3(5 * kHid, n), dim3(256), 0, s, p, g, C, nwaves);
    MICF_RETURN_LAUNCH();
  }
  if (quad) {
    if (mode == kTile) hipLaunchKernelGGL(offset_sample_bwd4_kernel<kTile>, grid, blk, 0, s, p, g, C, eps, tpw, nwaves, ts.e);
    else if (mode == kCells) hipLaunchKernelGGL(offset_sample_bwd4_kernel<kCells>, grid, blk, 0, s, p, g, C, eps, tpw, nwaves, 0);
    else hipLaunchKernelGGL(offset_sample_bwd4_kernel<kScatter>, grid, blk, 0, s, p, g, C, eps, tpw, nwaves, 0);
  } else {
    if (mode == kTile) hipLaunchKernelGGL(offset_sample_bwd_kernel<kTile>, grid, blk, 0, s, p, g, C, eps, tpw, nwaves, ts.e);
    else hipLaunchKernelGGL(offset_sample_bwd_kernel<kScatter>, grid, blk, 0, s, p, g, C, eps, tpw, nwaves, 0);
  }
  if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
  if (tiles) {
    hipLaunchKernelGGL(sample_gather_tile_kernel, dim3((unsigned)(8 * ((ts.ntiles + 7) / 8)), n), dim3(256), tile_lds, s, p, g, C, ts);
    if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
  }
  if (!have_ws) return MICF_OK;                      // (atomic parameter gradients: nothing to finish)
  if (cells) {
    const int64_t threads = T * (C / 4);
    hipLaunchKernelGGL(sample_gather_kernel, dim3((unsigned)((threads + 255) / 256), n), dim3(256), 0, s, p, g, C);
    if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
  }
  if (phase == 1) return MICF_OK;                    // the caller finishes later (micf_offset_head_bwd_finish[_grouped])
  hipLaunchKernelGGL(sample_finish_kernel, dim3(5 * kHid, n), dim3(256), 0, s, p, g, C, nwaves);
  MICF_RETURN_LAUNCH();
}

// 1 when a sampler backward at this grid leaves only head-parameter partial sums to its finishing launch: that launch may then run
// anywhere later on a stream ordered after the call, given the same workspace.  (Every grid since cell-list overflow is scattered
// by the backward kernel itself.)
extern "C" int micf_offset_head_finish_deferrable(int B, int D, int H, int W) {
  return (B <= 0 || D <= 0 || H <= 0 || W <= 0) ? 0 : 1;
}

extern "C" int micf_offset_sample_bwd(const float* dxs, const float* h, const float* ln_g, const float* ln_b,
                                      const float* w1, const float* xa, const float* flow, float* dxa, float* dh,
                                      float* dln_g, float* dln_b, float* dw1, int B, int D, int H, int W, int C, float eps,
                                      float* workspace, int64_t workspace_floats, micf_stream_t stream) {
  SampleBwdSet one{dxs, h, ln_g, ln_b, w1, xa, flow, dxa, dh, dln_g, dln_b, dw1, CellLists{nullptr, nullptr, nullptr, nullptr, 0, nullptr}, nullptr};
  return offset_sample_bwd_groups(&one, 1, B, D, H, W, C, eps, workspace, workspace_floats, (hipStream_t)stream);
}

extern "C" int micf_stn_fwd(const float* src, const float* flow, float* out, int B, int D, int H, int W, int C,
                            micf_stream_t stream) {
  if (!src || !flow || !out || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MICF_EINVAL;
  const Geo g{B, D, H, W};
  if (g.tokens() >= (1LL << 31)) return MICF_EUNSUPPORTED;
  hipLaunchKernelGGL(stn_fwd_kernel, dim3(ceil_div(g.tokens(), 4)), dim3(256), 0, (hipStream_t)stream, src, flow, out, g, C);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_stn_bwd(const float* dout, const float* src, const float* flow, float* dsrc, float* dflow, int B, int D,
                            int H, int W, int C, micf_stream_t stream) {
  if (!dout || !src || !flow || (!dsrc && !dflow) || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MICF_EINVAL;
  const Geo g{B, D, H, W};
  if (g.tokens() >= (1LL << 31)) return MICF_EUNSUPPORTED;
  hipLaunchKernelGGL(stn_bwd_kernel, dim3(ceil_div(g.tokens(), 4)), dim3(256), 0, (hipStream_t)stream, dout, src, flow, dsrc,
                     dflow, g, C);
  MICF_RETURN_LAUNCH();
}

// calls[i]: the (sets, n, grid, workspace) of a backward call that ran with phase 1; everything of phase 2 in one launch
int micf::offset_sample_finish_many(const SampleFinishCall* calls, int ncalls, hipStream_t s) {
  if (!calls || ncalls < 1) return MICF_EINVAL;
  FinishSets f;
  int total = 0;
  for (int c = 0; c < ncalls; ++c) {
    const SampleFinishCall& q = calls[c];
    if (!q.sets || q.n < 1 || q.n > 2 || q.B <= 0 || q.D <= 0 || q.H <= 0 || q.W <= 0) return MICF_EINVAL;
    const Geo g{q.B, q.D, q.H, q.W};
    const int64_t T = g.tokens();
    const int64_t per = micf_offset_sample_bwd_workspace(q.B, q.D, q.H, q.W);
    if (!q.workspace || q.workspace_floats < per * q.n || !aligned16(q.workspace)) return MICF_EINVAL;
    bool al = true;
    for (int i = 0; i < q.n; ++i) al = al && aligned16(q.sets[i].dxs) && aligned16(q.sets[i].xa) && aligned16(q.sets[i].dxa);
    BwdShape sh;
    bwd_shape(g, q.C, al, sh);                                   // (the launch shape of the phase-1 call)
    const int blocks = sh.blocks, wpb = sh.wpb;
    for (int i = 0; i < q.n; ++i) {
      if (total == kFinishMany) {
        hipLaunchKernelGGL(sample_finish_many_kernel, dim3(5 * kHid, total), dim3(256), 0, s, f, Geo{1, 1, 1, 1});
        if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
        total = 0;
      }
      f.partials[total] = q.workspace + i * partial_floats(T);
      f.dw1[total] = q.sets[i].dw1; f.dln_g[total] = q.sets[i].dln_g; f.dln_b[total] = q.sets[i].dln_b;
      f.nwaves[total] = blocks * wpb;
      ++total;
    }
  }
  hipLaunchKernelGGL(sample_finish_many_kernel, dim3(5 * kHid, total), dim3(256), 0, s, f, Geo{1, 1, 1, 1});
  MICF_RETURN_LAUNCH();
}
