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
    for (int c = lane; c < C; c += 64) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (ok[q]) acc += base[(int64_t)lin[q] * C + c] * wgt[q];
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
    for (int c = 4 * k; c < C; c += 64) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (ok[q]) {
          const float4 v = ld4(base + (int64_t)lin[q] * C + c);
          acc.x += v.x * wgt[q]; acc.y += v.y * wgt[q]; acc.z += v.z * wgt[q]; acc.w += v.w * wgt[q];
        }
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

template <bool SCATTER>
__device__ __forceinline__ void offset_sample_bwd_body(
    const float* __restrict__ dxs, const float* __restrict__ h, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
    const float* __restrict__ w1, const float* __restrict__ xa, const float* __restrict__ flow, float* __restrict__ dxa,
    float* __restrict__ dh, float* __restrict__ dln_g, float* __restrict__ dln_b, float* __restrict__ dw1, Geo g, int C, float eps,
    int tpw, CellLists cl, float* __restrict__ partials, int nwaves) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane & 15;
  const int64_t T = g.tokens();
  float acc_w[3] = {0.f, 0.f, 0.f}, acc_g = 0.f, acc_b = 0.f;       // per-lane (channel k) partials, lanes 0..15 flush
  for (int it = 0; it < tpw; ++it) {
    const int64_t t = ((int64_t)blockIdx.x * 4 + wave) * tpw + it;
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
    for (int c = lane; c < C; c += 64) {
      const float go = dxs[t * C + c];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (!ok[q]) continue;
        const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
        const int64_t a = boff + (int64_t)lin[q] * C + c;
        if (SCATTER) atomicAdd(dxa + a, wx[q] * wy[q] * wz[q] * go);
        const float val = xa[a] * go;
        gx += (dx ? val : -val) * wy[q] * wz[q];
        gy += (dy ? val : -val) * wx[q] * wz[q];
        gz += (dz ? val : -val) * wx[q] * wy[q];
      }
    }
    if (!SCATTER && lane == 0) {
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
      const int wg = blockIdx.x * 4 + wave;
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



template <bool SCATTER>
__device__ __forceinline__ void offset_sample_bwd4_body(
    const float* __restrict__ dxs, const float* __restrict__ h, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
    const float* __restrict__ w1, const float* __restrict__ xa, const float* __restrict__ flow, float* __restrict__ dxa,
    float* __restrict__ dh, float* __restrict__ dln_g, float* __restrict__ dln_b, float* __restrict__ dw1, Geo g, int C, float eps,
    int tpw, CellLists cl, float* __restrict__ partials, int nwaves) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane & 15, grp = lane >> 4;
  const int64_t T = g.tokens();
  float acc_w[3] = {0.f, 0.f, 0.f}, acc_g = 0.f, acc_b = 0.f;       // per-lane (group, channel k) partials
  for (int it = 0; it < tpw; ++it) {
    const int64_t t0 = (((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * tpw + it) * 4;
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
    for (int c = 4 * k; c < C; c += 64) {
      const float4 go = ld4(dxs + t * C + c);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (!ok[q]) continue;
        const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
        const int64_t a = boff + (int64_t)lin[q] * C + c;
        if (SCATTER && live) {
          const float wq = wx[q] * wy[q] * wz[q];
          atomicAdd(dxa + a, wq * go.x); atomicAdd(dxa + a + 1, wq * go.y);
          atomicAdd(dxa + a + 2, wq * go.z); atomicAdd(dxa + a + 3, wq * go.w);
        }
        const float4 xv = ld4(xa + a);
        const float val = xv.x * go.x + xv.y * go.y + xv.z * go.z + xv.w * go.w;
        gx += (dx ? val : -val) * wy[q] * wz[q];
        gy += (dy ? val : -val) * wx[q] * wz[q];
        gz += (dz ? val : -val) * wx[q] * wy[q];
      }
    }
    if (!SCATTER) {
      // a token whose cell list is full (normally none) scatters its d(xa) contributions atomically right here -- before the gather
      // launch reads / writes d(xa), so the finishing launch has no overflow pass left and only sums head-parameter partials
      int ovf = 0;
      if (live && k == 0) {
        const int cell = cell_of(tp, b, g.D, g.H, g.W);
        if (cell >= 0) {
          const int slot = atomicAdd(cl.count + cell, 1);
          if (slot < cl.cap) cl.list[(int64_t)cell * kCellCap + slot] = (int)t;
          else ovf = 1;
        }
      }
      ovf = __shfl(ovf, lane & 48, 64);
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
    if (!SCATTER && live && k < 8 && cl.w8) {           // lane k < 8: the weight of corner k (dz, dy, dx = bits of k)
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
      const int wg = blockIdx.x * (blockDim.x >> 6) + wave;
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
template <bool SCATTER>
__global__ void __launch_bounds__(256) offset_sample_bwd_kernel(const SampleBwdSets p, Geo g, int C, float eps, int tpw, int nwaves) {
  const SampleBwdSet& q = p.s[blockIdx.y];
  offset_sample_bwd_body<SCATTER>(q.dxs, q.h, q.ln_g, q.ln_b, q.w1, q.xa, q.flow, q.dxa, q.dh, q.dln_g, q.dln_b, q.dw1, g, C, eps, tpw, q.cl,
                                  q.partials, nwaves);
}
template <bool SCATTER>
__global__ void __launch_bounds__(256) offset_sample_bwd4_kernel(const SampleBwdSets p, Geo g, int C, float eps, int tpw, int nwaves) {
  const SampleBwdSet& q = p.s[blockIdx.y];
  offset_sample_bwd4_body<SCATTER>(q.dxs, q.h, q.ln_g, q.ln_b, q.w1, q.xa, q.flow, q.dxa, q.dh, q.dln_g, q.dln_b, q.dw1, g, C, eps, tpw, q.cl,
                                   q.partials, nwaves);
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
static bool use_cells(int64_t T) { return T >= 4096; }   // small grids keep the atomic scatter (launch-bound anyway)

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
  bool al = true;
  for (int i = 0; i < n; ++i) {
    const SampleBwdSet& q = sets[i];
    if (!q.dxs || !q.h || !q.ln_g || !q.ln_b || !q.w1 || !q.xa || !q.flow || !q.dxa || !q.dh || !q.dln_g || !q.dln_b || !q.dw1)
      return MICF_EINVAL;
    al = al && aligned16(q.dxs) && aligned16(q.xa) && aligned16(q.dxa);
  }
  // quad kernels (4 tokens per wave) when the channel rows allow 16-byte accesses
  static const int64_t quad_min = [] { const char* e = getenv("MICF_SAMPLE_QUAD_MIN"); return e ? (int64_t)atoll(e) : (int64_t)4096; }();
  const bool quad = T >= quad_min && (C % 4 == 0) && al;   // tiny grids: 1 token per wave
  const int tpw = quad ? quads_per_wave(T) : tok_per_wave(T);
  const int wpb = quad ? quad_waves_per_block(T) : 4;
  const int blocks = ceil_div(T, (quad ? 4 * wpb : 4) * tpw);
  const int nwaves = blocks * wpb;
  const CellLists none{nullptr, nullptr, nullptr, nullptr, 0, nullptr};
  const int64_t per = micf_offset_sample_bwd_workspace(B, D, H, W);
  const bool have_ws = workspace && workspace_floats >= per * n && aligned16(workspace);
  const bool cells = have_ws && use_cells(T) && quad && cell_count(B, D, H, W) * kCellCap < (1LL << 31);
  SampleBwdSets p;
  const int64_t nc = (cell_count(B, D, H, W) + 3) / 4 * 4;
  // workspace: [partials g0 | partials g1 | counters g0 | counters g1 | lists + overflow g0 | lists + overflow g1]
  // (the counters of both groups are contiguous: one memset)
  int* counters = have_ws ? reinterpret_cast<int*>(workspace + n * partial_floats(T)) : nullptr;
  int* lists = counters ? counters + n * (nc + 4) : nullptr;
  int cap = kCellCap;
  if (const char* env = getenv("MICF_CELL_CAP")) {           // test hook: force the overflow pass
    cap = atoi(env);
    cap = cap < 0 ? 0 : (cap > kCellCap ? kCellCap : cap);
  }
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
  const bool scatter = !cells;
  const dim3 grid(blocks, n), blk(64 * wpb);
  if (quad) {
    if (scatter) hipLaunchKernelGGL(offset_sample_bwd4_kernel<true>, grid, blk, 0, s, p, g, C, eps, tpw, nwaves);
    else hipLaunchKernelGGL(offset_sample_bwd4_kernel<false>, grid, blk, 0, s, p, g, C, eps, tpw, nwaves);
  } else {
    if (scatter) hipLaunchKernelGGL(offset_sample_bwd_kernel<true>, grid, blk, 0, s, p, g, C, eps, tpw, nwaves);
    else hipLaunchKernelGGL(offset_sample_bwd_kernel<false>, grid, blk, 0, s, p, g, C, eps, tpw, nwaves);
  }
  if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
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
    const bool quad = T >= 4096 && (q.C % 4 == 0) && al;         // (the launch shape of the phase-1 call: same formula)
    const int tpw = quad ? quads_per_wave(T) : tok_per_wave(T);
    const int wpb = quad ? quad_waves_per_block(T) : 4;
    const int blocks = ceil_div(T, (quad ? 4 * wpb : 4) * tpw);
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
