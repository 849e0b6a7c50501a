// conv3_bwdx.hip -- data gradient of the 3x3x3 / pad 1 convolution with FEW output channels (N <= 16: conv_offset[0],
// MS.py:314, 354-356) as a direct convolution on the matrix cores.
//
//   dx[t, c] (=|+=) sum_{tap} sum_{n < 16} dy[t - off(tap), n] * w[n][c][tap]        c in [x1 | x2] (2C input channels)
//
// The reduction side is tiny (27 taps x 16 channels, the whole dy halo of a 128-token tile is 34 KB of LDS) and the output
// side is wide, so: rows (MFMA i) = input channels c, columns (j) = 16 tokens, k = the 16 dy channels of one tap.
//   * dy halo: staged once per workgroup, voxel stride 20 floats -> one conflict-free ds_read_b128 per (tap, token row) gives
//     the B operand of 4 k-steps (k-permutation r = 4*lr + s, as in gemm_dma.h).
//   * weights: pre-transposed to wt[tap][c][16 n] (a 27*Cin*16-float scratch, written by a tiny kernel of the same call), so
//     the A operand of 4 k-steps for 16 channels is ONE coalesced 16-byte load per lane straight from L2 into registers --
//     no LDS staging, no barrier inside the tap loop; the next tap's weights are prefetched under the current tap's MFMAs.
//   * every wave owns 2 token rows x NCT channel tiles (2*NCT accumulators); per tap: 2 LDS reads + NCT loads for 8*NCT MFMAs.
// The implicit-GEMM path (conv3.hip) needed 360 us for the 32^3 x 2 stage (15 TFLOP/s); see DESIGN.md section 3 for this kernel's numbers.
#include <cstdlib>
#include "common.h"
#include "conv3_layout.h"
#include "gemm_dma.h"

namespace micf {

constexpr int xKS = 20;      // LDS voxel stride (floats): 16 channels + 4 pad

struct BwdxArgs {
  const float* dy; int N;                       // channels-last [T, N], N <= 16, N % 4 == 0
  const float* wt;                              // [27][O][16]
  float* d1; float* d2; int oc1, oc2, acc1, acc2;
  int O;                                        // oc1 + oc2, multiple of 16
  int B, D, H, W, tiles_d, tiles_h, tiles_w, ngroups;
  const float* dyb; const float* wtb; float* d1b; float* d2b;      // second pointer set (blockIdx.z == 1)
};

__global__ void __launch_bounds__(256) conv3_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int Cin) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id < conv3_bwd_layout_items(Cin)) conv3_bwd_layout_write(w, wt, N, Cin, id);
}

// TW: tokens of a tile along w (16 or 8); a column tile is 16/TW h-rows x TW.  Tile = 2 (d) x 4*(16/TW) (h) x TW = 128 tokens.
template <int TW, int NCT, bool BF16>
__global__ void __launch_bounds__(256) conv3_bwdx_kernel(BwdxArgs a) {
  constexpr int CH = 16 / TW, TH = 4 * CH, TD = 2;
  constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HALO = HD * HH * HW;
  __shared__ __attribute__((aligned(16))) float Xs[HALO * xKS];
  if (blockIdx.z) { a.dy = a.dyb; a.wt = a.wtb; a.d1 = a.d1b; a.d2 = a.d2b; }      // the other modality's head
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lr = lane >> 4;
  int q = blockIdx.x;
  // XCD-contiguous tile order (gridDim.x % 8 == 0: the XCD of a workgroup is blockIdx.x % 8 whatever y / z): the halo planes
  // neighbouring tiles share are fetched into ONE XCD's L2
  if (gridDim.x >= 64 && (gridDim.x & 7) == 0) q = (q & 7) * (gridDim.x >> 3) + (q >> 3);
  const int tw = q % a.tiles_w; q /= a.tiles_w;
  const int th = q % a.tiles_h; q /= a.tiles_h;
  const int td = q % a.tiles_d; const int b = q / a.tiles_d;
  const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
  const int64_t DHW = (int64_t)a.D * a.H * a.W;
  const int ct0 = blockIdx.y * NCT;                       // first channel tile of this workgroup
  const int nct = min(NCT, a.O / 16 - ct0);

  // ---- dy halo -> LDS (all loads first, then the stores)
  {
    constexpr int NH = (HALO * 4 + 255) / 256;
    float4 hv4[NH];
#pragma unroll
    for (int it = 0; it < NH; ++it) {
      const int idx = tid + it * 256;
      hv4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < HALO * 4) {
        const int hv = idx >> 2, g = idx & 3;
        const int hw = hv % HW, hh = (hv / HW) % HH, hd = hv / (HW * HH);
        const int dd = d0 + hd - 1, yy = h0 + hh - 1, ww = w0 + hw - 1;
        if ((unsigned)dd < (unsigned)a.D && (unsigned)yy < (unsigned)a.H && (unsigned)ww < (unsigned)a.W && 4 * g < a.N)
          hv4[it] = *reinterpret_cast<const float4*>(a.dy + ((int64_t)b * DHW + ((int64_t)dd * a.H + yy) * a.W + ww) * a.N + 4 * g);
      }
    }
#pragma unroll
    for (int it = 0; it < NH; ++it) {
      const int idx = tid + it * 256;
      if (idx < HALO * 4) {
        if constexpr (BF16)      // bf16 mode: the dy halo is rounded once here, not once per tap at the fragment read
          *reinterpret_cast<uint2*>(&reinterpret_cast<uint16_t*>(Xs)[(idx >> 2) * xKS + 4 * (idx & 3)]) =
              make_uint2(pack_bf16(hv4[it].x, hv4[it].y), pack_bf16(hv4[it].z, hv4[it].w));
        else
          *reinterpret_cast<float4*>(&Xs[(idx >> 2) * xKS + 4 * (idx & 3)]) = hv4[it];
      }
    }
  }
  // ---- this wave's two column tiles: ct = 2*wave + tj -> (ld, h group); lane li -> (lh, lw) inside it
  int ld[2], lh[2], lw[2];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int ct = 2 * wave + tj;
    ld[tj] = ct / 4;
    lh[tj] = (ct % 4) * CH + li / TW;
    lw[tj] = li % TW;
  }
  f32x4 acc[NCT][2];
#pragma unroll
  for (int t = 0; t < NCT; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const float* wp = a.wt + ((int64_t)(ct0 * 16 + li)) * 16 + 4 * lr;      // + tap * O * 16 + t * 256
  const int64_t tap_stride = (int64_t)a.O * 16;
  if constexpr (BF16) {
    // bf16: one 16-byte load per (tap pair, channel tile) is the A fragment (bf16 part of the layout), the dy halo is bf16 in LDS
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const uint16_t* Xh = reinterpret_cast<const uint16_t*>(Xs);
    const u32x4* wq = reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(a.wt + conv3_bwd_layout_f32(a.O)) +
                                                     (((int64_t)(ct0 * 16 + li)) * 4 + lr) * 8);       // + pair * O * 4 + t * 64  (u32x4)
    const int64_t pair_stride = (int64_t)a.O * 4;
    const u32x4 zq = {0u, 0u, 0u, 0u};
    auto tap_pair_b = [&](int p, u32x4 (&bq)[2]) {          // the two column tiles' B fragments of tap pair p, from the bf16 halo
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        uint2 h[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int tap = 2 * p + e;
          const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
          const int zd = ld[tj] + 2 - kd, zh = lh[tj] + 2 - kh, zw = lw[tj] + 2 - kw;
          const uint2 v = *reinterpret_cast<const uint2*>(&Xh[((zd * HH + zh) * HW + zw) * xKS + 4 * lr]);
          h[e] = tap < 27 ? v : make_uint2(0u, 0u);
        }
        bq[tj] = u32x4{h[0].x, h[0].y, h[1].x, h[1].y};
      }
    };
    if constexpr (NCT >= 6) {
      // six channel tiles per workgroup (>= 128 workgroups: the big grids): 24 registers per stage of weight fragments -- one
      // pair of look-ahead in a ROLLED loop keeps the kernel at 116 registers = 4 workgroups per CU, and the other three waves of
      // a SIMD cover the load (a deeper unrolled ring: 172 registers, or 17-33 spilled at a 128 cap)
      u32x4 aq[NCT], anq[NCT];
#pragma unroll
      for (int t = 0; t < NCT; ++t) aq[t] = (t < nct) ? wq[t * 64] : zq;
      __syncthreads();
#pragma unroll 1
      for (int p = 0; p < 14; ++p) {
        if (p + 1 < 14) {
#pragma unroll
          for (int t = 0; t < NCT; ++t) anq[t] = (t < nct) ? wq[(p + 1) * pair_stride + t * 64] : zq;
        }
        u32x4 bq[2];
        tap_pair_b(p, bq);
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq[t]), __builtin_bit_cast(bf16x8, bq[0]), acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq[t]), __builtin_bit_cast(bf16x8, bq[1]), acc[t][1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NCT; ++t) aq[t] = anq[t];
      }
    } else {
      // few channel tiles per workgroup (the 8^3 / 4^3 / 16^3 launches, ~one workgroup per CU: nobody else hides a load): a ring of
      // xAhead + 1 tap pairs, fully unrolled -- the pair consumed by iteration p was requested xAhead iterations earlier.  (Round
      // 4's rolled loop ended every iteration in s_waitcnt vmcnt(0) -- its register copy needs the prefetched value -- so the
      // look-ahead covered 4 MFMAs of a ~600-cycle L2 latency, 14 times per workgroup: found in the ISA, round 5.)
      constexpr int xAhead = 3;
      u32x4 ring[xAhead + 1][NCT];
#pragma unroll
      for (int r = 0; r < xAhead; ++r)
#pragma unroll
        for (int t = 0; t < NCT; ++t) ring[r][t] = (t < nct) ? wq[r * pair_stride + t * 64] : zq;
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 14; ++p) {
        if (p + xAhead < 14) {
#pragma unroll
          for (int t = 0; t < NCT; ++t) ring[(p + xAhead) % (xAhead + 1)][t] = (t < nct) ? wq[(p + xAhead) * pair_stride + t * 64] : zq;
        }
        u32x4 bq[2];
        tap_pair_b(p, bq);
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
          const u32x4 aq = ring[p % (xAhead + 1)][t];
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq), __builtin_bit_cast(bf16x8, bq[0]), acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq), __builtin_bit_cast(bf16x8, bq[1]), acc[t][1], 0, 0, 0);
        }
      }
    }
  }
  float4 av[NCT], an[NCT], aprev[NCT], bprev[2];
#pragma unroll
  for (int t = 0; t < NCT; ++t) av[t] = (t < nct && !BF16) ? *reinterpret_cast<const float4*>(wp + t * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (!BF16) __syncthreads();
#pragma unroll 1
  for (int tap = 0; tap < (BF16 ? 0 : 27); ++tap) {
    if (tap + 1 < 27) {
#pragma unroll
      for (int t = 0; t < NCT; ++t)
        an[t] = (t < nct) ? *reinterpret_cast<const float4*>(wp + (tap + 1) * tap_stride + t * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    float4 bv[2];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int zd = ld[tj] + 2 - kd, zh = lh[tj] + 2 - kh, zw = lw[tj] + 2 - kw;      // source voxel = token - (k - 1), halo origin -1
      bv[tj] = *reinterpret_cast<const float4*>(&Xs[((zd * HH + zh) * HW + zw) * xKS + 4 * lr]);
    }
    if constexpr (!BF16) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float b0 = s == 0 ? bv[0].x : (s == 1 ? bv[0].y : (s == 2 ? bv[0].z : bv[0].w));
        const float b1 = s == 0 ? bv[1].x : (s == 1 ? bv[1].y : (s == 2 ? bv[1].z : bv[1].w));
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
          const float av_s = s == 0 ? av[t].x : (s == 1 ? av[t].y : (s == 2 ? av[t].z : av[t].w));
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_s, b0, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_s, b1, acc[t][1], 0, 0, 0);
        }
      }
    } else {
      // bf16: k = 32 = (16 dy channels) x (two taps); the odd tap of a pair issues the MFMAs, the last tap is padded with zeros
      if ((tap & 1) || tap == 26) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool pair = tap & 1;
        const bf16x8 bb0 = pair ? to_bf16x8(bprev[0], bv[0]) : to_bf16x8(bv[0], z);
        const bf16x8 bb1 = pair ? to_bf16x8(bprev[1], bv[1]) : to_bf16x8(bv[1], z);
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
          const bf16x8 ba = pair ? to_bf16x8(aprev[t], av[t]) : to_bf16x8(av[t], z);
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb0, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb1, acc[t][1], 0, 0, 0);
        }
      } else {
        bprev[0] = bv[0]; bprev[1] = bv[1];
#pragma unroll
        for (int t = 0; t < NCT; ++t) aprev[t] = av[t];
      }
    }
#pragma unroll
    for (int t = 0; t < NCT; ++t) av[t] = an[t];
  }
  // ---- epilogue: D row = channel 4*lr + v of tile t (float4 over v), column = token li.  Accumulating outputs: the old values of
  // a token's NCT channel tiles are requested together before the first is used (2 exposed round trips instead of 2 NCT dependent
  // load -> add -> store sequences: at the 8^3 / 4^3 stages, one workgroup per CU, that chain was 8 of the kernel's 13.6 us).
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int dd = d0 + ld[tj], yy = h0 + lh[tj], ww = w0 + lw[tj];
    if (dd >= a.D || yy >= a.H || ww >= a.W) continue;
    const int64_t tok = (int64_t)b * DHW + ((int64_t)dd * a.H + yy) * a.W + ww;
    auto where = [&](int t, int& accf) -> float* {
      const int c = (ct0 + t) * 16 + 4 * lr;
      if (t >= nct) return nullptr;
      if (c < a.oc1) { accf = a.acc1; return a.d1 ? a.d1 + tok * a.oc1 + c : nullptr; }
      accf = a.acc2;
      return a.d2 ? a.d2 + tok * a.oc2 + (c - a.oc1) : nullptr;
    };
    constexpr int EB = NCT >= 6 ? 2 : NCT;               // tiles per batch (six tiles: 2 x 4 registers of old values keep 4 workgroups per CU)
#pragma unroll
    for (int t0 = 0; t0 < NCT; t0 += EB) {
      float4 old[EB];
#pragma unroll
      for (int u = 0; u < EB; ++u) {
        int accf = 0;
        const float* p = where(t0 + u, accf);
        old[u] = (p && accf) ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < EB; ++u) {
        int accf = 0;
        float* p = where(t0 + u, accf);
        if (!p) continue;
        const f32x4 v = acc[t0 + u][tj];
        *reinterpret_cast<float4*>(p) = make_float4(v[0] + old[u].x, v[1] + old[u].y, v[2] + old[u].z, v[3] + old[u].w);
      }
    }
  }
}

template <int TW, bool BF16>
static hipError_t launch_bwdx(BwdxArgs& a, hipStream_t stream) {
  constexpr int CH = 16 / TW, TH = 4 * CH;
  a.tiles_d = (a.D + 1) / 2;
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = (a.W + TW - 1) / TW;
  const int64_t blocks = (int64_t)a.B * a.tiles_d * a.tiles_h * a.tiles_w;
  const int cts = a.O / 16;
  // many token tiles: 6 channel tiles per workgroup (the halo is staged once per 96 channels); few: 2, to spread over the CUs
  // (threshold swept in round 4: 128 workgroups)
  if (blocks * ((cts + 5) / 6) >= 128)
    hipLaunchKernelGGL((conv3_bwdx_kernel<TW, 6, BF16>), dim3((unsigned)blocks, (cts + 5) / 6, a.ngroups), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((conv3_bwdx_kernel<TW, 2, BF16>), dim3((unsigned)blocks, (cts + 1) / 2, a.ngroups), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// MICF_EUNSUPPORTED when the shape is outside what this kernel covers (caller falls back to the implicit GEMM).
int conv3_bwd_data_x(const float* dy, const float* w, float* wt, float* dx1, int c1, int acc1, float* dx2, int c2, int acc2, int B,
                     int D, int H, int W, int N, hipStream_t stream, int dtype, int prepared) {
  const Conv3BwdSet one{dy, w, wt, dx1, dx2};
  return conv3_bwd_data_x_groups(&one, 1, c1, acc1, c2, acc2, B, D, H, W, N, stream, dtype, prepared);
}

// 1 or 2 data gradients of the same shape in ONE launch (blockIdx.z)
int conv3_bwd_data_x_groups(const Conv3BwdSet* sets, int ng, int c1, int acc1, int c2, int acc2, int B, int D, int H, int W, int N,
                            hipStream_t stream, int dtype, int prepared) {
  if (!sets || ng < 1 || ng > 2) return MICF_EINVAL;
  const int O = c1 + c2;
  if (N > 16 || (N & 3) || (O & 15) || (c1 & 3) || (c2 & 3) || W < 4) return MICF_EUNSUPPORTED;
  for (int i = 0; i < ng; ++i)
    if (!aligned16(sets[i].dy) || !aligned16(sets[i].wt) || (sets[i].dx1 && !aligned16(sets[i].dx1)) || (sets[i].dx2 && !aligned16(sets[i].dx2)) ||
        (!sets[i].dx1) != (!sets[0].dx1) || (!sets[i].dx2) != (!sets[0].dx2))
      return MICF_EUNSUPPORTED;
  const float *dy = sets[0].dy;
  float *wt = sets[0].wt, *dx1 = sets[0].dx1, *dx2 = sets[0].dx2;
  const int64_t n = conv3_bwd_layout_items(O);
  if (!prepared) {
    for (int i = 0; i < ng; ++i) {
      hipLaunchKernelGGL(conv3_wt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sets[i].w, sets[i].wt, N, O);
      if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
    }
  }
  BwdxArgs a{};
  a.dyb = sets[ng - 1].dy; a.wtb = sets[ng - 1].wt; a.d1b = sets[ng - 1].dx1; a.d2b = sets[ng - 1].dx2; a.ngroups = ng;
  a.dy = dy; a.N = N; a.wt = wt; a.d1 = dx1; a.d2 = dx2; a.oc1 = c1; a.oc2 = c2; a.acc1 = acc1; a.acc2 = acc2; a.O = O;
  a.B = B; a.D = D; a.H = H; a.W = W;
  const hipError_t e = dtype == MICF_DTYPE_BF16 ? ((W >= 12) ? launch_bwdx<16, true>(a, stream) : launch_bwdx<8, true>(a, stream))
                                                 : ((W >= 12) ? launch_bwdx<16, false>(a, stream) : launch_bwdx<8, false>(a, stream));
  return e == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

}  // namespace micf
