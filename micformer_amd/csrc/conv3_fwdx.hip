// conv3_fwdx.hip -- forward of the 3x3x3 / pad 1 convolution with FEW output channels (N <= 16: conv_offset[0], MS.py:314,
// 354-356), channels-last output, as a direct convolution on the matrix cores.
//
//   y[t, n] = bias[n] + sum_{tap} sum_c in[t + off(tap), c] * w[n][c][tap]              in = [x1 | x2]
//
// rows (MFMA i) = the 16 output channels, columns (j) = 16 tokens, k = 16 input channels of one (chunk, tap).
//   * input halo of a 128-token tile: staged in LDS one 16-channel chunk at a time (voxel stride 20 floats -> one conflict-free
//     ds_read_b128 per (tap, token row) is the B operand of 4 k-steps); the NEXT chunk is fetched into registers before the
//     MFMAs of the current one and committed after them, so the global latency hides under the matrix work;
//   * weights: pre-transposed to wt[chunk][tap][16 n][16 c] (scratch, written by a tiny kernel of the same call): the A operand
//     of 4 k-steps is ONE coalesced 16-byte load per lane from L2, prefetched 3 taps ahead in a register ring -- no LDS staging
//     of weights, no barrier inside the 27-tap loop;
//   * few token tiles (8^3 / 16^3 stages): the channel chunks are split over blockIdx.y and the partial sums added atomically
//     into the pre-zeroed output.
// The LDS-weights direct kernel this replaces (conv3_direct.hip) needed 129 us at the 32^3 x 2 stage (41 TFLOP/s).
#include <cstdlib>
#include "common.h"
#include "conv3_layout.h"
#include "gemm_dma.h"

namespace micf {

constexpr int fKS = 20;      // LDS voxel stride (floats): 16 channels + 4 pad
constexpr int fAhead = 3;    // taps of weight prefetch

struct FwdxArgs {
  const float* x1; const float* x2; int c1, c2;
  const float* wt;                              // [chunks][27][16][16]
  const float* bias; float* y; int N;           // channels-last [T, N]
  const float* x1b; const float* x2b; const float* wtb; const float* biasb; float* yb;   // second pointer set (blockIdx.z == 1)
  int B, D, H, W, tiles_d, tiles_h, tiles_w;
  int chunks, chunks_per_block;                 // gridDim.y = ceil(chunks / chunks_per_block); > 1 block per tile -> atomic output
};

__global__ void __launch_bounds__(256) conv3_wtf_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int Cin, int chunks) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id < conv3_fwd_layout_items(Cin)) conv3_fwd_layout_write(w, wt, N, Cin, id);
}

template <int TW, bool BF16>
__global__ void __launch_bounds__(256) conv3_fwdx_kernel(FwdxArgs a) {
  constexpr int CH = 16 / TW, TH = 4 * CH, TD = 2;
  constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HALO = HD * HH * HW;
  constexpr int NH = (HALO * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Xs[HALO * fKS];
  if (blockIdx.z) { a.x1 = a.x1b; a.x2 = a.x2b; a.wt = a.wtb; a.bias = a.biasb; a.y = a.yb; }   // the other modality's head
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
  const int Cin = a.c1 + a.c2;
  const int kc_begin = blockIdx.y * a.chunks_per_block;
  const int kc_end = min(a.chunks, kc_begin + a.chunks_per_block);

  // halo chunk -> registers (global loads only)
  auto fetch = [&](int kc, float4 (&hv)[NH]) {
#pragma unroll
    for (int it = 0; it < NH; ++it) {
      const int idx = tid + it * 256;
      hv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < HALO * 4) {
        const int v = idx >> 2, g = idx & 3;
        const int hw = v % HW, hh = (v / HW) % HH, hd = v / (HW * HH);
        const int dd = d0 + hd - 1, yy = h0 + hh - 1, ww = w0 + hw - 1;
        const int c = kc * 16 + 4 * g;
        if ((unsigned)dd < (unsigned)a.D && (unsigned)yy < (unsigned)a.H && (unsigned)ww < (unsigned)a.W && c < Cin) {
          const int64_t tok = (int64_t)b * DHW + ((int64_t)dd * a.H + yy) * a.W + ww;
          hv[it] = c < a.c1 ? *reinterpret_cast<const float4*>(a.x1 + tok * a.c1 + c)
                            : *reinterpret_cast<const float4*>(a.x2 + tok * a.c2 + (c - a.c1));
        }
      }
    }
  };
  // bf16 mode keeps the halo in LDS as bf16 (same index arithmetic, 2-byte elements): every element is rounded ONCE at the
  // commit instead of once per tap at the fragment read -- the conversions were 39 VALU instructions per MFMA
  uint16_t* Xh = reinterpret_cast<uint16_t*>(Xs);
  auto commit = [&](const float4 (&hv)[NH]) {
#pragma unroll
    for (int it = 0; it < NH; ++it) {
      const int idx = tid + it * 256;
      if (idx < HALO * 4) {
        if constexpr (BF16)
          *reinterpret_cast<uint2*>(&Xh[(idx >> 2) * fKS + 4 * (idx & 3)]) = make_uint2(pack_bf16(hv[it].x, hv[it].y), pack_bf16(hv[it].z, hv[it].w));
        else
          *reinterpret_cast<float4*>(&Xs[(idx >> 2) * fKS + 4 * (idx & 3)]) = hv[it];
      }
    }
  };

  // this wave's two column tiles: ct = 2*wave + tj -> (ld, h group); lane li -> (lh, lw) inside it
  int vbase[2];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int ct = 2 * wave + tj;
    const int ld = ct / 4, lh = (ct % 4) * CH + li / TW, lw = li % TW;
    vbase[tj] = ((ld * HH + lh) * HW + lw) * fKS + 4 * lr;            // halo voxel of tap (0,0,0) for this token, k slot 4*lr
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float4 hv[NH];
  fetch(kc_begin, hv);
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    __syncthreads();                                                    // previous chunk fully consumed
    commit(hv);
    __syncthreads();
    if (kc + 1 < kc_end) fetch(kc + 1, hv);                             // lands while the MFMAs below run
    const float* wp = a.wt + ((int64_t)kc * 27 * 16 + li) * 16 + 4 * lr;  // + tap * 256
    if constexpr (BF16) {
      // bf16: weights from the bf16 part of the layout (one 16-byte load = the A fragment of a tap pair, no conversion), halo as
      // bf16 from LDS; k = 32 = (16 channels of this chunk) x (two taps), the unpaired 27th tap is zero-padded on both sides
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4* wq = reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(a.wt + conv3_fwd_layout_f32(Cin)) +
                                                       (((int64_t)kc * 14 * 16 + li) * 4 + lr) * 8);     // + pair * 64 (u32x4 units)
      u32x4 ringq[fAhead + 1];
#pragma unroll
      for (int t = 0; t < fAhead; ++t) ringq[t] = wq[t * 64];
#pragma unroll
      for (int p = 0; p < 14; ++p) {
        if (p + fAhead < 14) ringq[(p + fAhead) % (fAhead + 1)] = wq[(p + fAhead) * 64];
        const u32x4 aq = ringq[p % (fAhead + 1)];
        uint2 h[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int tap = 2 * p + e;
          if (tap < 27) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int off = ((kd * HH + kh) * HW + kw) * fKS;
            h[e][0] = *reinterpret_cast<const uint2*>(&Xh[vbase[0] + off]);
            h[e][1] = *reinterpret_cast<const uint2*>(&Xh[vbase[1] + off]);
          } else {
            h[e][0] = make_uint2(0u, 0u); h[e][1] = h[e][0];
          }
        }
        const u32x4 q0 = {h[0][0].x, h[0][0].y, h[1][0].x, h[1][0].y}, q1 = {h[0][1].x, h[0][1].y, h[1][1].x, h[1][1].y};
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq), __builtin_bit_cast(bf16x8, q0), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq), __builtin_bit_cast(bf16x8, q1), acc[1], 0, 0, 0);
      }
      continue;
    }
    float4 ring[fAhead + 1];
    float4 aprev = make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 hprev0 = make_uint2(0u, 0u), hprev1 = hprev0;
#pragma unroll
    for (int t = 0; t < fAhead; ++t) ring[t] = *reinterpret_cast<const float4*>(wp + t * 256);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      if (tap + fAhead < 27) ring[(tap + fAhead) % (fAhead + 1)] = *reinterpret_cast<const float4*>(wp + (tap + fAhead) * 256);
      const float4 av = ring[tap % (fAhead + 1)];
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int off = ((kd * HH + kh) * HW + kw) * fKS;                 // source voxel = token + (k - 1), halo origin -1
      if constexpr (!BF16) {
        const float4 b0 = *reinterpret_cast<const float4*>(&Xs[vbase[0] + off]);
        const float4 b1 = *reinterpret_cast<const float4*>(&Xs[vbase[1] + off]);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1.w, acc[1], 0, 0, 0);
      } else {
        // bf16: k = 32 = (16 channels of this chunk) x (two taps); the odd tap of a pair issues the MFMA, the last tap is padded
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const uint2 h0 = *reinterpret_cast<const uint2*>(&Xh[vbase[0] + off]);
        const uint2 h1 = *reinterpret_cast<const uint2*>(&Xh[vbase[1] + off]);
        if ((tap & 1) || tap == 26) {
          const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          const bool pair = tap & 1;
          const bf16x8 ba = pair ? to_bf16x8(aprev, av) : to_bf16x8(av, z);
          const u32x4 q0 = pair ? u32x4{hprev0.x, hprev0.y, h0.x, h0.y} : u32x4{h0.x, h0.y, 0u, 0u};
          const u32x4 q1 = pair ? u32x4{hprev1.x, hprev1.y, h1.x, h1.y} : u32x4{h1.x, h1.y, 0u, 0u};
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, __builtin_bit_cast(bf16x8, q0), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, __builtin_bit_cast(bf16x8, q1), acc[1], 0, 0, 0);
        } else {
          aprev = av; hprev0 = h0; hprev1 = h1;
        }
      }
    }
  }
  // epilogue: D row = output channel 4*lr + v (float4 over v), column = token li of column tile tj
  const bool atomic_out = gridDim.y > 1;
  const int n0 = 4 * lr;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int ct = 2 * wave + tj;
    const int dd = d0 + ct / 4, yy = h0 + (ct % 4) * CH + li / TW, ww = w0 + li % TW;
    if (dd >= a.D || yy >= a.H || ww >= a.W || n0 >= a.N) continue;
    float* p = a.y + ((int64_t)b * DHW + ((int64_t)dd * a.H + yy) * a.W + ww) * a.N + n0;
    f32x4 v = acc[tj];
    if (a.bias && blockIdx.y == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n0 + e < a.N) v[e] += a.bias[n0 + e];
    }
    if (atomic_out) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n0 + e < a.N) atomicAdd(p + e, v[e]);
    } else if (n0 + 3 < a.N && (a.N & 3) == 0) {
      *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n0 + e < a.N) p[e] = v[e];
    }
  }
}

int64_t conv3_fwdx_workspace(int N, int c1, int c2) {
  if (N <= 0 || N > 16 || c1 + c2 <= 0) return 0;
  return conv3_fwd_layout_floats(c1 + c2);
}

// MICF_EUNSUPPORTED when the shape is outside what this kernel covers (caller falls back).
int conv3_fwd_x(const float* x1, int c1, const float* x2, int c2, const float* w, const float* bias, float* y, float* wt, int B, int D,
                int H, int W, int N, hipStream_t stream, int dtype, int prepared) {
  const Conv3FwdSet one{x1, x2, w, bias, y, wt};
  return conv3_fwd_x_groups(&one, 1, c1, c2, B, D, H, W, N, stream, dtype, prepared, 0);
}

bool conv3_fwd_x_splits(int B, int D, int H, int W, int c1, int c2) {      // does the direct kernel accumulate atomically here?
  const int chunks = (c1 + c2 + 15) / 16;
  const int tw_ = W >= 12 ? 16 : 8, th_ = 4 * (16 / tw_);
  const int64_t blocks = (int64_t)B * ((D + 1) / 2) * ((H + th_ - 1) / th_) * ((W + tw_ - 1) / tw_);
  return blocks < 256 && chunks > 1;
}

// 1 or 2 convolutions of the same shape in ONE launch (blockIdx.z).  y_zeroed: the caller already cleared the outputs (only
// matters where the channel chunks are split over workgroups and accumulated atomically).
int conv3_fwd_x_groups(const Conv3FwdSet* sets, int n, int c1, int c2, int B, int D, int H, int W, int N, hipStream_t stream, int dtype,
                       int prepared, int y_zeroed) {
  if (!sets || n < 1 || n > 2) return MICF_EINVAL;
  if (N > 16 || (c1 & 3) || (c2 & 3) || W < 4) return MICF_EUNSUPPORTED;
  for (int i = 0; i < n; ++i)
    if (!aligned16(sets[i].x1) || (sets[i].x2 && !aligned16(sets[i].x2)) || !aligned16(sets[i].y) || !aligned16(sets[i].wt)) return MICF_EUNSUPPORTED;
  const float *x1 = sets[0].x1, *x2 = sets[0].x2, *w = sets[0].w, *bias = sets[0].bias;
  float *y = sets[0].y, *wt = sets[0].wt;
  const int Cin = c1 + c2, chunks = (Cin + 15) / 16;
  const int64_t nw = conv3_fwd_layout_items(Cin);
  if (!prepared) {      // (prepared: wt already holds this layout, written once per step by micf_conv3_weight_prep_grouped)
    for (int i = 0; i < n; ++i) {
      hipLaunchKernelGGL(conv3_wtf_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, sets[i].w, sets[i].wt, N, Cin, chunks);
      if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
    }
  }
  FwdxArgs a{};
  a.x1 = x1; a.x2 = x2 ? x2 : x1; a.c1 = c1; a.c2 = c2; a.wt = wt; a.bias = bias; a.y = y; a.N = N;
  a.B = B; a.D = D; a.H = H; a.W = W; a.chunks = chunks;
  const Conv3FwdSet& sb = sets[n - 1];
  a.x1b = sb.x1; a.x2b = sb.x2 ? sb.x2 : sb.x1; a.wtb = sb.wt; a.biasb = sb.bias; a.yb = sb.y;
  const int tw_ = W >= 12 ? 16 : 8, th_ = 4 * (16 / tw_);
  a.tiles_d = (D + 1) / 2; a.tiles_h = (H + th_ - 1) / th_; a.tiles_w = (W + tw_ - 1) / tw_;
  const int64_t blocks = (int64_t)B * a.tiles_d * a.tiles_h * a.tiles_w;
  // enough token tiles: one workgroup walks all channel chunks; otherwise spread the chunks (atomic accumulation into y)
  int ysplit = 1;
  constexpr int split_target = 128;       // workgroups aimed at on small grids (swept in round 4: 16 .. 1024)
  if (blocks < 256) { ysplit = (int)((split_target + blocks - 1) / blocks); if (ysplit > chunks) ysplit = chunks; if (ysplit < 1) ysplit = 1; }
  a.chunks_per_block = (chunks + ysplit - 1) / ysplit;
  ysplit = (chunks + a.chunks_per_block - 1) / a.chunks_per_block;
  if (ysplit > 1 && !y_zeroed)
    for (int i = 0; i < n; ++i)
      if (hipMemsetAsync(sets[i].y, 0, sizeof(float) * (size_t)B * D * H * W * N, stream) != hipSuccess) return MICF_ELAUNCH;
  const dim3 grid((unsigned)blocks, ysplit, n);
  if (dtype == MICF_DTYPE_BF16) {
    if (tw_ == 16) hipLaunchKernelGGL((conv3_fwdx_kernel<16, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv3_fwdx_kernel<8, true>), grid, dim3(256), 0, stream, a);
  } else {
    if (tw_ == 16) hipLaunchKernelGGL((conv3_fwdx_kernel<16, false>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv3_fwdx_kernel<8, false>), grid, dim3(256), 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

}  // namespace micf
