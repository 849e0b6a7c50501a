// conv3_wgrad.hip -- weight gradient of the 3x3x3 / pad 1 convolution as a DIRECT, LDS-tiled kernel.
//   dW[n][c][tap] = sum_t dy[t, n] * in[nbr(t, tap), c]          (conv_offset[0]: MS.py:314; out_conv: MS.py:1046)
// The implicit-GEMM form gathers every input element 27 times through L1 and feeds a 16-wide MFMA with N = 8/16 columns;
// at fp32 the matrix pipe has the same peak as the VALU (157 TFLOP/s), so this kernel instead stages a 4x8x8 token tile
// of dy and its 6x10x10 halo of a 24-channel input chunk in LDS ONCE and lets every thread own a (tap, 4-channel) slice of
// the gradient: 4c x N accumulators in registers, one 16-byte LDS read of the input + N/4 broadcast reads of dy per token
// for 4*N FMAs.  A workgroup walks many token tiles before flushing its partial with atomicAdd.
#include <mutex>

#include "common.h"

namespace micf {

constexpr int kCC = 24;                 // input channels per workgroup chunk (6 float4 groups)
constexpr int kTD = 4, kTH = 8, kTW = 8;
constexpr int kTileTok = kTD * kTH * kTW;                   // 256
constexpr int kHD = kTD + 2, kHH = kTH + 2, kHW = kTW + 2;   // 6 x 10 x 10
constexpr int kHaloTok = kHD * kHH * kHW;                   // 600

template <int N>
__global__ void __launch_bounds__(256) conv3_wgrad_kernel(const float* __restrict__ dy, int dy_layout,
                                                          const float* __restrict__ x1, int c1,
                                                          const float* __restrict__ x2, int c2, float* __restrict__ dw,
                                                          float* __restrict__ dbias, int B, int D, int H, int W,
                                                          int tiles_d, int tiles_h, int tiles_w, int tiles_per_block) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sx = smem;                                  // [kHaloTok][kCC]
  float* sdy = smem + kHaloTok * kCC;                // [kTileTok][N]
  const int Cin = c1 + c2;
  const int chunk = blockIdx.x;                      // channel chunk
  const int cbase = chunk * kCC;
  const int cvalid = (Cin - cbase < kCC) ? Cin - cbase : kCC;
  const int tid = threadIdx.x;
  // gradient slice of this thread: tap (0..26) x channel group cg (0..5)
  const int tap = tid / 6, cg = tid % 6;
  const bool owner = tid < 27 * 6;
  const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
  float acc[N][4];
#pragma unroll
  for (int n = 0; n < N; ++n) { acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f; }
  float bsum = 0.f;                                  // dbias partial: thread tid < N sums dy[:, tid]

  const int64_t DHW = (int64_t)D * H * W;
  const int ntiles = B * tiles_d * tiles_h * tiles_w;
  const int t_begin = blockIdx.y * tiles_per_block;
  const int t_end = (t_begin + tiles_per_block < ntiles) ? t_begin + tiles_per_block : ntiles;
  for (int tile = t_begin; tile < t_end; ++tile) {
    int q = tile;
    const int tw = q % tiles_w; q /= tiles_w;
    const int th = q % tiles_h; q /= tiles_h;
    const int td = q % tiles_d; const int b = q / tiles_d;
    const int d0 = td * kTD, h0 = th * kTH, w0 = tw * kTW;
    __syncthreads();                                 // previous tile fully consumed
    // ---- stage the halo of the input chunk: 600 voxels x 6 float4
#pragma unroll 5
    for (int idx = tid; idx < kHaloTok * 6; idx += 256) {
      const int v = idx / 6, g = idx % 6;
      const int hw = v % kHW, hh = (v / kHW) % kHH, hd = v / (kHW * kHH);
      const int dd = d0 + hd - 1, yy = h0 + hh - 1, ww = w0 + hw - 1;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      const int c = cbase + 4 * g;
      if ((unsigned)dd < (unsigned)D && (unsigned)yy < (unsigned)H && (unsigned)ww < (unsigned)W && 4 * g < cvalid) {
        const int64_t tok = (int64_t)b * DHW + ((int64_t)dd * H + yy) * W + ww;
        if (4 * g + 3 < cvalid && c + 3 < Cin && ((c < c1) == (c + 3 < c1))) {
          val = c < c1 ? *reinterpret_cast<const float4*>(x1 + tok * c1 + c)
                       : *reinterpret_cast<const float4*>(x2 + tok * c2 + (c - c1));
        } else {
          float t4[4] = {0.f, 0.f, 0.f, 0.f};
          for (int e = 0; e < 4; ++e) {
            const int ce = c + e;
            if (4 * g + e < cvalid) t4[e] = ce < c1 ? x1[tok * c1 + ce] : x2[tok * c2 + (ce - c1)];
          }
          val = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
      }
      *reinterpret_cast<float4*>(sx + v * kCC + 4 * g) = val;
    }
    // ---- stage dy of the 256 tile tokens (zero outside the volume)
#pragma unroll
    for (int idx = tid; idx < kTileTok * N; idx += 256) {
      int tok, n;
      if (dy_layout == 0) { tok = idx / N; n = idx % N; } else { n = idx / kTileTok; tok = idx % kTileTok; }
      const int lw = tok % kTW, lh = (tok / kTW) % kTH, ld = tok / (kTW * kTH);
      const int dd = d0 + ld, yy = h0 + lh, ww = w0 + lw;
      float val = 0.f;
      if (dd < D && yy < H && ww < W) {
        const int64_t vox = ((int64_t)dd * H + yy) * W + ww;
        val = dy_layout == 0 ? dy[((int64_t)b * DHW + vox) * N + n] : dy[((int64_t)b * N + n) * DHW + vox];
      }
      sdy[tok * N + n] = val;
    }
    __syncthreads();
    if (dbias && chunk == 0 && tid < N) {
      float s = 0.f;
      for (int t = 0; t < kTileTok; ++t) s += sdy[t * N + tid];
      bsum += s;
    }
    if (owner) {
      for (int ld = 0; ld < kTD; ++ld)
        for (int lh = 0; lh < kTH; ++lh) {
          const float* xrow = sx + (((ld + kd) * kHH + (lh + kh)) * kHW + kw) * kCC + 4 * cg;
          const float* drow = sdy + ((ld * kTH + lh) * kTW) * N;
#pragma unroll
          for (int lw = 0; lw < kTW; ++lw) {
            const float4 xv = *reinterpret_cast<const float4*>(xrow + lw * kCC);
#pragma unroll
            for (int n4 = 0; n4 < N; n4 += 4) {
              const float4 dv = *reinterpret_cast<const float4*>(drow + lw * N + n4);
              const float dvv[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc[n4 + e][0] += dvv[e] * xv.x; acc[n4 + e][1] += dvv[e] * xv.y;
                acc[n4 + e][2] += dvv[e] * xv.z; acc[n4 + e][3] += dvv[e] * xv.w;
              }
            }
          }
        }
    }
  }
  if (owner) {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cbase + 4 * cg + e;
        if (4 * cg + e < cvalid) atomicAdd(dw + ((int64_t)n * Cin + c) * 27 + tap, acc[n][e]);
      }
  }
  if (dbias && chunk == 0 && tid < N) atomicAdd(dbias + tid, bsum);
}

// returns MICF_EUNSUPPORTED when the shape is not handled (caller falls back to the implicit-GEMM path)
int conv3_wgrad_direct(const float* dy, int dy_layout, const float* x1, int c1, const float* x2, int c2, float* dw,
                       float* dbias, int B, int D, int H, int W, int N, hipStream_t stream) {
  if (N != 8 && N != 16) return MICF_EUNSUPPORTED;
  if (!aligned16(x1) || (x2 && !aligned16(x2)) || c1 % 4 || c2 % 4) return MICF_EUNSUPPORTED;
  const int Cin = c1 + c2;
  const int chunks = (Cin + kCC - 1) / kCC;
  const int tiles_d = (D + kTD - 1) / kTD, tiles_h = (H + kTH - 1) / kTH, tiles_w = (W + kTW - 1) / kTW;
  const int ntiles = B * tiles_d * tiles_h * tiles_w;
  // aim for ~2 workgroups per CU; every workgroup flushes 27*24*N atomics, so it should see many tiles when there are many
  int groups = (512 + chunks - 1) / chunks;
  if (groups > ntiles) groups = ntiles;
  if (groups < 1) groups = 1;
  const int tpb = (ntiles + groups - 1) / groups;
  groups = (ntiles + tpb - 1) / tpb;
  const size_t smem = sizeof(float) * (kHaloTok * kCC + kTileTok * N);
  static std::once_flag attr_once;       // > 64 KiB of dynamic LDS needs the opt-in once per process (not a stream operation)
  std::call_once(attr_once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgrad_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgrad_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  });
  dim3 grid(chunks, groups);
  if (N == 8) {
    hipLaunchKernelGGL(conv3_wgrad_kernel<8>, grid, dim3(256), smem, stream, dy, dy_layout, x1, c1, x2 ? x2 : x1, c2,
                       dw, dbias, B, D, H, W, tiles_d, tiles_h, tiles_w, tpb);
  } else {
    hipLaunchKernelGGL(conv3_wgrad_kernel<16>, grid, dim3(256), smem, stream, dy, dy_layout, x1, c1, x2 ? x2 : x1,
                       c2, dw, dbias, B, D, H, W, tiles_d, tiles_h, tiles_w, tpb);
  }
  return hipGetLastError() == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

}  // namespace micf
