// conv3_direct.hip -- forward of the 3x3x3 / pad 1 convolution as a DIRECT convolution on the matrix cores, with the input
// halo staged in LDS once per token tile (27x reuse) instead of being gathered 27 times through L1 by the implicit-GEMM
// path.  Scratch-free fallback of conv3_fwdx.hip (used when the caller passes no workspace) and the reason the template
// still carries a BWD parameter: the data gradient now lives in conv3_bwdx.hip.
// (conv_offset[0] on cat[LN(x), xa]: MS.py:314, 354-356)
//
//   out[t, o] = sum_{tap, k} Wt(o, k, tap) * in[t (+/-) off(tap), k]
//     forward : k = input channel c (16-channel chunks of [x1 | x2]), o = output channel n (<= 16), Wt = w[o][k][tap]
//     backward: k = n (dy, <= 16 channels, staged once), o = input channel c (16 at a time), Wt = w[k][o][tap], taps flipped
//
// Workgroup = 4 waves on a 2 x 4 x 16 token tile (8 w-rows of 16 tokens); MFMA 16x16x4 fp32 with i = o (16), j = the 16
// tokens of a w-row, k = 4 channels.  LDS: halo 4 x 6 x 18 voxels x 16 channels (row stride 20 floats) + the weight chunk
// [27][16 k][16 o]: 62 KB, two workgroups per CU.  Every MFMA costs one conflict-free A read and one <= 2-way B read.
#include "common.h"

namespace micf {

constexpr int dTD = 2, dTH = 4, dTW = 16;
constexpr int dHD = dTD + 2, dHH = dTH + 2, dHW = dTW + 2;
constexpr int dHalo = dHD * dHH * dHW;      // 432 voxels
constexpr int dKC = 16;                      // channels per staged chunk
constexpr int dKS = 20;                      // LDS voxel stride (floats)

struct DirectArgs {
  // input side
  const float* in1; const float* in2; int ic1, ic2;     // channels-last sources (forward: x1|x2; backward: dy when layout 0)
  int in_layout;                                        // backward only: 0 channels-last dy [T, K], 1 NCDHW dy [B, K, V]
  int K;                                                // total reduction channels (forward: Cin; backward: N)
  // weights [N][Cin][27]
  const float* w; int Cin;
  // output side
  int O;                                                // total output channels (forward: N; backward: Cin)
  const float* bias; float* y; int y_layout;            // forward
  float* d1; float* d2; int oc1, oc2, acc1, acc2;       // backward: dx split at oc1
  int B, D, H, W, tiles_d, tiles_h, tiles_w;
  int ysplit;                                           // > 1: blockIdx.y splits output chunks (backward) / k chunks (forward, atomic)
};

template <bool BWD>
__global__ void __launch_bounds__(256, 2) conv3_direct_kernel(DirectArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[dHalo * dKS];
  __shared__ __attribute__((aligned(16))) float Ws[27 * dKC * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lr = lane >> 4;
  int q = blockIdx.x;
  const int tw = q % a.tiles_w; q /= a.tiles_w;
  const int th = q % a.tiles_h; q /= a.tiles_h;
  const int td = q % a.tiles_d; const int b = q / a.tiles_d;
  const int d0 = td * dTD, h0 = th * dTH, w0 = tw * dTW;
  const int64_t DHW = (int64_t)a.D * a.H * a.W;

  const int n_ochunk = (a.O + 15) / 16;
  const int n_kchunk = (a.K + dKC - 1) / dKC;
  // chunk ranges of this workgroup
  int oc_begin = 0, oc_end = n_ochunk, kc_begin = 0, kc_end = n_kchunk;
  if (a.ysplit > 1) {
    if (BWD) { const int per = (n_ochunk + a.ysplit - 1) / a.ysplit; oc_begin = blockIdx.y * per; oc_end = min(n_ochunk, oc_begin + per); }
    else { const int per = (n_kchunk + a.ysplit - 1) / a.ysplit; kc_begin = blockIdx.y * per; kc_end = min(n_kchunk, kc_begin + per); }
  }
  const bool atomic_out = !BWD && a.ysplit > 1;
  for (int oc = oc_begin; oc < oc_end; ++oc) {
    const int o0 = oc * 16;
    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = kc_begin; kc < kc_end; ++kc) {
      const int k0 = kc * dKC;
      __syncthreads();                               // previous chunk fully consumed
      // ---------------- stage the input halo chunk (skip when it is already resident: backward has ONE k chunk)
      if (!(BWD && oc > oc_begin && n_kchunk == 1)) {
        if (!BWD || a.in_layout == 0) {
          const int K1 = BWD ? a.K : a.ic1;          // split between the two channels-last sources
          constexpr int NH = (dHalo * 4 + 255) / 256;     // 7 float4 per thread, loads first, then stores
          float4 hv4[NH];
#pragma unroll
          for (int it = 0; it < NH; ++it) {
            const int idx = tid + it * 256;
            hv4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < dHalo * 4) {
              const int hv = idx >> 2, g = idx & 3;
              const int hw = hv % dHW, hh = (hv / dHW) % dHH, hd = hv / (dHW * dHH);
              const int dd = d0 + hd - 1, yy = h0 + hh - 1, ww = w0 + hw - 1;
              const int c = k0 + 4 * g;
              if ((unsigned)dd < (unsigned)a.D && (unsigned)yy < (unsigned)a.H && (unsigned)ww < (unsigned)a.W && c < a.K) {
                const int64_t tok = (int64_t)b * DHW + ((int64_t)dd * a.H + yy) * a.W + ww;
                if (c + 3 < a.K && ((c < K1) == (c + 3 < K1))) {
                  hv4[it] = c < K1 ? *reinterpret_cast<const float4*>(a.in1 + tok * K1 + c)
                                   : *reinterpret_cast<const float4*>(a.in2 + tok * a.ic2 + (c - K1));
                } else {
                  float t4[4] = {0.f, 0.f, 0.f, 0.f};
                  for (int e = 0; e < 4; ++e) {
                    const int ce = c + e;
                    if (ce < a.K) t4[e] = ce < K1 ? a.in1[tok * K1 + ce] : a.in2[tok * a.ic2 + (ce - K1)];
                  }
                  hv4[it] = make_float4(t4[0], t4[1], t4[2], t4[3]);
                }
              }
            }
          }
#pragma unroll
          for (int it = 0; it < NH; ++it) {
            const int idx = tid + it * 256;
            if (idx < dHalo * 4) *reinterpret_cast<float4*>(&Xs[(idx >> 2) * dKS + 4 * (idx & 3)]) = hv4[it];
          }
        } else {                                     // NCDHW dy: voxels contiguous per channel
          constexpr int NP = dHalo * dKC / 256;           // 27 scalars per thread
          float pv[NP];
#pragma unroll
          for (int it = 0; it < NP; ++it) {
            const int idx = tid + it * 256;
            const int n = idx / dHalo, hv = idx % dHalo;
            const int hw = hv % dHW, hh = (hv / dHW) % dHH, hd = hv / (dHW * dHH);
            const int dd = d0 + hd - 1, yy = h0 + hh - 1, ww = w0 + hw - 1;
            pv[it] = 0.f;
            if ((unsigned)dd < (unsigned)a.D && (unsigned)yy < (unsigned)a.H && (unsigned)ww < (unsigned)a.W && k0 + n < a.K)
              pv[it] = a.in1[((int64_t)b * a.K + k0 + n) * DHW + ((int64_t)dd * a.H + yy) * a.W + ww];
          }
#pragma unroll
          for (int it = 0; it < NP; ++it) {
            const int idx = tid + it * 256;
            Xs[(idx % dHalo) * dKS + idx / dHalo] = pv[it];
          }
        }
      }
      // ---------------- stage the weight chunk as Ws[k][o][27 taps]: the global side walks w[n][c][tap] contiguously
      // (432-float runs), the LDS side is conflict-free for these stores AND for the A-fragment reads below
      // (27 is odd: lanes (o = li, k-slot = lr) land on 32 distinct banks).
      {
        // all 27 loads of a thread are issued before the first LDS store (one memory round trip, not 27)
        float wv[27];
#pragma unroll
        for (int it = 0; it < 27; ++it) {
          const int idx = tid + it * 256;
          const int tap = idx % 27, p2 = idx / 27;
          const int o = BWD ? (p2 & 15) : (p2 >> 4), kk = BWD ? (p2 >> 4) : (p2 & 15);
          wv[it] = 0.f;
          if (o0 + o < a.O && k0 + kk < a.K) {
            const int n = BWD ? k0 + kk : o0 + o, c = BWD ? o0 + o : k0 + kk;
            wv[it] = a.w[((int64_t)n * a.Cin + c) * 27 + tap];
          }
        }
#pragma unroll
        for (int it = 0; it < 27; ++it) {
          const int idx = tid + it * 256;
          const int tap = idx % 27, p2 = idx / 27;
          const int o = BWD ? (p2 & 15) : (p2 >> 4), kk = BWD ? (p2 >> 4) : (p2 & 15);
          Ws[(kk * 16 + o) * 27 + tap] = wv[it];
        }
      }
      __syncthreads();
      // ---------------- 27 taps x 4 k-steps x 2 token rows of MFMA
      const int kmax = (a.K - k0 < dKC) ? a.K - k0 : dKC;
#pragma unroll 1
      for (int tap = 0; tap < 27; ++tap) {
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        int hbase[2];
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const int row = wave * 2 + tj, ld = row / dTH, lh = row % dTH;
          const int zd = BWD ? ld + 2 - kd : ld + kd, zh = BWD ? lh + 2 - kh : lh + kh, zw = BWD ? li + 2 - kw : li + kw;
          hbase[tj] = ((zd * dHH + zh) * dHW + zw) * dKS + lr;
        }
        const float* wp = Ws + (lr * 16 + li) * 27 + tap;
#pragma unroll
        for (int ks = 0; ks < dKC / 4; ++ks) {
          if (4 * ks < kmax) {
            const float av = wp[ks * 64 * 27];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Xs[hbase[0] + 4 * ks], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Xs[hbase[1] + 4 * ks], acc[1], 0, 0, 0);
          }
        }
      }
    }
    // ---------------- epilogue for this output chunk: D row i = o (4*lr + v), column j = token w0 + li
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int row = wave * 2 + tj, ld = row / dTH, lh = row % dTH;
      const int dd = d0 + ld, yy = h0 + lh, ww = w0 + li;
      if (dd >= a.D || yy >= a.H || ww >= a.W) continue;
      const int64_t vox = ((int64_t)dd * a.H + yy) * a.W + ww;
      const int64_t tok = (int64_t)b * DHW + vox;
      const int o = o0 + 4 * lr;
      if (o >= a.O) continue;
      f32x4 v = acc[tj];
      if (!BWD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (a.bias && o + e < a.O && (!atomic_out || blockIdx.y == 0)) v[e] += a.bias[o + e];
        if (atomic_out) {                            // channels-last only; y pre-zeroed by the launcher
          float* p = a.y + tok * a.O + o;
          for (int e = 0; e < 4; ++e) if (o + e < a.O) atomicAdd(p + e, v[e]);
        } else if (a.y_layout == 0) {
          float* p = a.y + tok * a.O + o;
          if (o + 3 < a.O && (a.O & 3) == 0) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
          else for (int e = 0; e < 4; ++e) if (o + e < a.O) p[e] = v[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (o + e < a.O) a.y[((int64_t)b * a.O + o + e) * DHW + vox] = v[e];
        }
      } else {
        if ((a.oc1 & 3) == 0 && (a.oc2 & 3) == 0 && o + 3 < a.O) {
          float* p; int accf;
          if (o < a.oc1) { p = a.d1 ? a.d1 + tok * a.oc1 + o : nullptr; accf = a.acc1; }
          else { p = a.d2 ? a.d2 + tok * a.oc2 + (o - a.oc1) : nullptr; accf = a.acc2; }
          if (p) {
            if (accf) { const float4 old = *reinterpret_cast<const float4*>(p); v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w; }
            *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
          }
        } else {
          for (int e = 0; e < 4; ++e) {
            const int c = o + e;
            if (c >= a.O) break;
            if (c < a.oc1) { if (a.d1) { float* p = a.d1 + tok * a.oc1 + c; *p = a.acc1 ? *p + v[e] : v[e]; } }
            else if (a.d2) { float* p = a.d2 + tok * a.oc2 + (c - a.oc1); *p = a.acc2 ? *p + v[e] : v[e]; }
          }
        }
      }
    }
  }
}

static void tile_counts(DirectArgs& a) {
  a.tiles_d = (a.D + dTD - 1) / dTD;
  a.tiles_h = (a.H + dTH - 1) / dTH;
  a.tiles_w = (a.W + dTW - 1) / dTW;
}

// Both return MICF_EUNSUPPORTED when the shape is outside what the direct kernel covers (caller falls back).
int conv3_fwd_direct(const float* x1, int c1, const float* x2, int c2, const float* w, const float* bias, float* y,
                     int y_layout, int B, int D, int H, int W, int N, hipStream_t stream) {
  if (N > 16 || (c1 & 3) || (c2 & 3) || !aligned16(x1) || (x2 && !aligned16(x2)) || !aligned16(y)) return MICF_EUNSUPPORTED;
  DirectArgs a{};
  a.in1 = x1; a.in2 = x2 ? x2 : x1; a.ic1 = c1; a.ic2 = c2; a.in_layout = 0; a.K = c1 + c2;
  a.w = w; a.Cin = c1 + c2; a.O = N; a.bias = bias; a.y = y; a.y_layout = y_layout;
  a.B = B; a.D = D; a.H = H; a.W = W;
  tile_counts(a);
  const int64_t blocks = (int64_t)B * a.tiles_d * a.tiles_h * a.tiles_w;
  if (blocks < 16 || W < 8) return MICF_EUNSUPPORTED;   // tiny grids (4^3 stage): the split-K implicit GEMM spreads better
  const int n_kchunk = (a.K + dKC - 1) / dKC;
  a.ysplit = 1;
  if (y_layout == 0 && blocks < 512 && n_kchunk > 1) {   // few token tiles: spread the channel chunks over workgroups too
    int want = (int)((1024 + blocks - 1) / blocks);
    a.ysplit = want < n_kchunk ? want : n_kchunk;
    if (a.ysplit > 1 && hipMemsetAsync(y, 0, sizeof(float) * (size_t)B * D * H * W * N, stream) != hipSuccess) return MICF_ELAUNCH;
  }
  hipLaunchKernelGGL(conv3_direct_kernel<false>, dim3((unsigned)blocks, a.ysplit), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

}  // namespace micf
