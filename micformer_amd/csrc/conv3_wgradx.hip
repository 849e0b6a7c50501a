// conv3_wgradx.hip -- weight gradient of the 3x3x3 / pad 1 convolution with 16 output channels (conv_offset[0], MS.py:314)
// on the matrix cores, with the WHOLE gradient slab of a workgroup resident in accumulator registers.
//
//   dW[n][c][tap] = sum_t dy[t, n] * in[t + off(tap), c]                 in = [x1 | x2], n < 16
//
// MFMA 16x16x4 fp32: rows i = 16 input channels, columns j = the 16 dy channels, k = 4 tokens.  A workgroup owns a slab of
// 48 input channels x 27 taps = 81 (tap, channel tile) pairs -> 21 accumulator tiles (84 AGPRs) in each of its 4 waves,
// and walks token tiles of 64 tokens: the input halo of the tile (the slab's 48 channels, voxel stride 52 floats: the four k
// lanes land on banks 0/16/32/48, conflict-free) is staged in LDS once, the dy fragment of a 16-token group is 4 registers,
// and every MFMA then costs exactly one 4-byte LDS read.  Partial slabs go to a workspace with coalesced 16-byte stores
// and a second kernel sums them over the workgroups and scatters into the [N][Cin][27] layout (+=).
// The LDS-tiled VALU kernel this replaces (conv3_wgrad.hip) reached 28 TFLOP/s at the 32^3 x 2 stage.
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "gemm_dma.h"

namespace micf {

constexpr int wCS = 48;                 // channels per slab (3 channel tiles): 67 KB of LDS, so a workgroup of the other stream fits beside it
constexpr int wDS = 20;                 // LDS stride of a dy row (floats): the four lane groups' tokens are 80 floats = 16 banks apart
constexpr int wXS = 52;                 // LDS voxel stride (floats): 4 * 52 = 208 = 16 (mod 64) -> the four k lanes hit disjoint banks
constexpr int wPairs = 27 * (wCS / 16); // 81 (tap, channel tile) pairs per slab
constexpr int wTPW = (wPairs + 3) / 4;  // 21 accumulator tiles per wave
constexpr int wSlabFloats = 4 * wTPW * 256;   // partial slab of one workgroup in the workspace (21.5 K floats)

constexpr int kWgxItems = 12;           // layers of one shape per launch (blockIdx.z): the two modalities' offset convs of every
                                        // depth slot a flush hands over (6 items at the six-slot stages)
struct WgxArgs {
  const float* dy[kWgxItems];           // channels-last [T, 16]
  const float* x1[kWgxItems]; const float* x2[kWgxItems];
  int c1, c2;
  float* ws;                            // [item][slabs][groups][wSlabFloats]
  float* bias_ws;                       // [item][groups][16]  (slab 0 only) or nullptr
  int B, D, H, W, tiles_d, tiles_h, tiles_w, tiles_per_group, groups, slabs, xcd_order;
};

// TW: tile extent along w (16 or 8).  Tile = 1 (d) x 64/TW (h) x TW (w) = 64 tokens; a token group is 16/TW h-rows x TW.
template <int TW, bool BF16>
__global__ void __launch_bounds__(256) conv3_wgradx_kernel(WgxArgs a) {
  constexpr int CH = 16 / TW, TH = 64 / TW;
  constexpr int HH = TH + 2, HW = TW + 2, HALO = 3 * HH * HW;
  extern __shared__ __attribute__((aligned(16))) float Xs[];           // [HALO][wXS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lr = lane >> 4;
  // XCD-aware order: workgroups are dealt to the 8 XCDs round-robin by linear id; the tile ranges (consecutive d planes, whose
  // halos overlap) handed to one XCD are made contiguous so the planes a workgroup shares with its neighbours hit that XCD's L2
  int slab = blockIdx.x, group = blockIdx.y;
  const int item = blockIdx.z;
  if (a.xcd_order) {
    const int L = gridDim.x * gridDim.y;
    const int lin = slab + gridDim.x * group;
    const int logical = (lin % 8) * (L / 8) + lin / 8;
    slab = logical % gridDim.x;
    group = logical / gridDim.x;
  }
  const float* __restrict__ a_dy = a.dy[item];
  const float* __restrict__ a_x1 = a.x1[item];
  const float* __restrict__ a_x2 = a.x2[item];
  const int Cin = a.c1 + a.c2;
  const int cbase = slab * wCS;
  const int64_t DHW = (int64_t)a.D * a.H * a.W;

  f32x4 acc[wTPW];
#pragma unroll
  for (int p = 0; p < wTPW; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;                                                    // colsum(dy) partial of column li (wave 0 only)
  int offs[wTPW];                                                      // LDS offset of (tap, channel tile) p of this wave
#pragma unroll
  for (int p = 0; p < wTPW; ++p) {
    const int pair = min(wave * wTPW + p, wPairs - 1);                  // the 3 surplus tiles of wave 3 recompute the last pair
    const int tap = pair / (wCS / 16), ct = pair % (wCS / 16);          // (never read back): no branch in the MFMA stream
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    offs[p] = (((kd - 1) * HH + (kh - 1)) * HW + (kw - 1)) * wXS + ct * 16;
  }

  const int ntiles = a.B * a.tiles_d * a.tiles_h * a.tiles_w;
  const int t_begin = group * a.tiles_per_group;
  const int t_end = min(ntiles, t_begin + a.tiles_per_group);
  // The halo and the dy rows of the NEXT tile are fetched into registers under the MFMAs of the current one (one workgroup per
  // CU: nothing else hides the load latency) and committed to LDS after the barrier that ends the current tile.  The MFMA phase
  // itself issues no global load (dy fragments come from LDS): loads return in order, one issued there would wait for the prefetch.
  constexpr int NQ = HALO * (wCS / 4);
  constexpr int NB = 16;
  static_assert(NQ <= 256 * NB, "one batch per tile");
  float* Dys = Xs + HALO * wXS;                                        // [64 tokens][wDS] dy of the tile
  float4 v[NB], vdy;
  auto fetch = [&](int tile) {
    int q = tile;
    const int tw = q % a.tiles_w; q /= a.tiles_w;
    const int th = q % a.tiles_h; q /= a.tiles_h;
    const int d0 = q % a.tiles_d; const int b = q / a.tiles_d;
    const int h0 = th * TH, w0 = tw * TW;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int idx = u * 256 + tid;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < NQ) {
        const int hv = idx / (wCS / 4), g = idx % (wCS / 4);
        const int hw = hv % HW, hh = (hv / HW) % HH, hd = hv / (HW * HH);
        const int dd = d0 + hd - 1, yy = h0 + hh - 1, ww = w0 + hw - 1;
        const int c = cbase + 4 * g;
        if ((unsigned)dd < (unsigned)a.D && (unsigned)yy < (unsigned)a.H && (unsigned)ww < (unsigned)a.W && c < Cin) {
          const int64_t tok = (int64_t)b * DHW + ((int64_t)dd * a.H + yy) * a.W + ww;
          v[u] = c < a.c1 ? *reinterpret_cast<const float4*>(a_x1 + tok * a.c1 + c)
                          : *reinterpret_cast<const float4*>(a_x2 + tok * a.c2 + (c - a.c1));
        }
      }
    }
    {
      const int tk = tid >> 2, yy = h0 + tk / TW, ww = w0 + tk % TW;    // tile-local token tk = lh * TW + lw
      vdy = make_float4(0.f, 0.f, 0.f, 0.f);
      if (yy < a.H && ww < a.W)
        vdy = *reinterpret_cast<const float4*>(a_dy + ((int64_t)b * DHW + ((int64_t)d0 * a.H + yy) * a.W + ww) * 16 + 4 * (tid & 3));
    }
  };
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                                                   // previous tile fully consumed
    // ---- commit: HALO voxels x 12 float4 (channels cbase .. cbase+47 of [x1 | x2]) and the 64 x 16 dy rows
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int idx = u * 256 + tid;
      if (idx < NQ) *reinterpret_cast<float4*>(&Xs[(idx / (wCS / 4)) * wXS + 4 * (idx % (wCS / 4))]) = v[u];
    }
    *reinterpret_cast<float4*>(&Dys[(tid >> 2) * wDS + 4 * (tid & 3)]) = vdy;
    // dy fragment of a 16-token group (k-permutation: in step s lane group lr supplies token j = 4*lr + s of the group)
    auto load_dy = [&](int g, float (&bv)[4]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[s] = g < 4 ? Dys[(g * 16 + 4 * lr + s) * wDS + li] : 0.f;
    };
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);
    float bv[4], bn[4];
    load_dy(0, bv);

    if constexpr (!BF16) {
    // ---- 4 token groups of 16 tokens
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
      load_dy(g + 1, bn);
      int vox[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int j = 4 * lr + s;
        const int lh = g * CH + j / TW, lw = j % TW;
        vox[s] = ((1 * HH + lh + 1) * HW + lw + 1) * wXS + li;          // centre voxel of the token, + channel li
      }
      if (a.bias_ws && wave == 0) bsum += bv[0] + bv[1] + bv[2] + bv[3];
      // s outer / pair inner: consecutive MFMAs hit different accumulators (no back-to-back dependency on one tile)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int p = 0; p < wTPW; ++p) {
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(Xs[vox[s] + offs[p]], bv[s], acc[p], 0, 0, 0);
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[s] = bn[s];
    }
    } else {
    // ---- bf16: k = 32 tokens per MFMA = two 16-token groups (lane group lr supplies tokens 4 lr + s of each)
#pragma unroll 1
    for (int g = 0; g < 4; g += 2) {
      float b2[4];
      load_dy(g + 1, b2);
      load_dy(g + 2, bn);
      int vox[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int j = 4 * lr + (s & 3);
        const int lh = (g + (s >> 2)) * CH + j / TW, lw = j % TW;
        vox[s] = ((1 * HH + lh + 1) * HW + lw + 1) * wXS + li;
      }
      if (a.bias_ws && wave == 0) bsum += (bv[0] + bv[1] + bv[2] + bv[3]) + (b2[0] + b2[1] + b2[2] + b2[3]);
      const bf16x8 bb = to_bf16x8(make_float4(bv[0], bv[1], bv[2], bv[3]), make_float4(b2[0], b2[1], b2[2], b2[3]));
#pragma unroll
      for (int p = 0; p < wTPW; ++p) {
        const float* xp = Xs + offs[p];
        const bf16x8 ba = to_bf16x8(make_float4(xp[vox[0]], xp[vox[1]], xp[vox[2]], xp[vox[3]]),
                                    make_float4(xp[vox[4]], xp[vox[5]], xp[vox[6]], xp[vox[7]]));
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[p], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[s] = bn[s];
    }
    }
  }
  // ---- partial slab -> workspace: [(wave*TPW + p)][lane][4], 16-byte stores
  float* out = a.ws + (((int64_t)item * a.slabs + slab) * a.groups + group) * wSlabFloats;
#pragma unroll
  for (int p = 0; p < wTPW; ++p)
    *reinterpret_cast<float4*>(out + ((wave * wTPW + p) * 64 + lane) * 4) = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
  if (a.bias_ws && slab == 0 && wave == 0) {
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 16) a.bias_ws[((int64_t)item * a.groups + group) * 16 + lane] = bsum;
  }
}

// ---- bf16 mode, round 3: the halo and the dy rows sit in LDS as bf16, token-major (no transposing staging), and the MFMA fragments
// -- 8 tokens of one channel per lane -- are read with the transposing LDS read ds_read_b64_tr_b16: lanes 4 j .. 4 j + 3 of a 16-lane
// group address token row j (8 bytes = 4 channels each), lane i receives column i.  The 4 rows of a read are ARBITRARY addresses,
// so a tap is just a byte offset on every row address (whole voxel rows: no sub-row misalignment), and one (tap, channel tile)
// costs 2 reads + 2 adds per MFMA where the fp32-staged kernel above pays 8 reads + 8 adds + 4 conversions (SQ counters put
// that kernel at 58 % issue-busy on its one wave per SIMD: instruction-bound).  34 KB of LDS instead of 72: a workgroup of the
// main chain fits beside it.  Same accumulator / workspace layout as above, same reduce kernel.
constexpr int wRSB = 104;               // bytes per halo voxel row: 48 bf16 + 8 pad (26 dwords: the 4 token rows of a read are bank-disjoint)
constexpr int wDYB = 40;                // bytes per dy token row: 16 bf16 + 8 pad
typedef short s16x4x __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_pair(unsigned a0, unsigned a1) {
  const s16x4x lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4x*)(uintptr_t)a0);
  const s16x4x hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4x*)(uintptr_t)a1);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int TW>
__global__ void __launch_bounds__(256) conv3_wgradx_b16_kernel(WgxArgs a) {
  constexpr int TH = 64 / TW;
  constexpr int HH = TH + 2, HW = TW + 2, HALO = 3 * HH * HW;
  extern __shared__ __attribute__((aligned(16))) char smem[];          // Xh [HALO][wRSB] | Dy [64][wDYB]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lr = lane >> 4;
  int slab = blockIdx.x, group = blockIdx.y;
  const int item = blockIdx.z;
  if (a.xcd_order) {
    const int L = gridDim.x * gridDim.y;
    const int lin = slab + gridDim.x * group;
    const int logical = (lin % 8) * (L / 8) + lin / 8;
    slab = logical % gridDim.x;
    group = logical / gridDim.x;
  }
  const float* __restrict__ a_dy = a.dy[item];
  const float* __restrict__ a_x1 = a.x1[item];
  const float* __restrict__ a_x2 = a.x2[item];
  const int Cin = a.c1 + a.c2;
  const int cbase = slab * wCS;
  const int64_t DHW = (int64_t)a.D * a.H * a.W;
  const unsigned xh = (unsigned)(uintptr_t)smem, dyl = xh + HALO * wRSB;

  f32x4 acc[wTPW];
#pragma unroll
  for (int p = 0; p < wTPW; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  int offs[wTPW];                                                      // byte offset of (tap, channel tile) p of this wave
#pragma unroll
  for (int p = 0; p < wTPW; ++p) {
    const int pair = min(wave * wTPW + p, wPairs - 1);
    const int tap = pair / (wCS / 16), ct = pair % (wCS / 16);
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    offs[p] = (((kd - 1) * HH + (kh - 1)) * HW + (kw - 1)) * wRSB + ct * 32;
  }
  // token t = 8 lr + 4 r + (li >> 2) of k-step ks (32 tokens = 32 / TW tile rows): its centre voxel row in the halo / its dy row
  unsigned abase[2][2], bbase[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int t = 8 * lr + 4 * r + (li >> 2);
      const int lh = ks * (32 / TW) + t / TW, lw = t % TW;
      abase[ks][r] = xh + (unsigned)(((1 * HH + lh + 1) * HW + lw + 1) * wRSB + 8 * (li & 3));
      bbase[ks][r] = dyl + (unsigned)((32 * ks + t) * wDYB + 8 * (li & 3));
    }
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);                         // column sums of dy (channels 4 (tid & 3) ..) over this thread's tokens

  const int ntiles = a.B * a.tiles_d * a.tiles_h * a.tiles_w;
  const int t_begin = group * a.tiles_per_group;
  const int t_end = min(ntiles, t_begin + a.tiles_per_group);
  constexpr int NQ = HALO * (wCS / 4);
  constexpr int NB = 16;
  static_assert(NQ <= 256 * NB, "one batch per tile");
  float4 v[NB], vdy;
  auto fetch = [&](int tile) {
    int q = tile;
    const int tw = q % a.tiles_w; q /= a.tiles_w;
    const int th = q % a.tiles_h; q /= a.tiles_h;
    const int d0 = q % a.tiles_d; const int b = q / a.tiles_d;
    const int h0 = th * TH, w0 = tw * TW;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int idx = u * 256 + tid;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < NQ) {
        const int hv = idx / (wCS / 4), g = idx % (wCS / 4);
        const int hw = hv % HW, hh = (hv / HW) % HH, hd = hv / (HW * HH);
        const int dd = d0 + hd - 1, yy = h0 + hh - 1, ww = w0 + hw - 1;
        const int c = cbase + 4 * g;
        if ((unsigned)dd < (unsigned)a.D && (unsigned)yy < (unsigned)a.H && (unsigned)ww < (unsigned)a.W && c < Cin) {
          const int64_t tok = (int64_t)b * DHW + ((int64_t)dd * a.H + yy) * a.W + ww;
          v[u] = c < a.c1 ? *reinterpret_cast<const float4*>(a_x1 + tok * a.c1 + c)
                          : *reinterpret_cast<const float4*>(a_x2 + tok * a.c2 + (c - a.c1));
        }
      }
    }
    const int tk = tid >> 2, yy = h0 + tk / TW, ww = w0 + tk % TW;      // tile-local token tk = lh * TW + lw
    vdy = make_float4(0.f, 0.f, 0.f, 0.f);
    if (yy < a.H && ww < a.W)
      vdy = *reinterpret_cast<const float4*>(a_dy + ((int64_t)b * DHW + ((int64_t)d0 * a.H + yy) * a.W + ww) * 16 + 4 * (tid & 3));
  };
  typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                                                   // previous tile fully consumed
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int idx = u * 256 + tid;
      if (idx < NQ)
        *reinterpret_cast<u32x2w*>(smem + (idx / (wCS / 4)) * wRSB + 8 * (idx % (wCS / 4))) = u32x2w{pack_bf16(v[u].x, v[u].y), pack_bf16(v[u].z, v[u].w)};
    }
    *reinterpret_cast<u32x2w*>(smem + HALO * wRSB + (tid >> 2) * wDYB + 8 * (tid & 3)) = u32x2w{pack_bf16(vdy.x, vdy.y), pack_bf16(vdy.z, vdy.w)};
    bs.x += vdy.x; bs.y += vdy.y; bs.z += vdy.z; bs.w += vdy.w;
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 bb = tr_pair(bbase[ks][0], bbase[ks][1]);
#pragma unroll
      for (int p = 0; p < wTPW; ++p) {
        const bf16x8 ba = tr_pair(abase[ks][0] + offs[p], abase[ks][1] + offs[p]);
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[p], 0, 0, 0);
      }
    }
  }
  // ---- partial slab -> workspace: [(wave*TPW + p)][lane][4], 16-byte stores
  float* out = a.ws + (((int64_t)item * a.slabs + slab) * a.groups + group) * wSlabFloats;
#pragma unroll
  for (int p = 0; p < wTPW; ++p)
    *reinterpret_cast<float4*>(out + ((wave * wTPW + p) * 64 + lane) * 4) = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
  if (a.bias_ws && slab == 0) {                                        // column sums of dy: over the lanes of one channel quad, then the waves
#pragma unroll
    for (int d = 4; d < 64; d <<= 1) {
      bs.x += __shfl_xor(bs.x, d, 64); bs.y += __shfl_xor(bs.y, d, 64); bs.z += __shfl_xor(bs.z, d, 64); bs.w += __shfl_xor(bs.w, d, 64);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (lane < 4) *reinterpret_cast<float4*>(red + (wave * 4 + lane) * 4) = bs;
    __syncthreads();
    if (tid < 16) a.bias_ws[((int64_t)item * a.groups + group) * 16 + tid] = red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid];
  }
}

// dw[n][c][tap] += sum over the groups' partial slabs;  dbias[n] += sum of the groups' column sums.
// Block = 64 slab elements x 4 slices of the group range (independent loads, unrolled), combined through LDS.
struct WgxOut { float* dw[kWgxItems]; float* dbias[kWgxItems]; };
__global__ void __launch_bounds__(256) conv3_wgradx_reduce_kernel(const float* __restrict__ ws_all, const float* __restrict__ bias_all,
                                                                  WgxOut o, int Cin, int slabs, int groups) {
  __shared__ float red[256];
  const int item = blockIdx.y;
  const float* __restrict__ ws = ws_all + (int64_t)item * slabs * groups * wSlabFloats;
  const float* __restrict__ bias_ws = bias_all ? bias_all + (int64_t)item * groups * 16 : nullptr;
  float* __restrict__ dw = o.dw[item];
  float* __restrict__ dbias = o.dbias[item];
  const int el = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int64_t id = (int64_t)blockIdx.x * 64 + el;
  const int64_t per_slab = (int64_t)wPairs * 256;
  const bool main = id < per_slab * slabs;
  const bool bias = !main && bias_ws && dbias && id < per_slab * slabs + 16;
  float acc = 0.f;
  int slab = 0, e = 0;
  if (main) {
    slab = (int)(id / per_slab);
    e = (int)(id % per_slab);
    const float* p = ws + (int64_t)slab * groups * wSlabFloats + e;
    int g = slice;
    for (; g + 28 < groups; g += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(g + 4 * u) * wSlabFloats];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; g < groups; g += 4) acc += p[(int64_t)g * wSlabFloats];
  } else if (bias) {
    const int n = (int)(id - per_slab * slabs);
    for (int g = slice; g < groups; g += 4) acc += bias_ws[g * 16 + n];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (slice != 0) return;
  acc = red[el] + red[64 + el] + red[128 + el] + red[192 + el];
  if (main) {
    const int pair = e >> 8, lane = (e >> 2) & 63, v = e & 3;
    const int tap = pair / (wCS / 16), ct = pair % (wCS / 16);
    const int n = lane & 15, c = slab * wCS + ct * 16 + 4 * (lane >> 4) + v;
    if (c < Cin) dw[((int64_t)n * Cin + c) * 27 + tap] += acc;
  } else if (bias) {
    dbias[(int)(id - per_slab * slabs)] += acc;
  }
}

struct WgxPlan { int slabs, groups, tiles_per_group, tw; int64_t floats; };

static WgxPlan wgx_plan(int B, int D, int H, int W, int Cin, int items = 1) {
  WgxPlan p;
  p.tw = W >= 12 ? 16 : 8;
  const int th = 64 / p.tw;
  const int ntiles = B * D * ((H + th - 1) / th) * ((W + p.tw - 1) / p.tw);
  p.slabs = (Cin + wCS - 1) / wCS;
  constexpr int wg_target = 256;
  int groups = (wg_target + p.slabs * items - 1) / (p.slabs * items);   // ~one workgroup per CU over all items
  if (groups < 4) groups = 4 < ntiles ? 4 : ntiles;
  if (groups > ntiles) groups = ntiles;
  p.tiles_per_group = (ntiles + groups - 1) / groups;
  p.groups = (ntiles + p.tiles_per_group - 1) / p.tiles_per_group;
  p.floats = (int64_t)p.slabs * p.groups * wSlabFloats + (int64_t)p.groups * 16;
  return p;
}

int64_t conv3_wgradx_workspace(int B, int D, int H, int W, int N, int c1, int c2, int items) {
  if (N != 16 || W < 4 || (c1 & 3) || (c2 & 3) || items < 1 || items > kWgxItems) return 0;   // (W = 4: masked 8-wide tiles)
  return wgx_plan(B, D, H, W, c1 + c2, items).floats * items;
}

// MICF_EUNSUPPORTED when the shape / workspace is outside what this kernel covers (caller falls back).
// n items of ONE shape per launch (blockIdx.z): dy / x1 / x2 / dw / dbias per item.
int conv3_wgradx_items(const float* const* dy, const float* const* x1, const float* const* x2, float* const* dw, float* const* dbias,
                       int n, int c1, int c2, int B, int D, int H, int W, int N, float* ws, int64_t ws_floats, hipStream_t stream,
                       int dtype) {
  const int64_t need = conv3_wgradx_workspace(B, D, H, W, N, c1, c2, n);
  if (need == 0 || !ws || ws_floats < need || !aligned16(ws)) return MICF_EUNSUPPORTED;
  const WgxPlan p = wgx_plan(B, D, H, W, c1 + c2, n);
  WgxArgs a{};
  WgxOut o{};
  bool any_bias = false;
  for (int i = 0; i < n; ++i) {
    if (!dy[i] || !x1[i] || !dw[i] || !aligned16(dy[i]) || !aligned16(x1[i]) || (x2 && x2[i] && !aligned16(x2[i]))) return MICF_EUNSUPPORTED;
    a.dy[i] = dy[i]; a.x1[i] = x1[i]; a.x2[i] = (x2 && x2[i]) ? x2[i] : x1[i];
    o.dw[i] = dw[i]; o.dbias[i] = dbias ? dbias[i] : nullptr;
    any_bias = any_bias || o.dbias[i];
  }
  a.c1 = c1; a.c2 = c2;
  a.ws = ws; a.bias_ws = any_bias ? ws + (int64_t)n * p.slabs * p.groups * wSlabFloats : nullptr;
  a.B = B; a.D = D; a.H = H; a.W = W;
  const int th = 64 / p.tw;
  a.tiles_d = D; a.tiles_h = (H + th - 1) / th; a.tiles_w = (W + p.tw - 1) / p.tw;
  a.tiles_per_group = p.tiles_per_group; a.groups = p.groups; a.slabs = p.slabs;
  a.xcd_order = ((p.slabs * p.groups) % 8 == 0) ? 1 : 0;   // XCD-contiguous tile ranges
  static std::once_flag attr_once;       // > 64 KiB of dynamic LDS needs the opt-in once per process
  std::call_once(attr_once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgradx_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgradx_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgradx_b16_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgradx_b16_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  });
  const dim3 grid(p.slabs, p.groups, n);
  if (p.tw == 16) {
    constexpr int HALO = 3 * (4 + 2) * (16 + 2);
    if (dtype == MICF_DTYPE_BF16) hipLaunchKernelGGL((conv3_wgradx_b16_kernel<16>), grid, dim3(256), HALO * wRSB + 64 * wDYB, stream, a);
    else hipLaunchKernelGGL((conv3_wgradx_kernel<16, false>), grid, dim3(256), sizeof(float) * (HALO * wXS + 64 * wDS), stream, a);
  } else {
    constexpr int HALO = 3 * (8 + 2) * (8 + 2);
    if (dtype == MICF_DTYPE_BF16) hipLaunchKernelGGL((conv3_wgradx_b16_kernel<8>), grid, dim3(256), HALO * wRSB + 64 * wDYB, stream, a);
    else hipLaunchKernelGGL((conv3_wgradx_kernel<8, false>), grid, dim3(256), sizeof(float) * (HALO * wXS + 64 * wDS), stream, a);
  }
  if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
  const int64_t ne = (int64_t)wPairs * 256 * p.slabs + 16;
  hipLaunchKernelGGL(conv3_wgradx_reduce_kernel, dim3((unsigned)((ne + 63) / 64), n), dim3(256), 0, stream, ws, a.bias_ws, o,
                     c1 + c2, p.slabs, p.groups);
  return hipGetLastError() == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

int conv3_wgradx(const float* dy, const float* x1, int c1, const float* x2, int c2, float* dw, float* dbias, int B, int D, int H,
                 int W, int N, float* ws, int64_t ws_floats, hipStream_t stream, int dtype) {
  return conv3_wgradx_items(&dy, &x1, &x2, &dw, &dbias, 1, c1, c2, B, D, H, W, N, ws, ws_floats, stream, dtype);
}

}  // namespace micf
