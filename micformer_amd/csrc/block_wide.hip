// block_wide.hip -- the window-local part of a (Cross)TransformerBlock3D for the FEW-TOKEN stages (C = 384 at 4^3 .. 8^3
// tokens per sample): same inputs, outputs and arithmetic as block_fwd.hip / block_bwd.hip, different decomposition.
//
// At the bottom of the U the block has a few hundred tokens and 1.8 M weights: a tile-per-workgroup kernel would stream the
// whole 3.5-7 MB weight set through 8-16 compute units, and the per-op path spends ~18 us per GEMM launch on 128 tokens.
// Here every GEMM of the block is split over its OUTPUT FEATURES instead: one wave owns a 16-token x 16-feature tile, holds its
// 16 activation rows in registers as MFMA B fragments (K <= 384 per wave: K = 4C products are split over the 4 waves of the
// workgroup and reduced through LDS), and streams one 16-row weight tile straight from L2 / HBM -- hundreds of independent
// waves per launch, each a single load round trip deep.  Everything row-local is fused around the GEMMs:
//
//   forward   F1  LN1 (registers) -> q | k | v of ONE head -> 8-token window attention of that head      (tile x head)
//             F2  x1 = x + s1 (o Wp^T + bp)                                                               (tile x 16 features)
//             F3  LN2 (registers) -> h = xn2 W1^T + b1, g = GELU(h)                                       (tile x 32 features)
//             F4  y = x1 + s2 (g W2^T + b2)                                       (tile x 16 features, K split over 4 waves)
//   backward  B1  dh = s2 (dy W2) GELU'(h)                                                                (tile x 32 features)
//             B2  dxn2 = dh W1                                                     (tile x 16 features, K split over 4 waves)
//             B3  dx1 = dy + LN2'(dxn2) (registers) -> do = s1 dx1 Wp of ONE head -> attention' -> dq dk dv  (tile x head)
//             B4  dq Wq (+) dkv Wkv                                                (tile x 16 features, K split over 3 waves)
//             B5  self: dx = dx1 + LN1'(.)                                                                (tile, 4 column slices)
//
// 4 + 5 launches per block PAIR (both modalities per launch, XCDs 0-3 / 4-7) instead of ~16 + ~20 per-op launches.  The
// LayerNorm gain / bias gradients leave as per-tile partial rows for micf_layernorm_bwd_finish, like the fused kernels.
// Operand layout: a lane (li = lane & 15, lr = lane >> 4) holds, of row li, the float4s at columns 16 i + 4 lr (fp32 MFMA
// 16x16x4 with the k-permutation of gemm_dma.h) or 32 (i / 2) + 8 lr + 4 (i & 1) (bf16 MFMA 16x16x32, natural order).
#include <cstdlib>

#include "block_fused.h"
#include "attn_fp8.h"

namespace micf {
namespace wide {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool BF16> struct WSel { typedef float T; };
template <> struct WSel<true> { typedef uint16_t T; };

template <int K, bool BF16> struct Frag { float4 v[K / 16]; };          // per-lane slice of a 16-row x K MFMA operand
template <int K> struct Frag<K, true> { u32x4 v[K / 32]; };

template <bool BF16> __device__ __forceinline__ int fcol(int i, int lr) {
  return BF16 ? 32 * (i >> 1) + 8 * lr + 4 * (i & 1) : 16 * i + 4 * lr;
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int K, bool BF16> __device__ __forceinline__ void load_raw(const float* __restrict__ row, int lr, bool ok, float4 (&r)[K / 16]) {
#pragma unroll
  for (int i = 0; i < K / 16; ++i) {
    const float4 t = ld4g(row + fcol<BF16>(i, lr));
    r[i] = ok ? t : f4zero();
  }
}
template <int K, bool BF16> __device__ __forceinline__ void pack(const float4 (&r)[K / 16], Frag<K, BF16>& f) {
  if constexpr (BF16) {
#pragma unroll
    for (int j = 0; j < K / 32; ++j)
      f.v[j] = u32x4{pack_bf16(r[2 * j].x, r[2 * j].y), pack_bf16(r[2 * j].z, r[2 * j].w), pack_bf16(r[2 * j + 1].x, r[2 * j + 1].y),
                     pack_bf16(r[2 * j + 1].z, r[2 * j + 1].w)};
  } else {
#pragma unroll
    for (int i = 0; i < K / 16; ++i) f.v[i] = r[i];
  }
}
// weight tile: rows n .. n + 15 (n a multiple of 16; this lane's output feature is n + li) of W [*, LD], K columns from k0.
// bf16 shadow weights are K16-blocked (block_fused.h): a wave's load instruction covers 1 KB of consecutive addresses.
template <int K, bool BF16> __device__ __forceinline__ void load_w(const typename WSel<BF16>::T* __restrict__ W, int n, int LD, int k0,
                                                                   int li, int lr, Frag<K, BF16>& f) {
  if constexpr (BF16) {
    const uint16_t* p = W + (int64_t)n * LD + k0 * 16 + li * 16 + 8 * (lr & 1);
#pragma unroll
    for (int j = 0; j < K / 32; ++j) f.v[j] = *reinterpret_cast<const u32x4*>(p + (2 * j + (lr >> 1)) * 256);
  } else {
    const float* p = W + (int64_t)n * LD + k0 * 16 + li * 16 + 4 * lr;
#pragma unroll
    for (int i = 0; i < K / 16; ++i) f.v[i] = *reinterpret_cast<const float4*>(p + 256 * i);
  }
}
// D[feature 4 lr + r][token li] = sum_k A[feature][k] B[token][k]: the lane ends up with 4 consecutive features of token li
template <int K, bool BF16> __device__ __forceinline__ float4 mma(const Frag<K, BF16>& a, const Frag<K, BF16>& b) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  if constexpr (BF16) {
#pragma unroll
    for (int j = 0; j < K / 32; ++j) {
      if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v[j]), __builtin_bit_cast(bf16x8, b.v[j]), acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v[j]), __builtin_bit_cast(bf16x8, b.v[j]), acc0, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < K / 16; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i].x, b.v[i].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i].y, b.v[i].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i].z, b.v[i].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i].w, b.v[i].w, acc1, 0, 0, 0);
    }
  }
  return make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
}

__device__ __forceinline__ float sum_lr(float v) {                     // over the 4 lanes (lr) that share a row
  v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float4 sum_li(float4 v) {                   // over the 16 rows (li) of the tile, per column
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    v.x += __shfl_xor(v.x, m, 64); v.y += __shfl_xor(v.y, m, 64); v.z += __shfl_xor(v.z, m, 64); v.w += __shfl_xor(v.w, m, 64);
  }
  return v;
}

struct Row { int tk; bool ok; };                                       // token of tile row r (clamped to 0 when out of range)
__device__ __forceinline__ Row tile_row(const TileGeo& geo, int tile, int r) {
  const int win = tile * 2 + (r >> 3);
  Row o;
  o.ok = win < geo.nwin;
  o.tk = o.ok ? geo.token(win, r & 7) : 0;
  return o;
}
__device__ __forceinline__ float row_scale(const TileGeo& geo, const float* s, const Row& row) {
  return (s && row.ok) ? s[(int)geo.f_rps.div((uint32_t)row.tk)] : 1.f;
}
// work item of this workgroup: the two groups take the XCD halves (each L2 then caches one weight set)
__device__ __forceinline__ void decode(int G, int& grp, int& idx) {
  if (G == 2) { const int xcd = blockIdx.x & 7; grp = xcd >> 2; idx = (int)(blockIdx.x >> 3) * 4 + (xcd & 3); }
  else { grp = 0; idx = blockIdx.x; }
}
static unsigned grid_for(int G, int n) { return G == 2 ? (unsigned)((n + 3) / 4 * 8) : (unsigned)n; }

// LayerNorm forward of the row slices in r (in place); the mean / rstd of the row come back
template <int C, bool BF16> __device__ __forceinline__ void ln_rows(float4 (&r)[C / 16], int lr, bool ok, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float eps, float& mu, float& rs) {
  constexpr int NF = C / 16;
  const float invC = 1.0f / (float)C;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NF; ++i) s += (r[i].x + r[i].y) + (r[i].z + r[i].w);
  mu = sum_lr(s) * invC;
  float qd = 0.f;
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const float d0 = r[i].x - mu, d1 = r[i].y - mu, d2 = r[i].z - mu, d3 = r[i].w - mu;
    qd += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  rs = 1.0f / sqrtf(sum_lr(qd) * invC + eps);
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int c = fcol<BF16>(i, lr);
    const float4 gm = ld4g(gamma + c), bt = ld4g(beta + c);
    float4 y = make_float4((r[i].x - mu) * rs * gm.x + bt.x, (r[i].y - mu) * rs * gm.y + bt.y, (r[i].z - mu) * rs * gm.z + bt.z,
                           (r[i].w - mu) * rs * gm.w + bt.w);
    r[i] = ok ? y : f4zero();
  }
}

// LayerNorm backward of the row slices: d (gradient w.r.t. the normalised output, pre-gain; replaced by the result),
// xr = the LayerNorm input rows (replaced by xhat); result = rs (g d - mean(g d) - xhat mean(g d xhat)).
template <int C, bool BF16> __device__ __forceinline__ void ln_bwd_rows(float4 (&d)[C / 16], float4 (&xr)[C / 16], float4 (&gd)[C / 16], int lr,
                                                                        const float* __restrict__ gamma, float mu, float rs) {
  constexpr int NF = C / 16;
  const float invC = 1.0f / (float)C;
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const float4 gm = ld4g(gamma + fcol<BF16>(i, lr));
    xr[i] = make_float4((xr[i].x - mu) * rs, (xr[i].y - mu) * rs, (xr[i].z - mu) * rs, (xr[i].w - mu) * rs);
    gd[i] = make_float4(gm.x * d[i].x, gm.y * d[i].y, gm.z * d[i].z, gm.w * d[i].w);
    sa += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
    sb += (gd[i].x * xr[i].x + gd[i].y * xr[i].y) + (gd[i].z * xr[i].z + gd[i].w * xr[i].w);
  }
  const float Am = sum_lr(sa) * invC, Bm = sum_lr(sb) * invC;
#pragma unroll
  for (int i = 0; i < NF; ++i)
    gd[i] = make_float4(rs * (gd[i].x - Am - xr[i].x * Bm), rs * (gd[i].y - Am - xr[i].y * Bm), rs * (gd[i].z - Am - xr[i].z * Bm),
                        rs * (gd[i].w - Am - xr[i].w * Bm));
}
// per-tile partial sums of the gain / bias gradients for the float4 slots i = first, first + step, ...: part[c] = sum_rows d xhat,
// part[C + c] = sum_rows d
template <int C, bool BF16> __device__ __forceinline__ void ln_partials(const float4 (&d)[C / 16], const float4 (&xh)[C / 16], int first, int step,
                                                                        int li, int lr, float* __restrict__ part) {
#pragma unroll
  for (int i = 0; i < C / 16; ++i) {
    if (i % step != first) continue;
    const float4 ag = sum_li(make_float4(d[i].x * xh[i].x, d[i].y * xh[i].y, d[i].z * xh[i].z, d[i].w * xh[i].w));
    const float4 ab = sum_li(d[i]);
    if (li == 0) {
      const int c = fcol<BF16>(i, lr);
      st4g(part + c, ag);
      st4g(part + C + c, ab);
    }
  }
}

struct FwdArgs { micf_block_fwd_group g[2]; TileGeo geo; int G, tiles; float eps, scale; int att8; };   // att8: attn_fp8.h
struct BwdArgs { micf_block_bwd_group g[2]; TileGeo geo; int G, tiles; float scale; int attn_mfma; };   // attn_mfma: attn16_bwd_bf16 (bf16 mode)

// ---------------------------------------------------------------------------------------------------------------- forward
// F1: wave = (tile, head).  LN1 -> q_h | k_h | v_h (+ bias) -> attention of the tile's 2 windows -> o_h
template <int C, int HD, bool BF16>
__global__ void __launch_bounds__(256) f1_kernel(const FwdArgs a) {
  constexpr int NF = C / 16, heads = C / HD, HT = HD / 16, QS = HD + 4, NTL = 3 * HT;
  using WT = typename WSel<BF16>::T;
  __shared__ __attribute__((aligned(16))) float sm[4][3][16][QS];
  int grp, idx;
  decode(a.G, grp, idx);
  if (idx >= a.tiles * (heads / 4)) return;
  const int tile = idx / (heads / 4), wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const int head = (idx % (heads / 4)) * 4 + wave;
  const micf_block_fwd_group& g = a.g[grp];
  const int64_t T = a.geo.T;
  const Row row = tile_row(a.geo, tile, li);
  const WT* wq = static_cast<const WT*>(g.wq), *wkv = static_cast<const WT*>(g.wkv);

  auto load_tile = [&](int t, Frag<C, BF16>& f) {                     // weight rows of tile t (q: 0..HT-1, k, v)
    const int part = t / HT, n = (part == 2 ? C : 0) + head * HD + 16 * (t % HT);
    load_w<C, BF16>(part == 0 ? wq : wkv, n, C, 0, li, lr, f);
  };
  Frag<C, BF16> fa[2];
  load_tile(0, fa[0]);                                                // in flight during the LayerNorm

  Frag<C, BF16> bq, bkv;
  {
    float4 raw[NF];
    load_raw<C, BF16>(g.x + (int64_t)row.tk * C, lr, row.ok, raw);
    float mu, rs;
    ln_rows<C, BF16>(raw, lr, row.ok, g.ln1_g, g.ln1_b, a.eps, mu, rs);
    if (head == 0 && row.ok) {
      if (g.xn) {
#pragma unroll
        for (int i = 0; i < NF; ++i) st4g(g.xn + (int64_t)row.tk * C + fcol<BF16>(i, lr), raw[i]);
      }
      if (lr == 0) { g.stats[row.tk] = mu; g.stats[T + row.tk] = rs; }
    }
    pack<C, BF16>(raw, bq);
    if (g.kvsrc) {
      load_raw<C, BF16>(g.kvsrc + (int64_t)row.tk * C, lr, row.ok, raw);
      pack<C, BF16>(raw, bkv);
    } else {
      bkv = bq;
    }
  }
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    if (t + 1 < NTL) load_tile(t + 1, fa[(t + 1) & 1]);
    const int part = t / HT, hh = t % HT;
    const int n = (part == 2 ? C : 0) + head * HD + 16 * hh + 4 * lr;
    float4 v = mma<C, BF16>(fa[t & 1], part == 0 ? bq : bkv);
    const float4 b = ld4g((part == 0 ? g.bq : g.bkv) + n);
    v = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
    *reinterpret_cast<float4*>(&sm[wave][part][li][16 * hh + 4 * lr]) = v;
    if (row.ok) {
      if (part == 0) st4g(g.q + (int64_t)row.tk * C + n, v);
      else st4g(g.kv + (int64_t)row.tk * 2 * C + n, v);
    }
  }
  __syncthreads();
  if (BF16 && a.att8) {                                               // e4m3 operands on the matrix cores: the wave's (tile, head) is one unit
    float4 ov[HD / 16];
    if (a.att8 == 2) attn16_fp8<HD>(&sm[wave][0][0][0], &sm[wave][1][0][0], &sm[wave][2][0][0], QS, a.scale, ov);
    else attn16_bf16<HD>(&sm[wave][0][0][0], &sm[wave][1][0][0], &sm[wave][2][0][0], QS, a.scale, ov);
    if (row.ok) {
#pragma unroll
      for (int cb = 0; cb < HD / 16; ++cb) st4g(g.o + (int64_t)row.tk * C + head * HD + 16 * cb + 4 * lr, ov[cb]);
    }
  } else if (lane < 16 && row.ok) {                                   // attention row = lane (li == lane): 8 keys of its window
    const int r0 = lane & ~7;
    float qr[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) qr[d] = sm[wave][0][lane][d] * a.scale;
    float sj[8], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc += qr[d] * sm[wave][1][r0 + j][d];
      sj[j] = acc;
      mx = fmaxf(mx, acc);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sj[j] = expf(sj[j] - mx); den += sj[j]; }
    const float inv = 1.0f / den;
    float oa[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) oa[d] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float pj = sj[j] * inv;
#pragma unroll
      for (int d = 0; d < HD; ++d) oa[d] += pj * sm[wave][2][r0 + j][d];
    }
#pragma unroll
    for (int d = 0; d < HD; d += 4)
      st4g(g.o + (int64_t)row.tk * C + head * HD + d, make_float4(oa[d], oa[d + 1], oa[d + 2], oa[d + 3]));
  }
}

// F2: wave = (tile, 16 features).  x1 = x + s1 (o Wp^T + bp)
template <int C, bool BF16>
__global__ void __launch_bounds__(256) f2_kernel(const FwdArgs a) {
  constexpr int NF = C / 16, NCB = C / 64;
  using WT = typename WSel<BF16>::T;
  int grp, idx;
  decode(a.G, grp, idx);
  if (idx >= a.tiles * NCB) return;
  const int tile = idx / NCB, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const int n0 = ((idx % NCB) * 4 + wave) * 16;
  const micf_block_fwd_group& g = a.g[grp];
  const Row row = tile_row(a.geo, tile, li);
  Frag<C, BF16> fa, b;
  load_w<C, BF16>(static_cast<const WT*>(g.wp), n0, C, 0, li, lr, fa);
  {
    float4 raw[NF];
    load_raw<C, BF16>(g.o + (int64_t)row.tk * C, lr, row.ok, raw);
    pack<C, BF16>(raw, b);
  }
  const float4 v = mma<C, BF16>(fa, b);
  if (row.ok) {
    const int n = n0 + 4 * lr;
    const float s1 = row_scale(a.geo, g.s1, row);
    const float4 bp = ld4g(g.bp + n), xv = ld4g(g.x + (int64_t)row.tk * C + n);
    st4g(g.x1 + (int64_t)row.tk * C + n,
         make_float4(xv.x + s1 * (v.x + bp.x), xv.y + s1 * (v.y + bp.y), xv.z + s1 * (v.z + bp.z), xv.w + s1 * (v.w + bp.w)));
  }
}

// F3: wave = (tile, 2 x 16 hidden features).  xn2 = LN2(x1); h = xn2 W1^T + b1; g = GELU(h)
template <int C, bool BF16>
__global__ void __launch_bounds__(256) f3_kernel(const FwdArgs a) {
  constexpr int NF = C / 16, Hd = 4 * C, NT = 2, NCB = Hd / (64 * NT);
  using WT = typename WSel<BF16>::T;
  int grp, idx;
  decode(a.G, grp, idx);
  if (idx >= a.tiles * NCB) return;
  const int tile = idx / NCB, cb = idx % NCB, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const int n0 = (cb * 4 + wave) * NT * 16;
  const micf_block_fwd_group& g = a.g[grp];
  const int64_t T = a.geo.T;
  const Row row = tile_row(a.geo, tile, li);
  const WT* w1 = static_cast<const WT*>(g.w1);
  Frag<C, BF16> fa[2], b;
  load_w<C, BF16>(w1, n0, C, 0, li, lr, fa[0]);
  {
    float4 raw[NF];
    load_raw<C, BF16>(g.x1 + (int64_t)row.tk * C, lr, row.ok, raw);
    float mu, rs;
    ln_rows<C, BF16>(raw, lr, row.ok, g.ln2_g, g.ln2_b, a.eps, mu, rs);
    if (cb == 0 && wave == 0 && row.ok) {
#pragma unroll
      for (int i = 0; i < NF; ++i) st4g(g.xn2 + (int64_t)row.tk * C + fcol<BF16>(i, lr), raw[i]);
      if (lr == 0) { g.stats[2 * T + row.tk] = mu; g.stats[3 * T + row.tk] = rs; }
    }
    pack<C, BF16>(raw, b);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t + 1 < NT) load_w<C, BF16>(w1, n0 + 16 * (t + 1), C, 0, li, lr, fa[(t + 1) & 1]);
    const float4 v = mma<C, BF16>(fa[t & 1], b);
    if (row.ok) {
      const int n = n0 + 16 * t + 4 * lr;
      const float4 b1 = ld4g(g.b1 + n);
      const float4 h = make_float4(v.x + b1.x, v.y + b1.y, v.z + b1.z, v.w + b1.w);
      st_h4<BF16>(g.h, (int64_t)row.tk * Hd + n, h);
      st4g(g.g + (int64_t)row.tk * Hd + n, make_float4(gelu_t<BF16>(h.x), gelu_t<BF16>(h.y), gelu_t<BF16>(h.z), gelu_t<BF16>(h.w)));
    }
  }
}

// F4: workgroup = (tile, 16 features), wave = one K quarter of the hidden dimension.  y = x1 + s2 (g W2^T + b2)
template <int C, bool BF16>
__global__ void __launch_bounds__(256) f4_kernel(const FwdArgs a) {
  constexpr int NF = C / 16, Hd = 4 * C, NCB = C / 16;
  using WT = typename WSel<BF16>::T;
  __shared__ float4 red[3][64];
  int grp, idx;
  decode(a.G, grp, idx);
  if (idx >= a.tiles * NCB) return;
  const int tile = idx / NCB, n0 = (idx % NCB) * 16, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const micf_block_fwd_group& g = a.g[grp];
  const Row row = tile_row(a.geo, tile, li);
  Frag<C, BF16> fa, b;
  load_w<C, BF16>(static_cast<const WT*>(g.w2), n0, Hd, wave * C, li, lr, fa);
  {
    float4 raw[NF];
    load_raw<C, BF16>(g.g + (int64_t)row.tk * Hd + wave * C, lr, row.ok, raw);
    pack<C, BF16>(raw, b);
  }
  float4 v = mma<C, BF16>(fa, b);
  if (wave) red[wave - 1][lane] = v;
  __syncthreads();
  if (wave == 0 && row.ok) {
#pragma unroll
    for (int w = 0; w < 3; ++w) { const float4 p = red[w][lane]; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
    const int n = n0 + 4 * lr;
    const float s2 = row_scale(a.geo, g.s2, row);
    const float4 b2 = ld4g(g.b2 + n), xv = ld4g(g.x1 + (int64_t)row.tk * C + n);
    st4g(g.y + (int64_t)row.tk * C + n,
         make_float4(xv.x + s2 * (v.x + b2.x), xv.y + s2 * (v.y + b2.y), xv.z + s2 * (v.z + b2.z), xv.w + s2 * (v.w + b2.w)));
  }
}

// --------------------------------------------------------------------------------------------------------------- backward
// B1: wave = (tile, 2 x 16 hidden features).  dh = s2 (dy W2) GELU'(h)
template <int C, bool BF16>
__global__ void __launch_bounds__(256) b1_kernel(const BwdArgs a) {
  constexpr int NF = C / 16, Hd = 4 * C, NT = 2, NCB = Hd / (64 * NT);
  using WT = typename WSel<BF16>::T;
  int grp, idx;
  decode(a.G, grp, idx);
  if (idx >= a.tiles * NCB) return;
  const int tile = idx / NCB, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const int n0 = ((idx % NCB) * 4 + wave) * NT * 16;
  const micf_block_bwd_group& g = a.g[grp];
  const Row row = tile_row(a.geo, tile, li);
  const WT* w2t = static_cast<const WT*>(g.w2t);
  Frag<C, BF16> fa[2], b;
  load_w<C, BF16>(w2t, n0, C, 0, li, lr, fa[0]);
  {
    float4 raw[NF];
    load_raw<C, BF16>(g.dy + (int64_t)row.tk * C, lr, row.ok, raw);
    pack<C, BF16>(raw, b);
  }
  const float s2 = row_scale(a.geo, g.s2, row);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t + 1 < NT) load_w<C, BF16>(w2t, n0 + 16 * (t + 1), C, 0, li, lr, fa[(t + 1) & 1]);
    const float4 v = mma<C, BF16>(fa[t & 1], b);
    if (row.ok) {
      const int n = n0 + 16 * t + 4 * lr;
      const float4 h = ld_h4<BF16>(g.h, (int64_t)row.tk * Hd + n);
      st4g(g.dh + (int64_t)row.tk * Hd + n, make_float4(s2 * v.x * gelu_grad_t<BF16>(h.x), s2 * v.y * gelu_grad_t<BF16>(h.y),
                                                        s2 * v.z * gelu_grad_t<BF16>(h.z), s2 * v.w * gelu_grad_t<BF16>(h.w)));
    }
  }
}

// B2: workgroup = (tile, 16 features), wave = one K quarter.  dxn2 = dh W1 -> g.dx (scratch until B4 / B5 overwrite it)
template <int C, bool BF16>
__global__ void __launch_bounds__(256) b2_kernel(const BwdArgs a) {
  constexpr int NF = C / 16, Hd = 4 * C, NCB = C / 16;
  using WT = typename WSel<BF16>::T;
  __shared__ float4 red[3][64];
  int grp, idx;
  decode(a.G, grp, idx);
  if (idx >= a.tiles * NCB) return;
  const int tile = idx / NCB, n0 = (idx % NCB) * 16, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const micf_block_bwd_group& g = a.g[grp];
  const Row row = tile_row(a.geo, tile, li);
  Frag<C, BF16> fa, b;
  load_w<C, BF16>(static_cast<const WT*>(g.w1t), n0, Hd, wave * C, li, lr, fa);
  {
    float4 raw[NF];
    load_raw<C, BF16>(g.dh + (int64_t)row.tk * Hd + wave * C, lr, row.ok, raw);
    pack<C, BF16>(raw, b);
  }
  float4 v = mma<C, BF16>(fa, b);
  if (wave) red[wave - 1][lane] = v;
  __syncthreads();
  if (wave == 0 && row.ok) {
#pragma unroll
    for (int w = 0; w < 3; ++w) { const float4 p = red[w][lane]; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
    st4g(g.dx + (int64_t)row.tk * C + n0 + 4 * lr, v);
  }
}

// B3: wave = (tile, head).  dx1 = dy + LN2'(dxn2); do_h = s1 dx1 Wp[:, head]; attention backward of the head -> dq_h dk_h dv_h
template <int C, int HD, bool BF16>
__global__ void __launch_bounds__(256) b3_kernel(const BwdArgs a) {
  constexpr int NF = C / 16, heads = C / HD, HT = HD / 16, QS = HD + 4;
  using WT = typename WSel<BF16>::T;
  __shared__ __attribute__((aligned(16))) float sm[4][4][16][QS];     // q, k, v, do of the head
  __shared__ float ps[4][16][16];                                     // P row | dS row of every attention row
  int grp, idx;
  decode(a.G, grp, idx);
  if (idx >= a.tiles * (heads / 4)) return;
  const int tile = idx / (heads / 4), wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const int head = (idx % (heads / 4)) * 4 + wave;
  const micf_block_bwd_group& g = a.g[grp];
  const int64_t T = a.geo.T;
  const Row row = tile_row(a.geo, tile, li);
  const WT* wpt = static_cast<const WT*>(g.wpt);
  Frag<C, BF16> fa[2], b;
  load_w<C, BF16>(wpt, head * HD, C, 0, li, lr, fa[0]);
  // the head's q | k | v rows -> LDS (also in flight during the LayerNorm backward)
#pragma unroll
  for (int hh = 0; hh < HT; ++hh) {
    const int n = head * HD + 16 * hh + 4 * lr;
    const float4 qv = ld4g(g.q + (int64_t)row.tk * C + n), kv = ld4g(g.kv + (int64_t)row.tk * 2 * C + n),
                 vv = ld4g(g.kv + (int64_t)row.tk * 2 * C + C + n);
    *reinterpret_cast<float4*>(&sm[wave][0][li][16 * hh + 4 * lr]) = row.ok ? qv : f4zero();
    *reinterpret_cast<float4*>(&sm[wave][1][li][16 * hh + 4 * lr]) = row.ok ? kv : f4zero();
    *reinterpret_cast<float4*>(&sm[wave][2][li][16 * hh + 4 * lr]) = row.ok ? vv : f4zero();
  }
  {
    float4 d[NF], xr[NF], gd[NF];
    load_raw<C, BF16>(g.dx + (int64_t)row.tk * C, lr, row.ok, d);
    load_raw<C, BF16>(g.x1 + (int64_t)row.tk * C, lr, row.ok, xr);
    const float mu = row.ok ? g.stats[2 * T + row.tk] : 0.f, rs = row.ok ? g.stats[3 * T + row.tk] : 0.f;
    ln_bwd_rows<C, BF16>(d, xr, gd, lr, g.ln2_g, mu, rs);
    if (g.ln2_part) ln_partials<C, BF16>(d, xr, head % heads, heads, li, lr, g.ln2_part + (int64_t)tile * 2 * C);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int c = fcol<BF16>(i, lr);
      const float4 dy = ld4g(g.dy + (int64_t)row.tk * C + c);
      gd[i] = row.ok ? make_float4(dy.x + gd[i].x, dy.y + gd[i].y, dy.z + gd[i].z, dy.w + gd[i].w) : f4zero();
      if (head == 0 && row.ok) {
        st4g(g.dx1 + (int64_t)row.tk * C + c, gd[i]);
        if (g.dx1_copy) st4g(g.dx1_copy + (int64_t)row.tk * C + c, gd[i]);
      }
    }
    pack<C, BF16>(gd, b);
  }
  const float s1 = row_scale(a.geo, g.s1, row);
#pragma unroll
  for (int hh = 0; hh < HT; ++hh) {
    if (hh + 1 < HT) load_w<C, BF16>(wpt, head * HD + 16 * (hh + 1), C, 0, li, lr, fa[(hh + 1) & 1]);
    const float4 v = mma<C, BF16>(fa[hh & 1], b);
    *reinterpret_cast<float4*>(&sm[wave][3][li][16 * hh + 4 * lr]) = make_float4(s1 * v.x, s1 * v.y, s1 * v.z, s1 * v.w);
  }
  __syncthreads();
  // attention backward: lane (li, lr) = attention row li (window li / 8, row i = li % 8) x channel quarter lr of the head: the score /
  // dP dot products are partial sums over HD / 4 channels that meet across the four lane groups; every lane owns HD / 4 channels of
  // dq, dk, dv.  (All 64 lanes work: with one lane per row, 16 lanes ran whole head rows -- 25 us at 4^3, 62 us per launch at the
  // large model's 10 x 10 x 8 stage.)
  if (BF16 && a.attn_mfma) {
    // bf16 mode (round 5): the adjoint of the wave's unit on the matrix cores (attn_fp8.h::attn16_bwd_bf16) -- a wave reads only
    // its own q | k | v | do rows: no second workgroup barrier, no P / dS exchange
    float4 dqv[HT], dkv[HT], dvv[HT];
    attn16_bwd_bf16<HD>(&sm[wave][0][0][0], &sm[wave][1][0][0], &sm[wave][2][0][0], QS, &sm[wave][3][0][0], QS, a.scale, dqv, dkv, dvv);
    if (row.ok) {
#pragma unroll
      for (int cb = 0; cb < HT; ++cb) {
        const int n = head * HD + 16 * cb + 4 * lr;
        st4g(g.dq + (int64_t)row.tk * C + n, dqv[cb]);
        st4g(g.dkv + (int64_t)row.tk * 2 * C + n, dkv[cb]);
        st4g(g.dkv + (int64_t)row.tk * 2 * C + C + n, dvv[cb]);
      }
    }
    return;                                             // (workgroup-uniform: every wave takes this branch)
  }
  constexpr int HP = HD / 4;
  const int i = li & 7, r0 = li & 8, c0 = lr * HP;
  float dq[HP];
  {
    float qr[HP], dor[HP];
#pragma unroll
    for (int d = 0; d < HP; ++d) { qr[d] = sm[wave][0][li][c0 + d] * a.scale; dor[d] = sm[wave][3][li][c0 + d]; }
    float p[8], dp[8], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float sacc = 0.f, dacc = 0.f;
#pragma unroll
      for (int d = 0; d < HP; ++d) { sacc += qr[d] * sm[wave][1][r0 + j][c0 + d]; dacc += dor[d] * sm[wave][2][r0 + j][c0 + d]; }
      sacc += __shfl_xor(sacc, 16, 64); dacc += __shfl_xor(dacc, 16, 64);
      sacc += __shfl_xor(sacc, 32, 64); dacc += __shfl_xor(dacc, 32, 64);
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
      if (lr == (j >> 1)) {                               // (the four lanes of a row hold identical rows: each stores its share)
        ps[wave][li][j] = p[j];
        ps[wave][li][8 + j] = ds;
      }
      const float dss = ds * a.scale;
#pragma unroll
      for (int d = 0; d < HP; ++d) dq[d] += dss * sm[wave][1][r0 + j][c0 + d];
    }
  }
  __syncthreads();
  if (row.ok) {                                                       // as key / value row j = i: column i of P and dS of the window
    float dk[HP], dv[HP];
#pragma unroll
    for (int d = 0; d < HP; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float pm = ps[wave][r0 + m][i], dsm = ps[wave][r0 + m][8 + i] * a.scale;
#pragma unroll
      for (int d = 0; d < HP; ++d) { dk[d] += dsm * sm[wave][0][r0 + m][c0 + d]; dv[d] += pm * sm[wave][3][r0 + m][c0 + d]; }
    }
#pragma unroll
    for (int d = 0; d < HP; d += 4) {
      st4g(g.dq + (int64_t)row.tk * C + head * HD + c0 + d, make_float4(dq[d], dq[d + 1], dq[d + 2], dq[d + 3]));
      st4g(g.dkv + (int64_t)row.tk * 2 * C + head * HD + c0 + d, make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]));
      st4g(g.dkv + (int64_t)row.tk * 2 * C + C + head * HD + c0 + d, make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]));
    }
  }
}

// B4: workgroup = (tile, 16 features), 3 waves = the K thirds [dq | dk | dv].  self: dxn = dq Wq + dkv Wkv -> g.dx (pre-LN1);
// cross: g.dx = dq Wq (pre-LN1 gradient of the q path), g.dxs = dkv Wkv (gradient of the sampled K/V source)
template <int C, bool BF16>
__global__ void __launch_bounds__(192) b4_kernel(const BwdArgs a) {
  constexpr int NF = C / 16, NCB = C / 16;
  using WT = typename WSel<BF16>::T;
  __shared__ float4 red[2][64];
  int grp, idx;
  decode(a.G, grp, idx);
  if (idx >= a.tiles * NCB) return;
  const int tile = idx / NCB, n0 = (idx % NCB) * 16, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const micf_block_bwd_group& g = a.g[grp];
  const Row row = tile_row(a.geo, tile, li);
  Frag<C, BF16> fa, b;
  if (wave == 0) load_w<C, BF16>(static_cast<const WT*>(g.wqt), n0, C, 0, li, lr, fa);
  else load_w<C, BF16>(static_cast<const WT*>(g.wkvt), n0, 2 * C, (wave - 1) * C, li, lr, fa);
  {
    float4 raw[NF];
    if (wave == 0) load_raw<C, BF16>(g.dq + (int64_t)row.tk * C, lr, row.ok, raw);
    else load_raw<C, BF16>(g.dkv + (int64_t)row.tk * 2 * C + (wave - 1) * C, lr, row.ok, raw);
    pack<C, BF16>(raw, b);
  }
  float4 v = mma<C, BF16>(fa, b);
  if (wave) red[wave - 1][lane] = v;
  __syncthreads();
  if (wave == 0 && row.ok) {
    const float4 p1 = red[0][lane], p2 = red[1][lane];
    const float4 kvp = make_float4(p1.x + p2.x, p1.y + p2.y, p1.z + p2.z, p1.w + p2.w);
    const int n = n0 + 4 * lr;
    if (g.dxs) {
      st4g(g.dx + (int64_t)row.tk * C + n, v);
      st4g(g.dxs + (int64_t)row.tk * C + n, kvp);
    } else {
      st4g(g.dx + (int64_t)row.tk * C + n, make_float4(v.x + kvp.x, v.y + kvp.y, v.z + kvp.z, v.w + kvp.w));
    }
  }
}

// B5 (self): workgroup = tile; every wave holds the tile's rows, wave w finishes the float4 slots i % 4 == w.
// dx = dx1 + LN1'(dxn) in place on g.dx
template <int C, bool BF16>
__global__ void __launch_bounds__(256) b5_kernel(const BwdArgs a) {
  constexpr int NF = C / 16;
  int grp, idx;
  decode(a.G, grp, idx);
  if (idx >= a.tiles) return;
  const int tile = idx, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const micf_block_bwd_group& g = a.g[grp];
  if (g.dxs) return;                                                  // (uniform per workgroup)
  const int64_t T = a.geo.T;
  const Row row = tile_row(a.geo, tile, li);
  float4 d[NF], xr[NF], gd[NF];
  load_raw<C, BF16>(g.dx + (int64_t)row.tk * C, lr, row.ok, d);
  load_raw<C, BF16>(g.x + (int64_t)row.tk * C, lr, row.ok, xr);
  const float mu = row.ok ? g.stats[row.tk] : 0.f, rs = row.ok ? g.stats[T + row.tk] : 0.f;
  __syncthreads();                                                    // all four waves hold dxn before anyone overwrites it
  ln_bwd_rows<C, BF16>(d, xr, gd, lr, g.ln1_g, mu, rs);
  if (g.ln1_part) ln_partials<C, BF16>(d, xr, wave, 4, li, lr, g.ln1_part + (int64_t)tile * 2 * C);
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    if ((i & 3) != wave || !row.ok) continue;
    const int c = fcol<BF16>(i, lr);
    const float4 x1 = ld4g(g.dx1 + (int64_t)row.tk * C + c);
    st4g(g.dx + (int64_t)row.tk * C + c, make_float4(x1.x + gd[i].x, x1.y + gd[i].y, x1.z + gd[i].z, x1.w + gd[i].w));
  }
}

template <int C, int HD, bool BF16>
static int launch_fwd(const FwdArgs& a, hipStream_t s) {
  constexpr int heads = C / HD;
  static_assert(heads % 4 == 0 && C % 64 == 0, "waves of a workgroup take 4 heads / 4 feature tiles");
  hipLaunchKernelGGL((f1_kernel<C, HD, BF16>), dim3(grid_for(a.G, a.tiles * (heads / 4))), dim3(256), 0, s, a);
  hipLaunchKernelGGL((f2_kernel<C, BF16>), dim3(grid_for(a.G, a.tiles * (C / 64))), dim3(256), 0, s, a);
  hipLaunchKernelGGL((f3_kernel<C, BF16>), dim3(grid_for(a.G, a.tiles * (4 * C / 128))), dim3(256), 0, s, a);
  hipLaunchKernelGGL((f4_kernel<C, BF16>), dim3(grid_for(a.G, a.tiles * (C / 16))), dim3(256), 0, s, a);
  MICF_RETURN_LAUNCH();
}
template <int C, int HD, bool BF16>
static int launch_bwd(const BwdArgs& a, bool any_self, hipStream_t s) {
  constexpr int heads = C / HD;
  hipLaunchKernelGGL((b1_kernel<C, BF16>), dim3(grid_for(a.G, a.tiles * (4 * C / 128))), dim3(256), 0, s, a);
  hipLaunchKernelGGL((b2_kernel<C, BF16>), dim3(grid_for(a.G, a.tiles * (C / 16))), dim3(256), 0, s, a);
  hipLaunchKernelGGL((b3_kernel<C, HD, BF16>), dim3(grid_for(a.G, a.tiles * (heads / 4))), dim3(256), 0, s, a);
  hipLaunchKernelGGL((b4_kernel<C, BF16>), dim3(grid_for(a.G, a.tiles * (C / 16))), dim3(192), 0, s, a);
  if (any_self) hipLaunchKernelGGL((b5_kernel<C, BF16>), dim3(grid_for(a.G, a.tiles)), dim3(256), 0, s, a);
  MICF_RETURN_LAUNCH();
}

}  // namespace wide

// tokens per tile of the few-token path for this shape (0 = not handled here)
int block_wide_tile_tokens(int C, int hd) {
  // head_dim 16 = the base model's 4^3 stage (64 tokens per sample: 8 tiles).  head_dim 32 at C = 384 is the large model's
  // 10 x 10 x 8 stage -- 100 tiles per launch, enough for the tile-per-workgroup kernels (block_fwd.hip / block_bwd.hip), which
  // keep a block's intermediates in LDS instead of passing them through HBM between nine launches
  if (C == 384 && hd == 16) return 16;
  return 0;
}

#define MICF_WIDE_DISPATCH(FN, ...)                                                             \
  do {                                                                                          \
    if (C == 384 && hd == 16) return bf ? FN<384, 16, true>(__VA_ARGS__) : FN<384, 16, false>(__VA_ARGS__); \
    if (C == 384 && hd == 32) return bf ? FN<384, 32, true>(__VA_ARGS__) : FN<384, 32, false>(__VA_ARGS__); \
  } while (0)

int block_fwd_wide(const micf_block_fwd_group* groups, int ngroups, int B, int D, int H, int W, int C, int heads, float eps,
                   float scale, int dtype, hipStream_t s) {
  wide::FwdArgs a;
  for (int i = 0; i < 2; ++i) a.g[i] = groups[i < ngroups ? i : 0];
  a.geo = make_tile_geo(B, D, H, W);
  a.G = ngroups; a.eps = eps; a.scale = scale;
  a.tiles = (a.geo.nwin + 1) / 2;
  a.att8 = dtype == MICF_DTYPE_BF16_ATTN_FP8 ? 2 : (dtype == MICF_DTYPE_BF16 ? 1 : 0);
  const int hd = C / heads;
  const bool bf = dtype == MICF_DTYPE_BF16 || dtype == MICF_DTYPE_BF16_ATTN_FP8;
  MICF_WIDE_DISPATCH(wide::launch_fwd, a, s);
  return MICF_EUNSUPPORTED;
}

int block_bwd_wide(const micf_block_bwd_group* groups, int ngroups, int B, int D, int H, int W, int C, int heads, float scale,
                   int dtype, hipStream_t s) {
  wide::BwdArgs a;
  bool any_self = false;
  for (int i = 0; i < 2; ++i) {
    a.g[i] = groups[i < ngroups ? i : 0];
    any_self |= a.g[i].dxs == nullptr;
  }
  a.geo = make_tile_geo(B, D, H, W);
  a.G = ngroups; a.scale = scale;
  a.tiles = (a.geo.nwin + 1) / 2;
  const int hd = C / heads;
  const bool bf = dtype == MICF_DTYPE_BF16;
  a.attn_mfma = bf ? 1 : 0;
  MICF_WIDE_DISPATCH(wide::launch_bwd, a, any_self, s);
  return MICF_EUNSUPPORTED;
}

}  // namespace micf
