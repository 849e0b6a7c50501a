// layernorm.hip -- nn.LayerNorm over the channel dim of channels-last token rows (MS.py:308,321,461,468,540,569,987,988),
// one 64-lane wave per row, optional two-source rows (replaces torch.cat + norm2, MS.py:1033-1034).
#include "common.h"

namespace micf {

// rows per workgroup: 32 (4 waves x 8) amortises the dgamma/dbeta flush on big token grids; 4 (one row per wave) keeps
// all 256 CUs busy on the 8^3 / 4^3 stages
static inline int ln_rows_per_block(int64_t rows) { return rows >= 16384 ? 32 : 4; }
constexpr int kLnMaxC = 4096;

__device__ __forceinline__ float ln_fetch(const float* x1, const float* x2, int c1, int c2, int64_t row, int c) {
  return c < c1 ? x1[row * c1 + c] : x2[row * c2 + (c - c1)];
}

__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int c1,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows, int C, float eps, int rpb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c2 = C - c1;
  for (int k = 0; k < rpb / 4; ++k) {
    const int64_t row = (int64_t)blockIdx.x * rpb + k * 4 + wave;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ln_fetch(x1, x2, c1, c2, row, c);
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = ln_fetch(x1, x2, c1, c2, row, c) - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    for (int c = lane; c < C; c += 64)
      y[row * C + c] = (ln_fetch(x1, x2, c1, c2, row, c) - mu) * rs * gamma[c] + beta[c];
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
  }
}

__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* dy, const float* __restrict__ x1,
                                                     const float* __restrict__ x2, int c1, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     float* dx1, float* dx2,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows,
                                                     int C, const float* add, int rpb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [2][C] block partials of dgamma, dbeta
  float* sg = sm;
  float* sb = sm + C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c2 = C - c1;
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  for (int k = 0; k < rpb / 4; ++k) {
    const int64_t row = (int64_t)blockIdx.x * rpb + k * 4 + wave;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    float a = 0.f, b = 0.f;       // sum g*dy, sum g*dy*xhat
    for (int c = lane; c < C; c += 64) {
      const float xh = (ln_fetch(x1, x2, c1, c2, row, c) - mu) * rs;
      const float gd = gamma[c] * dy[row * C + c];
      a += gd; b += gd * xh;
    }
    a = wave_sum(a) / (float)C;
    b = wave_sum(b) / (float)C;
    for (int c = lane; c < C; c += 64) {
      const float xh = (ln_fetch(x1, x2, c1, c2, row, c) - mu) * rs;
      const float d = dy[row * C + c];
      const float v = rs * (gamma[c] * d - a - xh * b);
      float* dst = c < c1 ? dx1 + row * c1 + c : dx2 + row * c2 + (c - c1);
      *dst = add ? add[row * C + c] + v : v;
      atomicAdd(&sg[c], d * xh);
      atomicAdd(&sb[c], d);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    if (dgamma) atomicAdd(dgamma + c, sg[c]);
    if (dbeta) atomicAdd(dbeta + c, sb[c]);
  }
}


// ------------------------------------------------------------------ v2: register-resident rows, 16-byte accesses
// A row of C floats is held by LPR lanes (VPL float4 each), so a wave normalises 64/LPR rows at once with ONE pass over
// HBM (the v1 kernels re-read the row three times through L1 and keep 16 of 64 lanes idle at C = 48).  The backward keeps
// the per-column dgamma / dbeta partials in registers across all rows a lane visits and flushes once per workgroup.
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR, int VPL>
__device__ __forceinline__ void ln_fwd_v2_body(const float* __restrict__ x, const float* __restrict__ x2, int c1,
                                                 const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* __restrict__ y,
                                                 float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int C,
                                                 float eps, int rpb) {
  // (x2: the row is cat[x (c1 columns), x2 (C - c1 columns)], both multiples of 4 -- MS.py:1033-1034; otherwise c1 = C)
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, rg = lane / LPR;
  const int64_t r_end = ((int64_t)(blockIdx.x + 1) * rpb < rows) ? (int64_t)(blockIdx.x + 1) * rpb : rows;
  float4 g[VPL], b[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int c = (sub + k * LPR) * 4;
    g[k] = c < C ? ldg4(gamma + c) : make_float4(0, 0, 0, 0);
    b[k] = c < C ? ldg4(beta + c) : make_float4(0, 0, 0, 0);
  }
  const float invC = 1.0f / (float)C;
  for (int64_t row = (int64_t)blockIdx.x * rpb + wave * RPW + rg; row < r_end; row += 4 * RPW) {
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int c = (sub + k * LPR) * 4;
      v[k] = c < C ? ldg4(c < c1 ? x + row * c1 + c : x2 + row * (C - c1) + (c - c1)) : make_float4(0, 0, 0, 0);
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mu = group_sum<LPR>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int c = (sub + k * LPR) * 4;
      if (c < C) { const float a0 = v[k].x - mu, a1 = v[k].y - mu, a2 = v[k].z - mu, a3 = v[k].w - mu; q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3); }
    }
    const float rs = 1.0f / sqrtf(group_sum<LPR>(q) * invC + eps);
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int c = (sub + k * LPR) * 4;
      if (c < C)
        *reinterpret_cast<float4*>(y + row * C + c) =
            make_float4((v[k].x - mu) * rs * g[k].x + b[k].x, (v[k].y - mu) * rs * g[k].y + b[k].y,
                        (v[k].z - mu) * rs * g[k].z + b[k].z, (v[k].w - mu) * rs * g[k].w + b[k].w);
    }
    if (sub == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
  }
}

template <int LPR, int VPL>
__device__ __forceinline__ void ln_bwd_v2_body(const float* dy, const float* __restrict__ x, const float* __restrict__ x2, int c1,
                                                 const float* __restrict__ mean,
                                                 const float* __restrict__ rstd, const float* __restrict__ gamma, float* dx, float* dx2,
                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C,
                                                 const float* add, float* __restrict__ partials, int rpb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4 waves][2][C]
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, rg = lane / LPR;
  const int64_t r_end = ((int64_t)(blockIdx.x + 1) * rpb < rows) ? (int64_t)(blockIdx.x + 1) * rpb : rows;
  float4 g[VPL], ag[VPL], ab[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int c = (sub + k * LPR) * 4;
    g[k] = c < C ? ldg4(gamma + c) : make_float4(0, 0, 0, 0);
    ag[k] = make_float4(0, 0, 0, 0);
    ab[k] = make_float4(0, 0, 0, 0);
  }
  const float invC = 1.0f / (float)C;
  for (int64_t row = (int64_t)blockIdx.x * rpb + wave * RPW + rg; row < r_end; row += 4 * RPW) {
    const float mu = mean[row], rs = rstd[row];
    float4 xh[VPL], d[VPL];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int c = (sub + k * LPR) * 4;
      if (c < C) {
        const float4 v = ldg4(c < c1 ? x + row * c1 + c : x2 + row * (C - c1) + (c - c1));
        d[k] = ldg4(dy + row * C + c);
        xh[k] = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
        const float g0 = g[k].x * d[k].x, g1 = g[k].y * d[k].y, g2 = g[k].z * d[k].z, g3 = g[k].w * d[k].w;
        sa += (g0 + g1) + (g2 + g3);
        sb += (g0 * xh[k].x + g1 * xh[k].y) + (g2 * xh[k].z + g3 * xh[k].w);
        ag[k].x += d[k].x * xh[k].x; ag[k].y += d[k].y * xh[k].y; ag[k].z += d[k].z * xh[k].z; ag[k].w += d[k].w * xh[k].w;
        ab[k].x += d[k].x; ab[k].y += d[k].y; ab[k].z += d[k].z; ab[k].w += d[k].w;
      } else { xh[k] = make_float4(0, 0, 0, 0); d[k] = make_float4(0, 0, 0, 0); }
    }
    const float A = group_sum<LPR>(sa) * invC, Bv = group_sum<LPR>(sb) * invC;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int c = (sub + k * LPR) * 4;
      if (c < C) {
        float4 o = make_float4(rs * (g[k].x * d[k].x - A - xh[k].x * Bv), rs * (g[k].y * d[k].y - A - xh[k].y * Bv),
                               rs * (g[k].z * d[k].z - A - xh[k].z * Bv), rs * (g[k].w * d[k].w - A - xh[k].w * Bv));
        if (add) { const float4 a = ldg4(add + row * C + c); o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
        *reinterpret_cast<float4*>(c < c1 ? dx + row * c1 + c : dx2 + row * (C - c1) + (c - c1)) = o;
      }
    }
  }
  // Column sums of the workgroup without LDS atomics (they retire one wave instruction per ~85 cycles and the 16 row groups of a
  // workgroup hit the same 2C addresses: 2048 conflicting atomics per workgroup were most of this kernel): the row groups of a wave
  // meet in shuffles, every wave writes ONE plain [2C] row, thread c adds the four rows.
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      ag[k].x += __shfl_xor(ag[k].x, o, 64); ag[k].y += __shfl_xor(ag[k].y, o, 64); ag[k].z += __shfl_xor(ag[k].z, o, 64); ag[k].w += __shfl_xor(ag[k].w, o, 64);
      ab[k].x += __shfl_xor(ab[k].x, o, 64); ab[k].y += __shfl_xor(ab[k].y, o, 64); ab[k].z += __shfl_xor(ab[k].z, o, 64); ab[k].w += __shfl_xor(ab[k].w, o, 64);
    }
    const int c = (sub + k * LPR) * 4;
    if (rg == 0 && c < C) {
      *reinterpret_cast<float4*>(sm + wave * 2 * C + c) = ag[k];
      *reinterpret_cast<float4*>(sm + wave * 2 * C + C + c) = ab[k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) sm[c] = (sm[c] + sm[2 * C + c]) + (sm[4 * C + c] + sm[6 * C + c]);
  __syncthreads();
  if (partials) {     // [block][2C]: summed later by micf_layernorm_bwd_finish -- hundreds of workgroups adding into the same 2C
                      // addresses with device-scope atomics serialise, and nobody on the backward chain needs dgamma / dbeta
    for (int c = threadIdx.x; c < 2 * C; c += 256) partials[(int64_t)blockIdx.x * 2 * C + c] = sm[c];
    return;
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    if (dgamma) atomicAdd(dgamma + c, sm[c]);
    if (dbeta) atomicAdd(dbeta + c, sm[C + c]);
  }
}

// grouped finish: dgamma[c] += sum_b partials[b][c], dbeta[c] += sum_b partials[b][C + c] for up to kLnFinishMax LayerNorms
constexpr int kLnFinishMax = 64;
struct LnFinishArgs {
  int n;
  int end[kLnFinishMax];                                   // running total of workgroups (one per 256 columns of [2C])
  const float* partials[kLnFinishMax]; float* dgamma[kLnFinishMax]; float* dbeta[kLnFinishMax];
  int blocks[kLnFinishMax]; int C[kLnFinishMax]; int nsplit[kLnFinishMax];
};
__global__ void __launch_bounds__(256) ln_finish_kernel(const LnFinishArgs a) {
  const int w = blockIdx.x;
  int k = 0;
  while (k < a.n - 1 && w >= a.end[k]) ++k;
  const int local = w - (k ? a.end[k - 1] : 0);
  const int C = a.C[k], nb = a.blocks[k], nsplit = a.nsplit[k];
  const int colblocks = (2 * C + 63) >> 6;
  // 64 columns x 4 slices per workgroup; long partial lists (the fused block kernels write one row per 16-32 tokens) are
  // split over `nsplit` workgroups, each row read by exactly one (workgroup, slice)
  const int c = (local % colblocks) * 64 + (threadIdx.x & 63), slice = (local / colblocks) * 4 + (threadIdx.x >> 6);
  const int stride = 4 * nsplit;
  __shared__ float red[256];
  float acc = 0.f;
  if (c < 2 * C) {
    const float* p = a.partials[k] + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = slice;
    for (; b + 3 * stride < nb; b += 4 * stride) {           // four independent loads in flight
      a0 += p[(int64_t)b * 2 * C]; a1 += p[(int64_t)(b + stride) * 2 * C];
      a2 += p[(int64_t)(b + 2 * stride) * 2 * C]; a3 += p[(int64_t)(b + 3 * stride) * 2 * C];
    }
    for (; b < nb; b += stride) a0 += p[(int64_t)b * 2 * C];
    acc = (a0 + a1) + (a2 + a3);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if ((threadIdx.x >> 6) == 0 && c < 2 * C) {
    acc = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
    // atomic: the splits of one item, and modules shared by the two modalities (swin.norm, PatchMerging / PatchExpand norms)
    // that appear as two items with the same destination in one launch
    if (c < C) { if (a.dgamma[k]) atomicAdd(a.dgamma[k] + c, acc); }
    else if (a.dbeta[k]) atomicAdd(a.dbeta[k] + (c - C), acc);
  }
}

// pick (LPR, VPL) for C (C % 4 == 0): smallest lane group that covers C/4 vectors, up to 8 vectors per lane
static bool ln_v2_shape(int C, int& lpr, int& vpl) {
  if (C % 4) return false;
  const int nv = C / 4;
  lpr = nv <= 16 ? 16 : (nv <= 32 ? 32 : 64);
  vpl = (nv + lpr - 1) / lpr;
  return vpl <= 8;
}
// workgroups of the vector backward (= rows of its partial gain / bias sums).  (2048 measured: the 131072-row launch 84 -> 58 us, the
// 65536-row one 37 -> 62 us, the step the same: 512 stays.)
constexpr int kLnBwdBlocks = 512;
static int ln_v2_rpb(int64_t rows, int lpr, int max_blocks) {
  const int step = 4 * (64 / lpr);
  int64_t rpb = (rows + max_blocks - 1) / max_blocks;
  rpb = (rpb + step - 1) / step * step;
  return (int)(rpb < step ? step : rpb);
}

// launch form: up to two independent LayerNorms of the same shape per launch (blockIdx.y): the two modalities of a pair
struct LnFwdSet { const float *x, *gamma, *beta; float *y, *mean, *rstd; const float* x2; int c1; };      // c1 = C: no x2
struct LnBwdSet { const float *dy, *x, *mean, *rstd, *gamma; float *dx, *dgamma, *dbeta; const float* add; float* partials;
                  const float* x2; float* dx2; int c1; };
struct LnFwdSets { LnFwdSet s[2]; float4* zero; int64_t zero4; };   // zero: optional side job (clears zero4 float4s)
struct LnBwdSets { LnBwdSet s[2]; };
template <int LPR, int VPL>
__global__ void __launch_bounds__(256) ln_fwd_v2(const LnFwdSets p, int64_t rows, int C, float eps, int rpb) {
  const LnFwdSet& q = p.s[blockIdx.y];
  if (p.zero && blockIdx.y == 0)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.zero4; i += (int64_t)gridDim.x * 256) p.zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  ln_fwd_v2_body<LPR, VPL>(q.x, q.x2, q.c1, q.gamma, q.beta, q.y, q.mean, q.rstd, rows, C, eps, rpb);
}
template <int LPR, int VPL>
__global__ void __launch_bounds__(256) ln_bwd_v2(const LnBwdSets p, int64_t rows, int C, int rpb) {
  const LnBwdSet& q = p.s[blockIdx.y];
  ln_bwd_v2_body<LPR, VPL>(q.dy, q.x, q.x2, q.c1, q.mean, q.rstd, q.gamma, q.dx, q.dx2, q.dgamma, q.dbeta, rows, C, q.add, q.partials, rpb);
}

#define MICF_LN_DISPATCH(KERNEL, SMEM, MAXB, NG, ...)                                                    \
  do {                                                                                                   \
    const int rpb = ln_v2_rpb(rows, lpr, MAXB);                                                          \
    const dim3 grid(ceil_div(rows, rpb), NG);                                                            \
    hipStream_t s_ = (hipStream_t)stream;                                                                \
    if (lpr == 16 && vpl == 1) hipLaunchKernelGGL((KERNEL<16, 1>), grid, dim3(256), SMEM, s_, __VA_ARGS__, rpb); \
    else if (lpr == 32 && vpl == 1) hipLaunchKernelGGL((KERNEL<32, 1>), grid, dim3(256), SMEM, s_, __VA_ARGS__, rpb); \
    else if (vpl == 1) hipLaunchKernelGGL((KERNEL<64, 1>), grid, dim3(256), SMEM, s_, __VA_ARGS__, rpb); \
    else if (vpl == 2) hipLaunchKernelGGL((KERNEL<64, 2>), grid, dim3(256), SMEM, s_, __VA_ARGS__, rpb); \
    else if (vpl <= 4) hipLaunchKernelGGL((KERNEL<64, 4>), grid, dim3(256), SMEM, s_, __VA_ARGS__, rpb); \
    else hipLaunchKernelGGL((KERNEL<64, 8>), grid, dim3(256), SMEM, s_, __VA_ARGS__, rpb);               \
  } while (0)

}  // namespace micf
using namespace micf;

extern "C" int micf_layernorm_fwd(const float* x1, const float* x2, int c1, const float* gamma, const float* beta,
                                  float* y, float* mean, float* rstd, int64_t rows, int C, float eps,
                                  micf_stream_t stream) {
  if (!x1 || !gamma || !beta || !y || rows < 0 || C <= 0 || c1 <= 0 || c1 > C || (c1 < C && !x2)) return MICF_EINVAL;
  if (rows == 0) return MICF_OK;
  int lpr, vpl;
  const bool cat_ok = c1 == C || (x2 && !(c1 & 3) && !((C - c1) & 3) && aligned16(x2));
  if (cat_ok && ln_v2_shape(C, lpr, vpl) && aligned16(x1) && aligned16(y) && aligned16(gamma) && aligned16(beta)) {
    LnFwdSets p;
    p.s[0] = p.s[1] = LnFwdSet{x1, gamma, beta, y, mean, rstd, c1 == C ? x1 : x2, c1};
    p.zero = nullptr; p.zero4 = 0;
    MICF_LN_DISPATCH(ln_fwd_v2, 0, 2048, 1, p, rows, C, eps);
    MICF_RETURN_LAUNCH();
  }
  const int rpb = ln_rows_per_block(rows);
  const int blocks = ceil_div(rows, rpb);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x1, x2 ? x2 : x1, c1, gamma, beta, y,
                     mean, rstd, rows, C, eps, rpb);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_layernorm_bwd_partial_rows(int64_t rows, int C, int c1) {
  int lpr, vpl;
  if (rows <= 0 || C <= 0 || c1 <= 0 || c1 > C || (c1 & 3) || ((C - c1) & 3) || C > kLnMaxC || !ln_v2_shape(C, lpr, vpl)) return 0;
  return ceil_div(rows, ln_v2_rpb(rows, lpr, kLnBwdBlocks));
}

extern "C" int micf_layernorm_bwd_finish(const micf_ln_finish_item* items, int n, micf_stream_t stream) {
  if (n < 0 || (n > 0 && !items)) return MICF_EINVAL;
  for (int first = 0; first < n; first += kLnFinishMax) {
    const int cnt = (n - first < kLnFinishMax) ? n - first : kLnFinishMax;
    LnFinishArgs a;
    a.n = cnt;
    int blocks = 0;
    for (int k = 0; k < cnt; ++k) {
      const micf_ln_finish_item& it = items[first + k];
      if (!it.partials || it.blocks <= 0 || it.C <= 0) return MICF_EINVAL;
      a.partials[k] = it.partials; a.dgamma[k] = it.dgamma; a.dbeta[k] = it.dbeta; a.blocks[k] = it.blocks; a.C[k] = it.C;
      int ns = it.blocks / 64;                              // >= 16 rows per slice before another split pays
      a.nsplit[k] = ns < 1 ? 1 : (ns > 32 ? 32 : ns);
      blocks += ceil_div(2 * it.C, 64) * a.nsplit[k];
      a.end[k] = blocks;
    }
    hipLaunchKernelGGL(ln_finish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
  }
  return MICF_OK;
}

extern "C" int micf_layernorm_bwd(const float* dy, const float* x1, const float* x2, int c1, const float* mean,
                                  const float* rstd, const float* gamma, float* dx1, float* dx2, float* dgamma,
                                  float* dbeta, int64_t rows, int C, const float* add, float* partials,
                                  micf_stream_t stream) {
  if (!dy || !x1 || !mean || !rstd || !gamma || !dx1 || rows < 0 || C <= 0 || c1 <= 0 || c1 > C ||
      (c1 < C && (!x2 || !dx2)))
    return MICF_EINVAL;
  if (C > kLnMaxC) return MICF_EUNSUPPORTED;
  if (rows == 0) return MICF_OK;
  int lpr, vpl;
  const bool cat_ok = c1 == C || (x2 && dx2 && !(c1 & 3) && !((C - c1) & 3) && aligned16(x2) && aligned16(dx2));
  if (cat_ok && ln_v2_shape(C, lpr, vpl) && aligned16(x1) && aligned16(dy) && aligned16(dx1) && aligned16(gamma) &&
      (!add || aligned16(add))) {
    LnBwdSets p;
    p.s[0] = p.s[1] = LnBwdSet{dy, x1, mean, rstd, gamma, dx1, dgamma, dbeta, add, partials, c1 == C ? x1 : x2, c1 == C ? dx1 : dx2, c1};
    MICF_LN_DISPATCH(ln_bwd_v2, 8 * C * sizeof(float), kLnBwdBlocks, 1, p, rows, C);
    MICF_RETURN_LAUNCH();
  }
  if (partials) return MICF_EUNSUPPORTED;          // the partial form exists for the vector kernel only (see ..._partial_rows)
  const int rpb = ln_rows_per_block(rows);
  const int blocks = ceil_div(rows, rpb);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, dy, x1,
                     x2 ? x2 : x1, c1, mean, rstd, gamma, dx1, dx2 ? dx2 : dx1, dgamma, dbeta, rows, C, add, rpb);
  MICF_RETURN_LAUNCH();
}

// ---- two LayerNorms of the same shape in ONE launch (the two modalities of a cross pair: LN1 forward / backward of both blocks)
extern "C" int micf_layernorm_fwd_pair(const micf_ln_pair_item* items, int n, int64_t rows, int C, float eps, float* zero,
                                       int64_t zero_floats, micf_stream_t stream) {
  if (!items || n < 1 || n > 2 || rows < 0 || C <= 0 || zero_floats < 0 || (zero_floats > 0 && !zero)) return MICF_EINVAL;
  if (zero_floats > 0 && ((zero_floats & 3) || !aligned16(zero))) return MICF_EINVAL;
  if (rows == 0) return zero_floats ? micf_zero(zero, zero_floats * 4, stream) : MICF_OK;
  int lpr, vpl;
  bool ok = ln_v2_shape(C, lpr, vpl);
  LnFwdSets p;
  for (int i = 0; i < 2; ++i) {
    const micf_ln_pair_item& it = items[i < n ? i : 0];
    if (!it.x || !it.gamma || !it.beta || !it.y) return MICF_EINVAL;
    ok = ok && aligned16(it.x) && aligned16(it.y) && aligned16(it.gamma) && aligned16(it.beta);
    p.s[i] = LnFwdSet{it.x, it.gamma, it.beta, it.y, it.mean, it.rstd, it.x, C};
  }
  if (!ok) {
    for (int i = 0; i < n; ++i) {
      const int rc = micf_layernorm_fwd(items[i].x, nullptr, C, items[i].gamma, items[i].beta, items[i].y, items[i].mean, items[i].rstd, rows, C, eps, stream);
      if (rc != MICF_OK) return rc;
    }
    return zero_floats ? micf_zero(zero, zero_floats * 4, stream) : MICF_OK;
  }
  p.zero = reinterpret_cast<float4*>(zero); p.zero4 = zero_floats / 4;
  MICF_LN_DISPATCH(ln_fwd_v2, 0, 2048, n, p, rows, C, eps);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_layernorm_bwd_pair(const micf_ln_bwd_pair_item* items, int n, int64_t rows, int C, micf_stream_t stream) {
  if (!items || n < 1 || n > 2 || rows < 0 || C <= 0) return MICF_EINVAL;
  if (C > kLnMaxC) return MICF_EUNSUPPORTED;
  if (rows == 0) return MICF_OK;
  int lpr, vpl;
  bool ok = ln_v2_shape(C, lpr, vpl);
  LnBwdSets p;
  for (int i = 0; i < 2; ++i) {
    const micf_ln_bwd_pair_item& it = items[i < n ? i : 0];
    if (!it.dy || !it.x || !it.mean || !it.rstd || !it.gamma || !it.dx) return MICF_EINVAL;
    ok = ok && aligned16(it.x) && aligned16(it.dy) && aligned16(it.dx) && aligned16(it.gamma) && (!it.add || aligned16(it.add));
    p.s[i] = LnBwdSet{it.dy, it.x, it.mean, it.rstd, it.gamma, it.dx, it.dgamma, it.dbeta, it.add, it.partials, it.x, it.dx, C};
  }
  if (!ok) {
    for (int i = 0; i < n; ++i) {
      const micf_ln_bwd_pair_item& it = items[i];
      const int rc = micf_layernorm_bwd(it.dy, it.x, nullptr, C, it.mean, it.rstd, it.gamma, it.dx, nullptr, it.dgamma, it.dbeta, rows, C, it.add,
                                        it.partials, stream);
      if (rc != MICF_OK) return rc;
    }
    return MICF_OK;
  }
  MICF_LN_DISPATCH(ln_bwd_v2, 8 * C * sizeof(float), kLnBwdBlocks, n, p, rows, C);
  MICF_RETURN_LAUNCH();
}
