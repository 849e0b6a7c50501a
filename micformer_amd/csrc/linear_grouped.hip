// linear_grouped.hip -- the weight gradients of MANY nn.Linear layers in one launch.
// Reference ops replaced: the dW / dbias halves of autograd's backward for F.linear in q/kv/proj (MS.py:188-201, 246-259)
// and Mlp fc1/fc2 (MS.py:28-34), for all layers of a training step whose token count is small.
//
// Why: nobody on the backward chain waits for a weight gradient -- only the optimizer does.  At the 8^3 / 4^3 token stages
// each dW is a 10-20 us launch of a few dozen workgroups plus a split-reduction launch, ~400 of them per step.  The host
// queues (input, output-gradient) pairs during backward and hands the list over once; a workgroup of the grouped kernel
// finds its (layer, tile, token split) from a prefix table carried in the kernel arguments, so a few launches fill the chip.
// Layers whose tokens fit one split accumulate into dW directly (each element has one owner: plain read-modify-write, no
// atomics); longer ones store per-split partials in the workspace and a grouped reduction adds them.
#include "common.h"
#include "gemm_dma.h"

namespace micf {

constexpr int kGroupMax = 32;          // layers per launch (kernel-argument budget: 32 * 88 B + table < 4 KiB)
constexpr int kGroupChunk = 1024;      // tokens per split

struct GItem {
  const float* a; const float* dy; const float* scale; float* out; float* dbias; float* dw;     // (a / dy: bf16 in the *_b16 kernel)
  int M, N, K, rps;
  int tiles_i, tiles, splits, direct;
  int ldw, pad_;                       // row stride of dw in floats (>= K: a column block of a wider weight matrix)
  int64_t split_stride;
};
struct GArgs {
  int n;
  int end[kGroupMax];                  // running total of workgroups up to and including item k
  GItem it[kGroupMax];
};

template <bool BF16>
__global__ void __launch_bounds__(256) wgrad_grouped_kernel(const GArgs g) {
  __shared__ __attribute__((aligned(1024))) float Ps[kDmaNS * 64 * kDmaBR];
  __shared__ __attribute__((aligned(1024))) float Qs[kDmaNS * 64 * kDmaBR];
  // XCD-aware order (gemm_dma.h): the tiles of one (layer, token split) read the same input / output-gradient rows
  const int total = g.end[g.n - 1];
  const int w = xcd_order(blockIdx.x, total);
  if (w >= total) return;
  int k = 0;
  while (k < g.n - 1 && w >= g.end[k]) ++k;
  const GItem& it = g.it[k];
  const int local = w - (k ? g.end[k - 1] : 0);
  const int split = local / it.tiles, t = local % it.tiles;
  const int bi = t % it.tiles_i, bj = t / it.tiles_i;
  const int i0 = bi * 64, j0 = bj * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lr = lane >> 4;
  const int r_begin = split * kGroupChunk;
  const int r_end = (r_begin + kGroupChunk < it.M) ? r_begin + kGroupChunk : it.M;
  const DmaOperand P{it.a, it.K, it.K}, Q{it.dy, it.N, it.N};
  const bool do_cs = (it.dbias != nullptr) && (bi == 0) && (tid < 64);

  f32x4 tot[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) tot[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ctot = 0.f;
  // one pass per sample segment: the DropPath scale is constant inside a sample, so it multiplies the segment's product
  for (int r = r_begin; r < r_end;) {
    int seg_end = r_end;
    float s = 1.f;
    if (it.scale) {
      const int b = r / it.rps;
      s = it.scale[b];
      const int lim = (b + 1) * it.rps;
      if (lim < seg_end) seg_end = lim;
    }
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    if (r != r_begin) __syncthreads();                       // the rings are reused by the next segment
    const int nslab = (seg_end - r) / kDmaBR;
    dma_tile_loop<true, true, kDmaNS, BF16>(P, Q, i0, j0, r, nslab, Ps, Qs, acc, do_cs, csum);
    // ragged tail of the segment (token counts that are no multiple of 16: the 5 x 5 x 4 grid of the large model's last stage --
    // 100 tokens per sample -- used to send every one of its layers to a per-layer launch with an atomic epilogue, 315 launches
    // and 1.6 ms per step of BASELINE config 4): at most 15 rows, added by the lanes that own the outputs; the operands are
    // rounded as the matrix-core path rounds them
    for (int rr = r + nslab * kDmaBR; rr < seg_end; ++rr) {
      const int jq = j0 + 4 * li + wave;
      float dyv = jq < it.N ? it.dy[(int64_t)rr * it.N + jq] : 0.f;
      if constexpr (BF16) dyv = __uint_as_float(pack_bf16(dyv, 0.f) << 16);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = i0 + 16 * lr + 4 * v;
        if (i >= it.K) continue;
        float4 a4 = *reinterpret_cast<const float4*>(it.a + (int64_t)rr * it.K + i);          // (K % 4 == 0)
        if constexpr (BF16) {
          const unsigned lo = pack_bf16(a4.x, a4.y), hi = pack_bf16(a4.z, a4.w);
          a4 = make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xFFFF0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xFFFF0000u));
        }
        acc[0][v] += dyv * a4.x; acc[1][v] += dyv * a4.y; acc[2][v] += dyv * a4.z; acc[3][v] += dyv * a4.w;
      }
      if (do_cs && j0 + tid < it.N) csum += it.dy[(int64_t)rr * it.N + j0 + tid];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { tot[q][0] += s * acc[q][0]; tot[q][1] += s * acc[q][1]; tot[q][2] += s * acc[q][2]; tot[q][3] += s * acc[q][3]; }
    ctot += s * csum;
    r = seg_end;
  }
  if (do_cs && j0 + tid < it.N) atomicAdd(it.dbias + j0 + tid, ctot);
  const int j = j0 + 4 * li + wave;
  if (j >= it.N) return;
  float* base = it.direct ? it.dw : it.out + (int64_t)split * it.split_stride;
  const int ld = it.direct ? it.ldw : it.K;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int i = i0 + 16 * lr + 4 * v;
    if (i >= it.K) continue;
    float* p = base + (int64_t)j * ld + i;
    float4 o = make_float4(tot[0][v], tot[1][v], tot[2][v], tot[3][v]);
    if (i + 4 <= it.K) {
      if (it.direct) { const float4 old = *reinterpret_cast<const float4*>(p); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
      *reinterpret_cast<float4*>(p) = o;
    } else {
      const float e[4] = {o.x, o.y, o.z, o.w};
      for (int q = 0; q < it.K - i; ++q) p[q] = it.direct ? p[q] + e[q] : e[q];
    }
  }
}


// ---- the same launch for layers whose OPERANDS ARE STORED AS bf16 (MICF_DTYPE_BF16 saves the activations a block's backward and
// its weight gradients re-read as bf16: half the bytes of the operand stream, and the rounding that the fp32 path does at every
// fragment read is done once, at the store).  a [M, K], dy [M, N] bf16 row-major; token steps of 32 (one v_mfma_f32_16x16x32_bf16).
//
// LDS stage of an operand = 32 tokens x 64 features of bf16 = 4 KiB = four 1 KiB pieces (one per wave: tokens 8 w .. 8 w + 7),
// written by one global_load_lds_dwordx4 per wave (16 B = 8 features per lane, lane-linear destination).  The MFMA fragments need
// 8 CONSECUTIVE TOKENS of one feature per lane -- a transposed read: ds_read_b64_tr_b16 (gfx950) hands lane i of a 16-lane group
// column i of the 4 x 16 block whose row j the lanes 4 j .. 4 j + 3 address (8 B each), so two of them per 16 x 32 fragment, no
// conversion and no shuffles.  The destination of a DMA is fixed (slot = lane) but its SOURCE is free, so the piece is stored
// with the 16-byte units of its token rows 2, 3, 6, 7 swapped pairwise (unit u -> u ^ 2) and the pieces 1088 B apart: the 64
// dwords a 32-lane LDS phase reads then fall into 64 different banks (natural order: rows r and r + 2 collide, 4-way with the
// second lane group).
constexpr int kB16Piece = 1024 + 64;                // bytes between the waves' pieces of a stage
constexpr int kB16Stage = 4 * kB16Piece;            // bytes per operand stage
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(unsigned stage_piece, int li, int unit0) {
  // rows (li >> 2) and (li >> 2) + 4 of this lane group's piece, features 16 t .. 16 t + 15 (units 2 t, 2 t + 1)
  const int r = li >> 2, u = unit0 + ((li & 3) >> 1);
  const unsigned a0 = stage_piece + (unsigned)((r * 8 + (u ^ (2 * ((r >> 1) & 1)))) * 16 + (li & 1) * 8);
  const unsigned a1 = a0 + 4 * 8 * 16;              // row + 4: same swizzle ((r + 4) >> 1) & 1 == (r >> 1) & 1
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ void dma16_b16(const uint16_t* gsrc, unsigned lds_byte) {
  dma16(reinterpret_cast<const float*>(gsrc), lds_byte);
}

__global__ void __launch_bounds__(256) wgrad_grouped_b16_kernel(const GArgs g) {
  __shared__ __attribute__((aligned(1024))) char Ps[kDmaNS * kB16Stage];
  __shared__ __attribute__((aligned(1024))) char Qs[kDmaNS * kB16Stage];
  const int total = g.end[g.n - 1];
  const int w = xcd_order(blockIdx.x, total);
  if (w >= total) return;
  int k = 0;
  while (k < g.n - 1 && w >= g.end[k]) ++k;
  const GItem& it = g.it[k];
  const int local = w - (k ? g.end[k - 1] : 0);
  const int split = local / it.tiles, t = local % it.tiles;
  const int bi = t % it.tiles_i, bj = t / it.tiles_i;
  const int i0 = bi * 64, j0 = bj * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lr = lane >> 4;
  const int r_begin = split * kGroupChunk;
  const int r_end = (r_begin + kGroupChunk < it.M) ? r_begin + kGroupChunk : it.M;
  const uint16_t* A = reinterpret_cast<const uint16_t*>(it.a);
  const uint16_t* DY = reinterpret_cast<const uint16_t*>(it.dy);
  const bool do_cs = (it.dbias != nullptr) && (bi == 0);

  // this lane's DMA source inside a 32-token step: slot = lane of piece `wave` -> token row 8 wave + (lane >> 3), unit (lane & 7)
  // with the swizzle undone; features beyond the operand are clamped to its last unit (their products are never stored)
  const int srow = lane >> 3, sunit = (lane & 7) ^ (2 * ((srow >> 1) & 1));
  int xa = i0 + 8 * sunit; if (xa > it.K - 8) xa = it.K - 8;
  int xd = j0 + 8 * sunit; if (xd > it.N - 8) xd = it.N - 8;
  const unsigned pl = lds_addr(Ps) + wave * kB16Piece, ql = lds_addr(Qs) + wave * kB16Piece;
  const unsigned pf = lds_addr(Ps) + lr * kB16Piece, qf = lds_addr(Qs) + lr * kB16Piece;   // fragment reads: lane group lr <-> piece lr

  f32x4 tot[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) tot[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ctot = 0.f;
  for (int r = r_begin; r < r_end;) {
    int seg_end = r_end;
    float s = 1.f;
    if (it.scale) {
      const int b = r / it.rps;
      s = it.scale[b];
      const int lim = (b + 1) * it.rps;
      if (lim < seg_end) seg_end = lim;
    }
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    if (r != r_begin) __syncthreads();
    const int nstep = (seg_end - r) / 32;
    const uint16_t* psrc = A + (int64_t)(r + 8 * wave + srow) * it.K + xa;
    const uint16_t* qsrc = DY + (int64_t)(r + 8 * wave + srow) * it.N + xd;
    const int64_t pstep = (int64_t)32 * it.K, qstep = (int64_t)32 * it.N;
#pragma unroll
    for (int st = 0; st < kDmaNS - 1; ++st) {
      if (st < nstep) {
        dma16_b16(psrc, pl + st * kB16Stage);
        dma16_b16(qsrc, ql + st * kB16Stage);
        psrc += pstep; qsrc += qstep;
      }
    }
    for (int st = 0; st < nstep; ++st) {
      wait_younger((nstep - 1 - st < kDmaNS - 2) ? nstep - 1 - st : kDmaNS - 2);
      __builtin_amdgcn_s_barrier();
      if (st + kDmaNS - 1 < nstep) {
        const int buf = (st + kDmaNS - 1) % kDmaNS;
        dma16_b16(psrc, pl + buf * kB16Stage);
        dma16_b16(qsrc, ql + buf * kB16Stage);
        psrc += pstep; qsrc += qstep;
      }
      const unsigned po = pf + (st % kDmaNS) * kB16Stage, qo = qf + (st % kDmaNS) * kB16Stage;
      const bf16x8 bb = tr_frag(qo, li, 2 * wave);                   // B[k = 8 lr ..][j = 16 wave + li]
      if (do_cs) {
#pragma unroll
        for (int e = 0; e < 8; ++e) csum += __uint_as_float(((unsigned)(unsigned short)bb[e]) << 16);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(po, li, 2 * q), bb, acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { tot[q][0] += s * acc[q][0]; tot[q][1] += s * acc[q][1]; tot[q][2] += s * acc[q][2]; tot[q][3] += s * acc[q][3]; }
    ctot += s * csum;
    r = seg_end;
  }
  const int j = j0 + 16 * wave + li;                  // output row (layer output feature); D rows = layer input features
  if (do_cs) {                                        // column sum of dy: this lane summed the tokens 8 lr .. 8 lr + 7 of every step
    ctot += __shfl_xor(ctot, 16, 64);
    ctot += __shfl_xor(ctot, 32, 64);
    if (lr == 0 && j < it.N) atomicAdd(it.dbias + j, ctot);
  }
  if (j >= it.N) return;
  float* base = it.direct ? it.dw : it.out + (int64_t)split * it.split_stride;
  const int ld = it.direct ? it.ldw : it.K;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = i0 + 16 * q + 4 * lr;
    if (i >= it.K) continue;
    float* p = base + (int64_t)j * ld + i;            // (K % 8 == 0: whole float4)
    float4 o = make_float4(tot[q][0], tot[q][1], tot[q][2], tot[q][3]);
    if (it.direct) { const float4 old = *reinterpret_cast<const float4*>(p); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
    *reinterpret_cast<float4*>(p) = o;
  }
}

// dw += sum over splits of the workspace partials, for every non-direct item of the group
__global__ void __launch_bounds__(256) wgrad_grouped_reduce_kernel(const GArgs g) {
  const int w = blockIdx.x;
  int k = 0;
  while (k < g.n - 1 && w >= g.end[k]) ++k;
  const GItem& it = g.it[k];
  const int local = w - (k ? g.end[k - 1] : 0);
  const int64_t n4 = (int64_t)it.N * it.K / 4;
  const int64_t e = (int64_t)local * 256 + threadIdx.x;
  if (e >= n4) return;
  float* dst = it.dw + ((4 * e) / it.K) * (int64_t)it.ldw + (4 * e) % it.K;      // (K % 4 == 0: a float4 never straddles rows)
  float4 acc = *reinterpret_cast<const float4*>(dst);
  int s = 0;
  for (; s + 4 <= it.splits; s += 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(it.out + (s + u) * it.split_stride + 4 * e);
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; s < it.splits; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(it.out + s * it.split_stride + 4 * e);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(dst) = acc;
}

static bool item_ok(const micf_wgrad_item& x) {
  if (x.operand_dtype != MICF_DTYPE_F32 && x.operand_dtype != MICF_DTYPE_BF16) return false;
  if (x.operand_dtype == MICF_DTYPE_BF16) {          // bf16 operands: token steps of 32, whole 16-byte units of 8 features
    if (x.M % 32 || x.N % 8 || x.K % 8 || x.N < 8 || x.K < 8) return false;
    if (x.dp_scale && x.rows_per_sample % 32) return false;
  }
  if (!x.a || !x.dy || !x.dw || x.M <= 0 || x.N < 4 || x.K < 4) return false;
  if (x.ldw != 0 && (x.ldw < x.K || x.ldw % 4)) return false;
  if (x.M >= (1LL << 30) || x.N % 4 || x.K % 4) return false;                         // (fp32 operands: any token count, ragged tails)
  if ((reinterpret_cast<uintptr_t>(x.a) | reinterpret_cast<uintptr_t>(x.dy) | reinterpret_cast<uintptr_t>(x.dw)) & 15) return false;
  if (x.dp_scale && (x.rows_per_sample <= 0 || x.M % x.rows_per_sample)) return false;
  return true;
}

static int item_splits(const micf_wgrad_item& x) { return (int)((x.M + kGroupChunk - 1) / kGroupChunk); }

}  // namespace micf

using namespace micf;

extern "C" int64_t micf_linear_bwd_weight_grouped_workspace(const micf_wgrad_item* items, int n) {
  if (!items || n <= 0) return 0;
  int64_t total = 0;
  for (int k = 0; k < n; ++k) {
    if (!item_ok(items[k])) return -1;
    const int sp = item_splits(items[k]);
    if (sp > 1) total += (int64_t)sp * items[k].N * items[k].K;
  }
  return total;
}

extern "C" int micf_linear_bwd_weight_grouped(const micf_wgrad_item* items, int n, float* workspace, int64_t workspace_floats,
                                              int dtype, micf_stream_t stream) {
  if (n < 0 || (n > 0 && !items)) return MICF_EINVAL;
  if (n == 0) return MICF_OK;
  const int64_t need = micf_linear_bwd_weight_grouped_workspace(items, n);
  if (need < 0) return MICF_EUNSUPPORTED;
  if (need > 0 && (!workspace || workspace_floats < need || (reinterpret_cast<uintptr_t>(workspace) & 15))) return MICF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int64_t ws_off = 0;
  // two passes: the layers with fp32 operands, then the layers whose operands were stored as bf16 (another kernel)
  for (int pass = 0; pass < 2; ++pass) {
    int idx[kGroupMax], cnt = 0;
    for (int k = 0; k <= n; ++k) {
      if (k < n && (items[k].operand_dtype == MICF_DTYPE_BF16) == (pass == 1)) idx[cnt++] = k;
      if (cnt == 0 || (cnt < kGroupMax && k < n)) continue;
      GArgs g, rg;
      g.n = cnt;
      rg.n = 0;
      int blocks = 0, rblocks = 0;
      for (int q = 0; q < cnt; ++q) {
        const micf_wgrad_item& x = items[idx[q]];
        GItem& d = g.it[q];
        d.a = x.a; d.dy = x.dy; d.scale = x.dp_scale; d.dbias = x.dbias; d.dw = x.dw;
        d.M = (int)x.M; d.N = x.N; d.K = x.K; d.rps = x.dp_scale ? (int)x.rows_per_sample : (int)x.M;
        d.ldw = x.ldw > 0 ? x.ldw : x.K; d.pad_ = 0;
        d.tiles_i = ceil_div(x.K, 64);
        d.tiles = d.tiles_i * ceil_div(x.N, 64);
        d.splits = item_splits(x);
        d.direct = d.splits == 1;
        d.split_stride = (int64_t)x.N * x.K;
        d.out = d.direct ? nullptr : workspace + ws_off;
        if (!d.direct) ws_off += d.splits * d.split_stride;
        blocks += d.tiles * d.splits;
        g.end[q] = blocks;
        if (!d.direct) {
          rg.it[rg.n] = d;
          rblocks += (int)((d.split_stride / 4 + 255) / 256);
          rg.end[rg.n] = rblocks;
          ++rg.n;
        }
      }
      if (pass == 1) hipLaunchKernelGGL(wgrad_grouped_b16_kernel, dim3((blocks + 7) / 8 * 8), dim3(256), 0, s, g);
      else if (dtype == MICF_DTYPE_BF16) hipLaunchKernelGGL(wgrad_grouped_kernel<true>, dim3((blocks + 7) / 8 * 8), dim3(256), 0, s, g);
      else hipLaunchKernelGGL(wgrad_grouped_kernel<false>, dim3((blocks + 7) / 8 * 8), dim3(256), 0, s, g);
      if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
      if (rg.n > 0) {
        hipLaunchKernelGGL(wgrad_grouped_reduce_kernel, dim3(rblocks), dim3(256), 0, s, rg);
        if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
      }
      cnt = 0;
    }
  }
  return MICF_OK;
}
