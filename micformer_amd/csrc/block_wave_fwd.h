// block_wave_fwd.h -- the WAVE-PRIVATE form of block_fwd.hip's launch for the C = 48 stages (base model 32^3: the 8 longest forward
// launches of a step), bf16 mode.  Included by block_fwd.hip (shares BlkFwdArgs).
//
// The tile-per-workgroup kernel splits every GEMM of a 32-token tile over 4 waves and exchanges every intermediate through LDS:
// ~11 barrier-to-barrier phases per tile, each a round trip of LDS writes, a barrier and LDS reads around 2-12 MFMAs per wave --
// the kernel is bound by exactly that bookkeeping (5000 non-MFMA instructions per wave and tile), not by the matrix cores or HBM.
// Here ONE WAVE owns 16 tokens (two 2x2x2 windows) for the whole block and nothing is exchanged at all:
//   * every GEMM runs TRANSPOSED, out^T[feature][token] = W[feature][k] act^T[k][token]: the weights are the A operand, the
//     activations the B operand, and the accumulator quad of lane (li, lr) -- 4 consecutive features of token li -- is, after
//     rounding to bf16, the B-operand piece of the NEXT product (k = features).  LayerNorm, bias, GELU, the residual adds and the
//     softmax work on those quads in registers; the per-token reductions are two lane exchanges (the 4 lane groups of a token).
//   * the A operands -- all five weight matrices of the block, 55 KB as bf16 -- are staged ONCE per workgroup into LDS as ready
//     MFMA fragments (one conflict-free ds_read_b128 / b64 per product), with the output rows of proj / fc1 / fc2 permuted so that a
//     lane's accumulator quads of two neighbouring 16-row blocks are 8 CONSECUTIVE features: 16-byte stores of h / g / xn / xn2,
//     32 consecutive bytes of x / x1 / y per lane, and the natural k order for the product that consumes them.
//   * attention: attn_fp8.h::attn16_bf16's two products per head on the q / k quads as they are; the V^T operand (rows = channels,
//     k = keys) comes from a second, operand-swapped product of the v weights (3 more MFMAs instead of a transpose through LDS).
// No barrier after the weight staging; a workgroup is 8 waves walking 16-token groups, two workgroups per CU (2 x 58 KB of LDS).
#pragma once

#include "block_wave.h"

namespace micf {
namespace wave48 {

constexpr int kTapBatch = 8;    // taps of a piece in flight in the fused sampling (4: measured the same)

// LDS (bytes): fragments, then the fp32 parameter vectors
constexpr int kQ32 = 0;                          // [9][64] x 16 B   q | k | v blocks, k = 0..31
constexpr int kQ16 = kQ32 + 9 * 1024;            // [9][64] x 8 B    ... k = 32..47
constexpr int kP16 = kQ16 + 9 * 512;             // [3][3][64] x 8 B proj: block, head (k = 16 h ..)
constexpr int kF32 = kP16 + 9 * 512;             // [12][64] x 16 B  fc1 blocks, k = 0..31
constexpr int kF16 = kF32 + 12 * 1024;           // [12][64] x 8 B
constexpr int kG32 = kF16 + 12 * 512;            // [3][6][64] x 16 B fc2: block, k chunk
constexpr int kVec = kG32 + 18 * 1024;           // floats: ln1_g ln1_b bq bkv(2C) bp ln2_g ln2_b b2 | b1(HID)
constexpr int kLdsBytes = kVec + (9 * C + HID + 2 * C) * 4;     // (+ gain | bias of the epilogue LayerNorm, micf_block_fwd_group.nln_g)


__device__ __forceinline__ void stage_weights(char* lds, const micf_block_fwd_group& g) {
  const uint16_t* wq = static_cast<const uint16_t*>(g.wq), *wkv = static_cast<const uint16_t*>(g.wkv), *wp = static_cast<const uint16_t*>(g.wp),
                 *w1 = static_cast<const uint16_t*>(g.w1), *w2 = static_cast<const uint16_t*>(g.w2);
  const int tid = threadIdx.x;
  for (int s = tid; s < 9 * 64; s += NTHR) {
    const int j = s >> 6, li = s & 15, lr = (s >> 4) & 3;
    const uint16_t* W = j < 3 ? wq : wkv;
    const int row = 16 * (j < 3 ? j : j - 3) + li;
    *reinterpret_cast<u32x4v*>(lds + kQ32 + s * 16) = *reinterpret_cast<const u32x4v*>(W + k16(row, 8 * lr, 3));
    *reinterpret_cast<u32x2v*>(lds + kQ16 + s * 8) = *reinterpret_cast<const u32x2v*>(W + k16(row, 32 + 4 * lr, 3));
  }
  for (int s = tid; s < 9 * 64; s += NTHR) {
    const int f = s >> 6, j = f / 3, h = f - 3 * j, li = s & 15, lr = (s >> 4) & 3;
    *reinterpret_cast<u32x2v*>(lds + kP16 + s * 8) = *reinterpret_cast<const u32x2v*>(wp + k16(row_p48(j, li), 16 * h + 4 * lr, 3));
  }
  for (int s = tid; s < 12 * 64; s += NTHR) {
    const int j = s >> 6, li = s & 15, lr = (s >> 4) & 3, row = row_p192(j, li);
    *reinterpret_cast<u32x4v*>(lds + kF32 + s * 16) = *reinterpret_cast<const u32x4v*>(w1 + k16(row, 8 * lr, 3));
    *reinterpret_cast<u32x2v*>(lds + kF16 + s * 8) = *reinterpret_cast<const u32x2v*>(w1 + k16(row, 32 + 4 * lr, 3));
  }
  for (int s = tid; s < 18 * 64; s += NTHR) {
    const int f = s >> 6, j = f / 6, kc = f - 6 * j, li = s & 15, lr = (s >> 4) & 3;
    *reinterpret_cast<u32x4v*>(lds + kG32 + s * 16) = *reinterpret_cast<const u32x4v*>(w2 + k16(row_p48(j, li), 32 * kc + 8 * lr, 12));
  }
  float* PV = reinterpret_cast<float*>(lds + kVec);
  const float* const srcs[9] = {g.ln1_g, g.ln1_b, g.bq, g.bkv, g.bp, g.ln2_g, g.ln2_b, g.b2, g.b1};
  const int offs[10] = {0, C, 2 * C, 3 * C, 5 * C, 6 * C, 7 * C, 8 * C, 9 * C, 9 * C + HID};
  for (int e4 = tid; e4 < (9 * C + HID) >> 2; e4 += NTHR) {
    const int e = e4 << 2;
    int k = 0;
#pragma unroll
    for (int j = 1; j < 9; ++j) k += (e >= offs[j]) ? 1 : 0;
    const float* sp = srcs[0];
    int so = offs[0];
#pragma unroll
    for (int j = 1; j < 9; ++j) if (k == j) { sp = srcs[j]; so = offs[j]; }
    *reinterpret_cast<float4*>(PV + e) = ld4g(sp + (e - so));
  }
  if (g.nln_g && tid < 2 * C / 4)
    *reinterpret_cast<float4*>(PV + 9 * C + HID + 4 * tid) = tid < C / 4 ? ld4g(g.nln_g + 4 * tid) : ld4g(g.nln_b + 4 * (tid - C / 4));
}


template <bool SAMP>
__global__ void __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(4, 4))) block_fwd_wave48_kernel(const BlkFwdArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char wlds[];
  const unsigned bid = blockIdx.x;
  int grp, wg, nwg;
  if (a.G == 2) { const int xcd = bid & 7; grp = xcd >> 2; wg = (int)(bid >> 3) * 4 + (xcd & 3); nwg = (int)(gridDim.x >> 3) * 4; }
  else { grp = 0; wg = bid; nwg = gridDim.x; }
  const micf_block_fwd_group& g = a.g[grp];
  stage_weights(wlds, g);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lr = lane >> 4;
  const float* PV = reinterpret_cast<const float*>(wlds + kVec);
  const float *p_ln1g = PV, *p_ln1b = PV + C, *p_bq = PV + 2 * C, *p_bkv = PV + 3 * C, *p_bp = PV + 5 * C, *p_ln2g = PV + 6 * C,
              *p_ln2b = PV + 7 * C, *p_b2 = PV + 8 * C, *p_b1 = PV + 9 * C;
  const int64_t T = a.geo.T;
  const int ngroup16 = (a.geo.nwin + 1) >> 1;
  const bool cross = g.kvsrc != nullptr || (SAMP && g.hid != nullptr);
  const bool save = !(a.debug & 1), save_hg = !(a.debug & 17);      // (the "block_debug" probe flags, as in block_fwd_tile; bit 0 is also the inference form)
  lds_barrier();

  // The inputs of a 16-token group: token ids, DropPath scales, the x rows and (cross) the K/V source rows -- given, or sampled here.
  // Fetched ONE GROUP AHEAD: vector-memory operations retire in order, so a load issued after a group's ~46 stores would wait for
  // every one of them to reach memory (the whole chip stores in the same phase: that drain was 40 % of the kernel).
  struct In { Row12 x, kvr; int64_t tk; float s1v, s2v; bool live0; };
  auto fetch = [&](int gt, In& in) {
    const int win = 2 * gt + (li >> 3);
    const bool live0 = win < a.geo.nwin, live = live0 && save;
    int b = 0, d = 0, hh = 0, w = 0;
    a.geo.coords(live0 ? win : 0, li & 7, b, d, hh, w);
    const int64_t tk = ((int64_t)(b * a.geo.D + d) * a.geo.H + hh) * a.geo.W + w;
    in.tk = tk; in.live0 = live0;
    in.s1v = g.s1 ? g.s1[b] : 1.f; in.s2v = g.s2 ? g.s2[b] : 1.f;
    Row12& x = in.x; Row12& kvr = in.kvr;
    x.load(g.x + tk * C, lr);
    if (!live0) x.zero();
    if (g.kvsrc) {
      kvr.load(g.kvsrc + tk * C, lr);
      if (!live0) kvr.zero();
    }
    if (SAMP && g.hid) {
      // the deformable sampling of this token (block_fwd.hip's prologue; MS.py:360-384, STN.py:9-32): 4 lanes share the 16-wide head
      const float4 hv = ld4g(g.hid + tk * kHid + 4 * lr);
      const float mu = sum4((hv.x + hv.y) + (hv.z + hv.w)) * (1.f / kHid);
      const float d0 = hv.x - mu, d1 = hv.y - mu, d2 = hv.z - mu, d3 = hv.w - mu;
      const float rs = 1.0f / sqrtf(sum4((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.f / kHid) + a.eps);
      const float4 lg = ld4g(g.ln16_g + 4 * lr), lb = ld4g(g.ln16_b + 4 * lr);
      const float g0 = gelu_f(d0 * rs * lg.x + lb.x), g1 = gelu_f(d1 * rs * lg.y + lb.y), g2 = gelu_f(d2 * rs * lg.z + lb.z),
                  g3 = gelu_f(d3 * rs * lg.w + lb.w);
      float fl[3];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const float4 wv = ld4g(g.w1c + ax * kHid + 4 * lr);
        fl[ax] = sum4((wv.x * g0 + wv.y * g1) + (wv.z * g2 + wv.w * g3));
      }
      fl[0] += ((float)d + 0.5f) / (float)a.geo.H * 2.f - 1.f;       // MS.py:335  ref[...,0] /= H_key
      fl[1] += ((float)hh + 0.5f) / (float)a.geo.W * 2.f - 1.f;      // MS.py:334  ref[...,1] /= W_key
      fl[2] += ((float)w + 0.5f) / (float)a.geo.D * 2.f - 1.f;       // MS.py:333  ref[...,2] /= D_key
      if (live && lr < 3) g.flow[tk * 3 + lr] = lr == 0 ? fl[0] : (lr == 1 ? fl[1] : fl[2]);
      const Taps tp = make_taps(d, hh, w, fl, a.geo.D, a.geo.H, a.geo.W);
      uint32_t row0 = (uint32_t)b * (uint32_t)(a.geo.D * a.geo.H * a.geo.W);      // (32-bit byte offsets from the wave-uniform base: one register per tap address)
      int lin[8]; float wgt[8]; bool okq[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int dz = q >> 2, dy = (q >> 1) & 1, dx = q & 1;
        lin[q] = 0;
        okq[q] = live0 && tp.finite && corner(tp, dz, dy, dx, a.geo.D, a.geo.H, a.geo.W, lin[q]);
        const float wx = dx ? tp.cx - tp.x0 : (tp.x0 + 1.f) - tp.cx;
        const float wy = dy ? tp.cy - tp.y0 : (tp.y0 + 1.f) - tp.cy;
        const float wz = dz ? tp.cz - tp.z0 : (tp.z0 + 1.f) - tp.cz;
        wgt[q] = okq[q] ? wx * wy * wz : 0.f;
      }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const int co = pc < 2 ? 8 * lr + 4 * pc : 32 + 4 * lr;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q0 = 0; q0 < 8; q0 += kTapBatch) {              // (all 8 taps of a piece in flight: three dependent batches per token)
          float4 tv[kTapBatch];
#pragma unroll
          for (int q = 0; q < kTapBatch; ++q) tv[q] = ld4g(at32(g.samp_src, ((row0 + (uint32_t)lin[q0 + q]) * C + co) * 4u));
#pragma unroll
          for (int q = 0; q < kTapBatch; ++q) {                  // (an invalid tap contributes exact zeros, whatever the row it re-read holds: selects, no branches)
            const bool ok = okq[q0 + q];
            const float wq = wgt[q0 + q];
            acc.x += (ok ? tv[q].x : 0.f) * wq; acc.y += (ok ? tv[q].y : 0.f) * wq; acc.z += (ok ? tv[q].z : 0.f) * wq; acc.w += (ok ? tv[q].w : 0.f) * wq;
          }
          // (ties the next batch's addresses to this batch's sums: the loads are invariant to the compiler and would otherwise all be
          // issued up front -- 24 rows = 96 registers -- whatever barrier stands between them)
          asm volatile("" : "+v"(row0), "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));
        }
        kvr.v[4 * pc] = acc.x; kvr.v[4 * pc + 1] = acc.y; kvr.v[4 * pc + 2] = acc.z; kvr.v[4 * pc + 3] = acc.w;
      }
    }
  };
  const int gstride = nwg * NWAVE;
  In cur, nxt;
  int gt = wg * NWAVE + wave;
  if (gt < ngroup16) fetch(gt, cur);
  for (; gt < ngroup16; gt += gstride) {
    asm volatile("" ::: "memory");       // (the weight fragments are loop-invariant LDS reads: hoisted, all 69 of them would live in registers)
    if (gt + gstride < ngroup16) fetch(gt + gstride, nxt);
    asm volatile("" ::: "memory");       // (... and the next group's loads stay up here, ahead of this group's stores)
    const Row12& x = cur.x; const Row12& kvr = cur.kvr;
    const int64_t tk = cur.tk;
    const float s1v = cur.s1v, s2v = cur.s2v;
    const bool live0 = cur.live0, live = live0 && save;
    // ---- LayerNorm 1 -> xn (bf16 fragments + HBM)
    bf16x8 xnA, kvA; bf16x4_t xnB, kvB;
    {
      Row12 xn;
      float mu, rs;
      layernorm12(x, p_ln1g, p_ln1b, lr, a.eps, xn, mu, rs);
      xnA = xn.lo(); xnB = xn.hi();
      if (live) {
        if (g.xn) {
          uint16_t* o = reinterpret_cast<uint16_t*>(g.xn) + tk * C;
          st4u(o + 8 * lr, xnA); st2u(o + 32 + 4 * lr, xnB);
        }
        if (lr == 0) { g.stats[tk] = mu; g.stats[T + tk] = rs; }
      }
    }
    if (cross) {
      kvA = kvr.lo(); kvB = kvr.hi();
      if (live && g.kvs16) {
        uint16_t* o = static_cast<uint16_t*>(g.kvs16) + tk * C;
        st4u(o + 8 * lr, kvA); st2u(o + 32 + 4 * lr, kvB);
      }
    } else { kvA = xnA; kvB = xnB; }

    // ---- per head h: q, k, v blocks (+ bias; v also with the operands swapped: tokens on the rows), then attention
    // (attn_fp8.h::attn16_bf16 on register operands): S^T = K Qs^T, softmax over the query's own window, O^T = V^T P^T.  One head
    // at a time: 16 accumulator registers live instead of 48.
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    bf16x4_t of[3];
    {
      const bool valid = (lr >> 1) == (li >> 3);
      uint16_t* qo = reinterpret_cast<uint16_t*>(g.q) + tk * C + 4 * lr;
      uint16_t* ko = reinterpret_cast<uint16_t*>(g.kv) + tk * 2 * C + 4 * lr;
      uint16_t* oo = reinterpret_cast<uint16_t*>(g.o) + tk * C + 4 * lr;
#pragma unroll
      for (int h = 0; h < 3; ++h) {
        f32x4 qkv[3], vsw;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int j = 3 * t + h;
          const bf16x8 wa = frag32(wlds, kQ32, j, lane);
          const bf16x4_t wb = frag16(wlds, kQ16, j, lane);
          const f32x4 acc = mfma48(wa, wb, t == 0 ? xnA : kvA, t == 0 ? xnB : kvB);
          const float4 bs = *reinterpret_cast<const float4*>(p_bq + 16 * j + 4 * lr);
          qkv[t] = f32x4{acc[0] + bs.x, acc[1] + bs.y, acc[2] + bs.z, acc[3] + bs.w};
          if (t == 2) {
            const f32x4 u = mfma48(kvA, kvB, wa, wb);
            const float bv = p_bkv[C + 16 * h + li];
            vsw = f32x4{u[0] + bv, u[1] + bv, u[2] + bv, u[3] + bv};
          }
        }
        const bf16x4_t kf = pack4q(qkv[1]);
        if (live) {
          st2u(qo + 16 * h, pack4q(qkv[0]));
          st2u(ko + 16 * h, kf);
          st2u(ko + C + 16 * h, pack4q(qkv[2]));
        }
        const bf16x4_t qf = pack4_bf16v(qkv[0][0] * a.scale, qkv[0][1] * a.scale, qkv[0][2] * a.scale, qkv[0][3] * a.scale);
        const f32x4 st = mfma16(kf, qf, z4);
        float m = valid ? fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])) : -INFINITY;
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = valid ? expf(st[r] - m) : 0.f;
        float sum = (e[0] + e[1]) + (e[2] + e[3]);
        sum += __shfl_xor(sum, 16, 64);
        const float inv = valid ? 1.0f / sum : 0.f;
        const bf16x4_t pf = pack4_bf16v(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);
        of[h] = pack4q(mfma16(pack4q(vsw), pf, z4));
        if (live) st2u(oo + 16 * h, of[h]);
      }
    }

    // ---- proj (+ bp); x1 = x + s1 * proj; LayerNorm 2
    Row12 x1;
    {
      float bp[12];
      vec12(p_bp, lr, bp);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        f32x4 acc = z4;
#pragma unroll
        for (int h = 0; h < 3; ++h) acc = mfma16(frag16(wlds, kP16, 3 * j + h, lane), of[h], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) x1.v[4 * j + r] = x.v[4 * j + r] + s1v * (acc[r] + bp[4 * j + r]);
      }
      if (!live0) x1.zero();
    }
    bf16x8 n2A; bf16x4_t n2B;
    {
      Row12 xn2;
      float mu, rs;
      layernorm12(x1, p_ln2g, p_ln2b, lr, a.eps, xn2, mu, rs);
      n2A = xn2.lo(); n2B = xn2.hi();
      if (live) {
        x1.store(g.x1 + tk * C, lr);
        uint16_t* o = reinterpret_cast<uint16_t*>(g.xn2) + tk * C;
        st4u(o + 8 * lr, n2A); st2u(o + 32 + 4 * lr, n2B);
        if (lr == 0) { g.stats[2 * T + tk] = mu; g.stats[3 * T + tk] = rs; }
      }
    }

    // ---- MLP: fc1 (+ b1) -> h, GELU -> g (both saved, 8 consecutive features per lane and block pair); each 32-wide chunk of g goes
    // straight into the three fc2 accumulators (k chunk c), so no fragment outlives its chunk
    {
      uint16_t* ho = g.h ? static_cast<uint16_t*>(g.h) + tk * HID + 8 * lr : nullptr;
      uint16_t* go = reinterpret_cast<uint16_t*>(g.g) + tk * HID + 8 * lr;
      f32x4 y3[3] = {z4, z4, z4};
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        float4 hv[2], gv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = 2 * c + u;
          const f32x4 acc = mfma48(frag32(wlds, kF32, j, lane), frag16(wlds, kF16, j, lane), n2A, n2B);
          const float4 bs = *reinterpret_cast<const float4*>(p_b1 + 32 * c + 8 * lr + 4 * u);
          hv[u] = make_float4(acc[0] + bs.x, acc[1] + bs.y, acc[2] + bs.z, acc[3] + bs.w);
          gv[u] = make_float4(gelu_t<true>(hv[u].x), gelu_t<true>(hv[u].y), gelu_t<true>(hv[u].z), gelu_t<true>(hv[u].w));
        }
        const bf16x8 gf = to_bf16x8(gv[0], gv[1]);
        if (live && save_hg) {
          if (ho) st4u(ho + 32 * c, to_bf16x8(hv[0], hv[1]));
          st4u(go + 32 * c, gf);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) y3[j] = mfma32(frag32(wlds, kG32, 6 * j + c, lane), gf, y3[j]);
      }
      float b2[12];
      vec12(p_b2, lr, b2);
      Row12 y;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) y.v[4 * j + r] = x1.v[4 * j + r] + s2v * (y3[j][r] + b2[4 * j + r]);
      if (live0) y.store(g.y + tk * C, lr);
      if (g.nln_g) {
        // epilogue: the NEXT block's LayerNorm of the row (the cross block's norm1, otherwise a launch of its own that re-reads y)
        Row12 z;
        float mu, rs;
        layernorm12(y, PV + 9 * C + HID, PV + 10 * C + HID, lr, a.eps, z, mu, rs);
        if (live0) {
          z.store(g.nln_y + tk * C, lr);
          if (lr == 0) { g.nln_mean[tk] = mu; g.nln_rstd[tk] = rs; }
          if (g.zero16) *reinterpret_cast<float4*>(g.zero16 + tk * 16 + 4 * lr) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    cur = nxt;
  }
}

static int launch_fwd_wave48(const BlkFwdArgs& a, hipStream_t s) {
  const int ngroup16 = (a.geo.nwin + 1) / 2;
  int nwg = (ngroup16 + NWAVE - 1) / NWAVE;
  if (nwg > 256) nwg = 256;                              // two workgroups per CU for each of the two groups: one resident round
  const unsigned grid = a.G == 2 ? (unsigned)((nwg + 3) / 4 * 8) : (unsigned)nwg;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_wave48_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_fwd_wave48_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const bool samp = a.g[0].hid != nullptr || a.g[1].hid != nullptr;
  if (samp) hipLaunchKernelGGL((block_fwd_wave48_kernel<true>), dim3(grid), dim3(NTHR), kLdsBytes, s, a);
  else hipLaunchKernelGGL((block_fwd_wave48_kernel<false>), dim3(grid), dim3(NTHR), kLdsBytes, s, a);
  MICF_RETURN_LAUNCH();
}

}  // namespace wave48
}  // namespace micf
