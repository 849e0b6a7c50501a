// block_wave_bwd.h -- the WAVE-PRIVATE form of block_bwd.hip's launch for the C = 48 stages, bf16 mode (block_wave_fwd.h is the
// forward; included by block_bwd.hip, shares BlkBwdArgs).  One wave owns 32 tokens = one row of the per-tile LayerNorm partial sums
// (micf_block_tile_tokens) as two 16-token groups in sequence; per group the whole adjoint chain stays in registers:
//   dy (+ LN'(pre_d): the producing LayerNorm's backward, block_bwd.hip's prologue) -> dh = s2 (dy W2) GELU'(h) -> dxn2 = dh W1 ->
//   dx1 = dy + LN2'(dxn2) -> do = s1 dx1 Wp -> attention adjoint per head (attn_fp8.h::attn16_bwd_bf16's seven products on register
//   operands) -> dq | dk | dv -> self: dx = dx1 + LN1'(dq Wq + dkv Wkv);  cross: dq Wq and dkv Wkv leave separately.
// Every product runs transposed (weights = A operand from LDS fragments, the gradient quads of 16 tokens = B operand), so the
// accumulator quads of one product are the k pieces of the next; the three channel-major operands of the attention adjoint
// (K^T, Qs^T, dO^T) are the row fragments sent through ONE product with the identity each (exact: bf16 x 1 summed with zeros) instead
// of a transpose through LDS.  The per-tile column sums of the LayerNorm gains / biases: a 16-lane butterfly per value (DPP adds),
// kept by one owner lane per channel over the two groups, written as the tile's row.
#pragma once

#include "block_wave.h"

namespace micf {
namespace wave48 {

// LDS (bytes): the five TRANSPOSED weight matrices as MFMA A-fragments, then the three gain vectors
constexpr int kB2 = 0;                           // fc2^T [192 rows, k = 48]:  [12][64] x 16 B (k 0..31)
constexpr int kB2h = kB2 + 12 * 1024;            //                            [12][64] x 8 B  (k 32..47)
constexpr int kB1 = kB2h + 12 * 512;             // fc1^T [48 rows, k = 192]:  [3][6][64] x 16 B
constexpr int kBP = kB1 + 18 * 1024;             // proj^T [48 rows, k = 48]:  [3][64] x 16 B
constexpr int kBPh = kBP + 3 * 1024;             //                            [3][64] x 8 B
constexpr int kBQ = kBPh + 3 * 512;              // q^T [48 rows, k = 48 in heads]:  [3][3][64] x 8 B
constexpr int kBKV = kBQ + 9 * 512;              // kv^T [48 rows, k = 96 in heads]: [3][6][64] x 8 B
constexpr int kBVec = kBKV + 18 * 512;           // floats: ln1_g | ln2_g | pre_g
constexpr int kBPark = kBVec + 3 * C * 4;          // per wave: a [16 tokens][48 channels] fp32 tile in layout P (wave-private: no barrier)
constexpr int kParkFloats = 4 * (16 * 12 + 8);     // slot of (lane group lr, token li, value e): lr * 200 + li * 12 + e (8 floats of padding per lane group: banks)
constexpr int kBwdLdsBytes = kBPark + NWAVE * kParkFloats * 4;

__device__ __forceinline__ void stage_weights_bwd(char* lds, const micf_block_bwd_group& g) {
  const uint16_t* wqt = static_cast<const uint16_t*>(g.wqt), *wkvt = static_cast<const uint16_t*>(g.wkvt), *wpt = static_cast<const uint16_t*>(g.wpt),
                 *w1t = static_cast<const uint16_t*>(g.w1t), *w2t = static_cast<const uint16_t*>(g.w2t);
  const int tid = threadIdx.x;
  for (int s = tid; s < 12 * 64; s += NTHR) {          // rows = hidden features in layout P: dh leaves in 16-byte pieces
    const int j = s >> 6, li = s & 15, lr = (s >> 4) & 3, row = row_p192(j, li);
    *reinterpret_cast<u32x4v*>(lds + kB2 + s * 16) = *reinterpret_cast<const u32x4v*>(w2t + k16(row, 8 * lr, 3));
    *reinterpret_cast<u32x2v*>(lds + kB2h + s * 8) = *reinterpret_cast<const u32x2v*>(w2t + k16(row, 32 + 4 * lr, 3));
  }
  for (int s = tid; s < 18 * 64; s += NTHR) {
    const int f = s >> 6, j = f / 6, kc = f - 6 * j, li = s & 15, lr = (s >> 4) & 3;
    *reinterpret_cast<u32x4v*>(lds + kB1 + s * 16) = *reinterpret_cast<const u32x4v*>(w1t + k16(row_p48(j, li), 32 * kc + 8 * lr, 12));
  }
  for (int s = tid; s < 3 * 64; s += NTHR) {           // rows = attention-output channels, head by head (natural order)
    const int j = s >> 6, li = s & 15, lr = (s >> 4) & 3, row = 16 * j + li;
    *reinterpret_cast<u32x4v*>(lds + kBP + s * 16) = *reinterpret_cast<const u32x4v*>(wpt + k16(row, 8 * lr, 3));
    *reinterpret_cast<u32x2v*>(lds + kBPh + s * 8) = *reinterpret_cast<const u32x2v*>(wpt + k16(row, 32 + 4 * lr, 3));
  }
  for (int s = tid; s < 9 * 64; s += NTHR) {
    const int f = s >> 6, j = f / 3, h = f - 3 * j, li = s & 15, lr = (s >> 4) & 3;
    *reinterpret_cast<u32x2v*>(lds + kBQ + s * 8) = *reinterpret_cast<const u32x2v*>(wqt + k16(row_p48(j, li), 16 * h + 4 * lr, 3));
  }
  for (int s = tid; s < 18 * 64; s += NTHR) {
    const int f = s >> 6, j = f / 6, h = f - 6 * j, li = s & 15, lr = (s >> 4) & 3;
    *reinterpret_cast<u32x2v*>(lds + kBKV + s * 8) = *reinterpret_cast<const u32x2v*>(wkvt + k16(row_p48(j, li), 16 * h + 4 * lr, 6));
  }
  float* PV = reinterpret_cast<float*>(lds + kBVec);
  for (int e = tid; e < 3 * C; e += NTHR) {
    const float* src = e < C ? g.ln1_g : (e < 2 * C ? g.ln2_g : g.pre_g);
    PV[e] = src ? src[e % C] : 0.f;
  }
}

// Column sums over the 16 tokens of a group: the lanes write their 12 values into the wave's LDS tile, lane c < 48 reads channel c's
// column back (16 reads, conflict-free up to the 2-way overlap of the padding) -- 40 LDS instructions and no live registers, where the
// 16-lane butterflies (4 DPP adds per value and quantity) cost 144 VALU instructions and 30 registers per LayerNorm.
__device__ __forceinline__ void park12(float* tile, int lr, int li, const float (&v)[12]) {
  float* p = tile + lr * 200 + li * 12;
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  *reinterpret_cast<float4*>(p + 8) = make_float4(v[8], v[9], v[10], v[11]);
}
__device__ __forceinline__ void unpark12(const float* tile, int lr, int li, float (&v)[12]) {
  const float* p = tile + lr * 200 + li * 12;
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4), c = *reinterpret_cast<const float4*>(p + 8);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ __forceinline__ float colsum16(const float* tile, int lane) {      // lanes 0..47: channel = lane
  const int c = lane < 48 ? lane : 0;
  const float* p = tile + (c < 32 ? (c >> 3) * 200 + (c & 7) : ((c - 32) >> 2) * 200 + 8 + (c & 3));
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < 16; t += 2) { s0 += p[12 * t]; s1 += p[12 * t + 12]; }
  return s0 + s1;
}

// LayerNorm backward of 16 token rows in layout P (block_bwd.hip::ln_bwd_tile's arithmetic): out = add + rs (g d - mean(g d) -
// xh mean(g d xh)); the column sums of d xh (gain) and d (bias) over the 16 tokens are added to pg / pb of lane c < 48 = channel c.
__device__ __forceinline__ void ln_bwd12(const float (&d)[12], const Row12& xin, float mu, float rs, const float* gam, int lr, int li,
                                         int lane, float* tile, const Row12& add, Row12& out, float& pg, float& pb) {
  float gm[12], dxh[12];
  vec12(gam, lr, gm);
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int e = 0; e < 12; e += 4) {
    float g4[4], xh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { xh[r] = (xin.v[e + r] - mu) * rs; g4[r] = gm[e + r] * d[e + r]; dxh[e + r] = d[e + r] * xh[r]; }
    sa += (g4[0] + g4[1]) + (g4[2] + g4[3]);
    sb += (g4[0] * xh[0] + g4[1] * xh[1]) + (g4[2] * xh[2] + g4[3] * xh[3]);
  }
  const float Am = sum4(sa) * (1.0f / C), Bm = sum4(sb) * (1.0f / C);
  park12(tile, lr, li, dxh);
#pragma unroll
  for (int e = 0; e < 12; ++e) out.v[e] = add.v[e] + rs * (gm[e] * d[e] - Am - ((xin.v[e] - mu) * rs) * Bm);
  pg += colsum16(tile, lane);
  asm volatile("" : "+v"(pg));                         // (evaluated HERE: left alone, the compiler sinks the 16 adds to the end of the tile and spills their operands)
  park12(tile, lr, li, d);
  pb += colsum16(tile, lane);
  asm volatile("" : "+v"(pb));
}

__device__ __forceinline__ void unpack8(const u32x4v& u, float (&o)[8]) {
#pragma unroll
  for (int w = 0; w < 4; ++w) { o[2 * w] = __uint_as_float(u[w] << 16); o[2 * w + 1] = __uint_as_float(u[w] & 0xFFFF0000u); }
}

// CROSS / PRE: both groups of the launch alike (the entry point falls back to the tile kernel otherwise): every path the other kinds
// need is compiled out -- a kernel that carried all three at once spilled 44 registers.
template <bool CROSS, bool PRE>
__global__ void __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(4, 4))) block_bwd_wave48_kernel(const BlkBwdArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char wlds[];
  const unsigned bid = blockIdx.x;
  int grp, wg, nwg;
  if (a.G == 2) { const int xcd = bid & 7; grp = xcd >> 2; wg = (int)(bid >> 3) * 4 + (xcd & 3); nwg = (int)(gridDim.x >> 3) * 4; }
  else { grp = 0; wg = bid; nwg = gridDim.x; }
  const micf_block_bwd_group& g = a.g[grp];
  stage_weights_bwd(wlds, g);
  const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* PV = reinterpret_cast<const float*>(wlds + kBVec);
  const float *p_ln1g = PV, *p_ln2g = PV + C, *p_preg = PV + 2 * C;
  const uint32_t T = (uint32_t)a.geo.T;
  const int ngroup16 = (a.geo.nwin + 1) >> 1;
  constexpr bool cross = CROSS, pre = PRE;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  lds_barrier();

  for (int unit = wg * NWAVE + wave; unit < a.tiles; unit += nwg * NWAVE) {
    float pg_pre = 0.f, pb_pre = 0.f, pg2 = 0.f, pb2 = 0.f, pg1 = 0.f, pb1 = 0.f;
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      const int gt = 2 * unit + sub;
      if (gt >= ngroup16) break;
      asm volatile("" ::: "memory");       // (the weight fragments are loop-invariant LDS reads: keep them out of the registers)
      // ... and so is everything derived from the lane index: hoisted out of the loops, ~35 registers of shuffle sources, fragment
      // offsets and masks sat in (spilled) registers for the whole kernel.  An opaque copy per group makes them cheap recomputations.
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      const int li = lane & 15, lr = lane >> 4;
      // the identity as a B fragment: column li, k = 4 lr .. 4 lr + 3
      const bf16x4_t ident = pack4_bf16v(4 * lr == li ? 1.f : 0.f, 4 * lr + 1 == li ? 1.f : 0.f, 4 * lr + 2 == li ? 1.f : 0.f, 4 * lr + 3 == li ? 1.f : 0.f);
      const bool valid = (lr >> 1) == (li >> 3);          // attention: the quad lies in the token's own window
      float* tile = reinterpret_cast<float*>(wlds + kBPark) + wave * kParkFloats;
      const int win = 2 * gt + (li >> 3);
      const bool live0 = win < a.geo.nwin, live = live0 && !(a.attn_mfma & 2);
      int b = 0, d_ = 0, hh = 0, w = 0;
      a.geo.coords(live0 ? win : 0, li & 7, b, d_, hh, w);
      const uint32_t tk = (uint32_t)(((b * a.geo.D + d_) * a.geo.H + hh) * a.geo.W + w);
      const float s1v = g.s1 ? g.s1[b] : 1.f, s2v = g.s2 ? g.s2[b] : 1.f;
      const uint32_t rowC = tk * C;                     // element offset of the token's 48-wide rows

      // ---- dy (+ the producing LayerNorm's backward)
      Row12 dy;
      dy.load(at32(g.dy, rowC * 4u), lr);
      if (!live) dy.zero();
      if (pre) {
        Row12 pd, px;
        pd.load(at32(g.pre_d, rowC * 4u), lr);
        px.load(at32(g.pre_x, rowC * 4u), lr);
        float pm = *at32(g.pre_mean, tk * 4u), pr = *at32(g.pre_rstd, tk * 4u);
        if (!live) { pd.zero(); px.zero(); pm = 0.f; pr = 0.f; }
        Row12 t;
        ln_bwd12(pd.v, px, pm, pr, p_preg, lr, li, lane, tile, dy, t, pg_pre, pb_pre);
        dy = t;
        if (!live) dy.zero();
      }
      const bf16x8 dyA = dy.lo();
      const bf16x4_t dyB = dy.hi();
      // (the fp32 rows wait in the wave's LDS tile until dx1 = dy + ...: 12 registers the MLP phase and the second half's loads need)
      park12(tile, lr, li, dy.v);
      // ---- the saved fc1 pre-activation (8 consecutive features per lane and block pair)
      u32x4v hreg[6];
      {
        const uint16_t* hp = static_cast<const uint16_t*>(g.h);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          hreg[c] = *reinterpret_cast<const u32x4v*>(at32(hp, (tk * HID + 32 * c + 8 * lr) * 2u));
          if (!live) hreg[c] = u32x4v{0u, 0u, 0u, 0u};
        }
      }

      // ---- MLP backward: per 32-wide chunk of the hidden features dh = s2 (dy W2) GELU'(h) (saved), dxn2 += dh W1
      // (no store in here: the loads of the second half are issued BEFORE this group's first store -- vector-memory operations
      // retire in order, a load behind a store waits for the store's acknowledgement)
      f32x4 dn2[3] = {z4, z4, z4};
      bf16x8 dhs[6];
      {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          float h8[8];
          unpack8(hreg[c], h8);
          float4 dv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int j = 2 * c + u;
            const f32x4 acc = mfma48(frag32(wlds, kB2, j, lane), frag16(wlds, kB2h, j, lane), dyA, dyB);
            dv[u] = make_float4(s2v * acc[0] * gelu_grad_t<true>(h8[4 * u]), s2v * acc[1] * gelu_grad_t<true>(h8[4 * u + 1]),
                                s2v * acc[2] * gelu_grad_t<true>(h8[4 * u + 2]), s2v * acc[3] * gelu_grad_t<true>(h8[4 * u + 3]));
          }
          dhs[c] = to_bf16x8(dv[0], dv[1]);
#pragma unroll
          for (int j = 0; j < 3; ++j) dn2[j] = mfma32(frag32(wlds, kB1, 6 * j + c, lane), dhs[c], dn2[j]);
          __builtin_amdgcn_sched_barrier(0);             // (a chunk's fragment reads stay inside the chunk: hoisted, the six chunks' 144 registers of weights spill)
        }
      }
      // ---- what LayerNorm 2 and the first head read (requested BEFORE this group's first store: vector-memory operations retire in
      // order).  The later heads' q / k / v rows are requested one head ahead and the LayerNorm-1 input with the last head: behind stores,
      // i.e. behind their acknowledgement by L2 (~1 us; plain stores), against 18 + 14 more registers held through the whole second half.
      const uint16_t* qp = reinterpret_cast<const uint16_t*>(g.q);
      const uint16_t* kp = reinterpret_cast<const uint16_t*>(g.kv);
      bf16x4_t qn, kn, vn;
      auto load_head = [&](int h) {
        qn = *reinterpret_cast<const bf16x4_t*>(at32(qp, (rowC + 16 * h + 4 * lr) * 2u));
        kn = *reinterpret_cast<const bf16x4_t*>(at32(kp, (2 * rowC + 16 * h + 4 * lr) * 2u));
        vn = *reinterpret_cast<const bf16x4_t*>(at32(kp, (2 * rowC + C + 16 * h + 4 * lr) * 2u));
        if (!live) { qn = bf16x4_t{0, 0, 0, 0}; kn = qn; vn = qn; }
      };
      load_head(0);
      Row12 x1;
      x1.load(at32(g.x1, rowC * 4u), lr);
      float mu2 = *at32(g.stats, (2 * T + tk) * 4u), rs2 = *at32(g.stats, (3 * T + tk) * 4u);
      if (!live) { x1.zero(); mu2 = 0.f; rs2 = 0.f; }
      Row12 x;
      float mu1 = 0.f, rs1 = 0.f;

      asm volatile("" ::: "memory");
      if (live) {                                        // ... and now the first half's outputs: the bf16 copy of dy, dh
        if (g.dy16) {
          uint16_t* o = static_cast<uint16_t*>(g.dy16);
          st4u(at32(o, (rowC + 8 * lr) * 2u), dyA); st2u(at32(o, (rowC + 32 + 4 * lr) * 2u), dyB);
        }
        uint16_t* dho = reinterpret_cast<uint16_t*>(g.dh);
#pragma unroll
        for (int c = 0; c < 6; ++c) st4u(at32(dho, (tk * HID + 32 * c + 8 * lr) * 2u), dhs[c]);
      }

      // ---- dx1 = dy + LN2'(dxn2)
      __builtin_amdgcn_sched_barrier(0);
      Row12 dx1;
      {
        float d[12];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) d[4 * j + r] = dn2[j][r];
        Row12 dyp;
        unpark12(tile, lr, li, dyp.v);
        ln_bwd12(d, x1, mu2, rs2, p_ln2g, lr, li, lane, tile, dyp, dx1, pg2, pb2);
        if (!live) dx1.zero();
      }
      const bf16x8 d1A = dx1.lo();
      const bf16x4_t d1B = dx1.hi();
      if (live) {
        uint16_t* o = reinterpret_cast<uint16_t*>(g.dx1);
        st4u(at32(o, (rowC + 8 * lr) * 2u), d1A); st2u(at32(o, (rowC + 32 + 4 * lr) * 2u), d1B);
        if (g.dx1_copy) dx1.store(at32(g.dx1_copy, rowC * 4u), lr);
      }

      // ---- per head: do = s1 dx1 Wp (this head's 16 channels), the attention adjoint, dq | dk | dv out
      bf16x4_t dqf[3], dkf[3], dvf[3];
      {
        uint16_t* dqo = reinterpret_cast<uint16_t*>(g.dq);
        uint16_t* dko = reinterpret_cast<uint16_t*>(g.dkv);
#pragma unroll
        for (int h = 0; h < 3; ++h) {
          const bf16x4_t qcur = qn, kf = kn, vf = vn;
          if (h < 2) load_head(h + 1);
          else if (!cross) {
            x.load(at32(g.x, rowC * 4u), lr);
            mu1 = *at32(g.stats, tk * 4u); rs1 = *at32(g.stats, (T + tk) * 4u);
            if (!live) { x.zero(); mu1 = 0.f; rs1 = 0.f; }
          }
          const f32x4 dacc = mfma48(frag32(wlds, kBP, h, lane), frag16(wlds, kBPh, h, lane), d1A, d1B);
          const bf16x4_t of = pack4_bf16v(s1v * dacc[0], s1v * dacc[1], s1v * dacc[2], s1v * dacc[3]);
          const uint2 qu = __builtin_bit_cast(uint2, qcur);
          const bf16x4_t qf = pack4_bf16v(__uint_as_float(qu.x << 16) * a.scale, __uint_as_float(qu.x & 0xFFFF0000u) * a.scale,
                                          __uint_as_float(qu.y << 16) * a.scale, __uint_as_float(qu.y & 0xFFFF0000u) * a.scale);
          const f32x4 s1 = mfma16(kf, qf, z4);           // S[query li][keys 4 lr ..]
          const f32x4 p1 = mfma16(vf, of, z4);           // dP[query li][keys 4 lr ..]
          const f32x4 s2 = mfma16(qf, kf, z4);           // S[queries 4 lr ..][key li]
          const f32x4 p2 = mfma16(of, vf, z4);           // dP[queries 4 lr ..][key li]
          float m = valid ? fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])) : -INFINITY;
          m = fmaxf(m, __shfl_xor(m, 16, 64));
          float e[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = valid ? expf(s1[r] - m) : 0.f;
          float sum = (e[0] + e[1]) + (e[2] + e[3]);
          sum += __shfl_xor(sum, 16, 64);
          const float inv = valid ? 1.0f / sum : 0.f;
          float dot = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) { e[r] *= inv; dot += valid ? e[r] * p1[r] : 0.f; }
          dot += __shfl_xor(dot, 16, 64);
          float ds1[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) ds1[r] = valid ? e[r] * (p1[r] - dot) : 0.f;
          const bf16x4_t dsf1 = pack4_bf16v(ds1[0], ds1[1], ds1[2], ds1[3]);          // dS^T[keys 4 lr ..][query li]
          float e2[4], ds2[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {                  // layout 2: the statistics of query 4 lr + r live in lane (li = that query, its window's pair)
            const int qi = 4 * lr + r, src = qi + 16 * (2 * (qi >> 3));
            const float mq = __shfl(m, src, 64), iq = __shfl(inv, src, 64), dq_ = __shfl(dot, src, 64);
            e2[r] = valid ? expf(s2[r] - mq) * iq : 0.f;
            ds2[r] = valid ? e2[r] * (p2[r] - dq_) : 0.f;
          }
          const bf16x4_t pf2 = pack4_bf16v(e2[0], e2[1], e2[2], e2[3]);               // P[queries 4 lr ..][key li]
          const bf16x4_t dsf2 = pack4_bf16v(ds2[0], ds2[1], ds2[2], ds2[3]);          // dS[queries 4 lr ..][key li]
          // channel-major operands (rows = channels 16 h + li, k = tokens 4 lr ..): the row fragments times the identity
          const bf16x4_t kt = pack4q(mfma16(kf, ident, z4)), qt = pack4q(mfma16(qf, ident, z4)), ot = pack4q(mfma16(of, ident, z4));
          const f32x4 ga = mfma16(kt, dsf1, z4);         // dQ^T[ch][query li] / scale
          const f32x4 gb = mfma16(qt, dsf2, z4);         // dK^T[ch][key li]
          const f32x4 gc = mfma16(ot, pf2, z4);          // dV^T[ch][key li]
          dqf[h] = pack4_bf16v(ga[0] * a.scale, ga[1] * a.scale, ga[2] * a.scale, ga[3] * a.scale);
          dkf[h] = pack4q(gb);
          dvf[h] = pack4q(gc);
          if (live) {
            st2u(at32(dqo, (rowC + 16 * h + 4 * lr) * 2u), dqf[h]);
            st2u(at32(dko, (2 * rowC + 16 * h + 4 * lr) * 2u), dkf[h]);
            st2u(at32(dko, (2 * rowC + C + 16 * h + 4 * lr) * 2u), dvf[h]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      // ---- the input gradients
      __builtin_amdgcn_sched_barrier(0);
      if (!cross) {
        float d[12];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          f32x4 acc = z4;
#pragma unroll
          for (int h = 0; h < 3; ++h) acc = mfma16(frag16(wlds, kBQ, 3 * j + h, lane), dqf[h], acc);
#pragma unroll
          for (int h = 0; h < 6; ++h) acc = mfma16(frag16(wlds, kBKV, 6 * j + h, lane), h < 3 ? dkf[h] : dvf[h - 3], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) d[4 * j + r] = acc[r];
        }
        Row12 dx;
        ln_bwd12(d, x, mu1, rs1, p_ln1g, lr, li, lane, tile, dx1, dx, pg1, pb1);
        if (live) dx.store(at32(g.dx, rowC * 4u), lr);
      } else {
        Row12 dxq, dxs;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          f32x4 acc = z4, acs = z4;
#pragma unroll
          for (int h = 0; h < 3; ++h) acc = mfma16(frag16(wlds, kBQ, 3 * j + h, lane), dqf[h], acc);
#pragma unroll
          for (int h = 0; h < 6; ++h) acs = mfma16(frag16(wlds, kBKV, 6 * j + h, lane), h < 3 ? dkf[h] : dvf[h - 3], acs);
#pragma unroll
          for (int r = 0; r < 4; ++r) { dxq.v[4 * j + r] = acc[r]; dxs.v[4 * j + r] = acs[r]; }
        }
        if (live) {
          dxq.store(at32(g.dx, rowC * 4u), lr);
          dxs.store(at32(g.dxs, rowC * 4u), lr);
        }
      }
    }
    // ---- the tile's row of LayerNorm partial sums: dgamma | dbeta, lane c < 48 owns channel c
    if (lane0 < C) {
      const int64_t row = (int64_t)unit * 2 * C;
      if (g.ln2_part) { g.ln2_part[row + lane0] = pg2; g.ln2_part[row + C + lane0] = pb2; }
      if (!cross && g.ln1_part) { g.ln1_part[row + lane0] = pg1; g.ln1_part[row + C + lane0] = pb1; }
      if (pre) { g.pre_part[row + lane0] = pg_pre; g.pre_part[row + C + lane0] = pb_pre; }
    }
  }
}

static int launch_bwd_wave48(const BlkBwdArgs& a, hipStream_t s) {
  int nwg = (a.tiles + NWAVE - 1) / NWAVE;
  if (nwg > 256) nwg = 256;
  const unsigned grid = a.G == 2 ? (unsigned)((nwg + 3) / 4 * 8) : (unsigned)nwg;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_bwd_wave48_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_bwd_wave48_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_bwd_wave48_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const bool cross = a.g[0].dxs != nullptr, pre = a.g[0].pre_d != nullptr;
  if (cross) hipLaunchKernelGGL((block_bwd_wave48_kernel<true, false>), dim3(grid), dim3(NTHR), kBwdLdsBytes, s, a);
  else if (pre) hipLaunchKernelGGL((block_bwd_wave48_kernel<false, true>), dim3(grid), dim3(NTHR), kBwdLdsBytes, s, a);
  else hipLaunchKernelGGL((block_bwd_wave48_kernel<false, false>), dim3(grid), dim3(NTHR), kBwdLdsBytes, s, a);
  MICF_RETURN_LAUNCH();
}

}  // namespace wave48
}  // namespace micf
