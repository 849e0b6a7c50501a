// attn_fp8.h -- window attention with fp8 (OCP e4m3, what gfx950's conversion and matrix instructions implement) operands on the
// matrix cores: BASELINE config 4's "fp8 MFMA for QK^T / AV" (the reference's own reduced-precision site is the fp16 autocast of
// its validation forward, utils.py:236-238; it has no fp8 path -- this is the north_star's precision leg, MICF_DTYPE_BF16_ATTN_FP8).
//
// A 2x2x2 window has 8 tokens, a matrix-core tile 16 rows: one unit is TWO windows x one head.
//   S^T = K (q scale)^T     v_mfma_f32_16x16x32_fp8_fp8, k = the head's channels (HD = 32 fills the 32-deep tile exactly; HD = 16
//                           feeds zeros to the upper half), rows = keys, columns = queries: the accumulator quad of lane (li, lr)
//                           is S[query li][keys 4 lr .. 4 lr + 3]
//   P = softmax over the 8 keys of the query's own window: 4 values in the lane + one exchange with lane (li, lr ^ 1); the other
//       window's 8 keys are masked (P = 0)
//   O^T = V^T P^T           rows = 16 channels (HD / 16 products), k = the 16 keys (upper half of the 32-deep tile zero), columns =
//                           queries: lane (li, lr) ends up with 4 consecutive channels of query li -- the store shape of the tile kernels
// Operands are rounded to e4m3 (round-to-nearest-even: v_cvt_pk_fp8_f32) where they enter a fragment: q * scale, k, v and P; the
// products are exact in fp32 and accumulate in fp32.  The backward is the bf16 path's (straight-through: it differentiates the
// unquantised attention of the saved q / k / v).
#pragma once
#include "common.h"
#include "gemm_dma.h"

namespace micf {

// e4m3's largest finite value is 448: an operand beyond it (an activation outlier in k / v) must SATURATE, whatever overflow mode
// the conversion instruction is in (NaN in the non-saturating one) -- one v_med3_f32 per value; NaN inputs stay NaN.
__device__ __forceinline__ float sat_fp8(float v) { return __builtin_amdgcn_fmed3f(v, -448.f, 448.f); }

// 8 floats -> 8 e4m3 bytes (element e in byte e), saturating
__device__ __forceinline__ long pack8_fp8(const float4& lo, const float4& hi) {
  int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(lo.x), sat_fp8(lo.y), 0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(lo.z), sat_fp8(lo.w), w0, true);
  int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(hi.x), sat_fp8(hi.y), 0, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(hi.z), sat_fp8(hi.w), w1, true);
  return (long)(((unsigned long)(unsigned)w1 << 32) | (unsigned long)(unsigned)w0);
}
// (P only: softmax outputs lie in [0, 1], no saturation needed)
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}
// the value a float has after a round trip through e4m3 (the VALU restatement of the same arithmetic: window_attn.hip)
__device__ __forceinline__ float round_fp8(float v) {
  const int w = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(v), 0.f, 0, false);
  return __builtin_amdgcn_cvt_f32_fp8(w, 0);
}

// One unit.  qp / kp / vp: row 0 of the 16-token group at the head's first channel (LDS or global, fp32, 16-byte aligned), ld: row
// stride in floats.  All 64 lanes of a wave call it together.  out[cb] = O[query li][16 cb + 4 lr .. + 3].
template <int HD>
__device__ __forceinline__ void attn16_fp8(const float* qp, const float* kp, const float* vp, int ld, float scale, float4 (&out)[HD / 16]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  const bool kin = 8 * lr < HD;                                   // (HD = 16: lane groups 2, 3 hold the zero half of the k range)
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ko = kin ? 8 * lr : 0;
  float4 q0 = *reinterpret_cast<const float4*>(qp + li * ld + ko), q1 = *reinterpret_cast<const float4*>(qp + li * ld + ko + 4);
  float4 k0 = *reinterpret_cast<const float4*>(kp + li * ld + ko), k1 = *reinterpret_cast<const float4*>(kp + li * ld + ko + 4);
  q0 = make_float4(q0.x * scale, q0.y * scale, q0.z * scale, q0.w * scale);
  q1 = make_float4(q1.x * scale, q1.y * scale, q1.z * scale, q1.w * scale);
  if (!kin) { q0 = z4; q1 = z4; k0 = z4; k1 = z4; }
  const f32x4 st = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack8_fp8(k0, k1), pack8_fp8(q0, q1), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  // softmax of query li over the keys of ITS window: this lane's quad is valid iff its key quad lies in that window
  const bool valid = (lr >> 1) == (li >> 3);
  float m = valid ? fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])) : -INFINITY;
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  float e[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] = valid ? expf(st[r] - m) : 0.f;
  float sum = (e[0] + e[1]) + (e[2] + e[3]);
  sum += __shfl_xor(sum, 16, 64);
  const float inv = valid ? 1.0f / sum : 0.f;
  const unsigned pw = pack4_fp8(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);
  // P^T fragment: lane (li, lr') supplies keys 8 lr' .. 8 lr' + 7 of query li = the packed quads of lanes (li, 2 lr') and (li, 2 lr' + 1)
  const int src = li + 16 * (2 * (lr & 1));
  const unsigned plo = (unsigned)__shfl((int)pw, src, 64), phi = (unsigned)__shfl((int)pw, src + 16, 64);
  const long pfrag = lr < 2 ? (long)(((unsigned long)phi << 32) | (unsigned long)plo) : 0L;
#pragma unroll
  for (int cb = 0; cb < HD / 16; ++cb) {
    // V^T fragment: rows = channels 16 cb + li, k = keys 8 lr .. 8 lr + 7 (lane groups 2, 3: the zero half)
    float v[8];
    const float* vb = vp + (lr < 2 ? 8 * lr : 0) * ld + 16 * cb + li;
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = vb[t * ld];
    const long vfrag = lr < 2 ? pack8_fp8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7])) : 0L;
    const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(vfrag, pfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    out[cb] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---- the same unit with bf16 operands (MICF_DTYPE_BF16's default attention on the block kernels, round 4): v_mfma_f32_16x16x16_bf16
// (k = 16: one product per 16 channels of the head for S^T, and -- the 16 keys being exactly one k range -- the accumulator quad of
// S^T IS the P^T operand of O^T = V^T P^T: no exchange between the lanes at all).  q * scale, k, v and P are rounded to bf16 (RNE)
// where they enter a fragment, fp32 accumulation; the VALU form (the fp32 parity mode) keeps q / k / v / P in fp32.
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4_t pack4_bf16v(float a, float b, float c, float d) {
  const unsigned lo = pack_bf16(a, b), hi = pack_bf16(c, d);
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(bf16x4_t, u32x2_t{lo, hi});
}

template <int HD>
__device__ __forceinline__ void attn16_bf16(const float* qp, const float* kp, const float* vp, int ld, float scale, float4 (&out)[HD / 16]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) {
    const float4 q4 = *reinterpret_cast<const float4*>(qp + li * ld + 16 * ks + 4 * lr);
    const float4 k4 = *reinterpret_cast<const float4*>(kp + li * ld + 16 * ks + 4 * lr);
    st = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack4_bf16v(k4.x, k4.y, k4.z, k4.w),
                                                  pack4_bf16v(q4.x * scale, q4.y * scale, q4.z * scale, q4.w * scale), st, 0, 0, 0);
  }
  const bool valid = (lr >> 1) == (li >> 3);
  float m = valid ? fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])) : -INFINITY;
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  float e[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] = valid ? expf(st[r] - m) : 0.f;
  float sum = (e[0] + e[1]) + (e[2] + e[3]);
  sum += __shfl_xor(sum, 16, 64);
  const float inv = valid ? 1.0f / sum : 0.f;
  const bf16x4_t pfrag = pack4_bf16v(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);       // P^T[keys 4 lr ..][query li]
#pragma unroll
  for (int cb = 0; cb < HD / 16; ++cb) {
    const float* vb = vp + 4 * lr * ld + 16 * cb + li;              // V^T fragment: rows = channels 16 cb + li, k = keys 4 lr .. 4 lr + 3
    const bf16x4_t vfrag = pack4_bf16v(vb[0], vb[ld], vb[2 * ld], vb[3 * ld]);
    const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vfrag, pfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    out[cb] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---- the BACKWARD of the same unit on the matrix cores (round 5; the bf16 mode and its fp8-attention variant: e4m3 cannot hold
// gradients without a per-tensor scale -- its smallest normal number is 2^-6 -- so the adjoint products take bf16 operands in
// both).  Two windows x one head, q / k / v / dO rows in LDS as fp32 (row stride ldq for q / k / v, ldo for dO):
//   layout 1 (lane (li, lr) = query li, keys 4 lr ..):  S^T = K Qs^T, dP^T = V dO^T  ->  P, dS = P (dP - <P, dP>);  dQ^T = K^T dS^T
//   layout 2 (lane (li, lr) = key li, queries 4 lr ..):  S = Qs K^T, dP = dO V^T     ->  P, dS from the row statistics of layout 1
//                                                        (12 lane reads);  dK^T = Qs^T dS,  dV^T = dO^T P
// -- in both layouts the accumulator quad of the score product IS the k-operand of the next product (16 keys / 16 queries = one k
// range of v_mfma_f32_16x16x16_bf16), so nothing is transposed through LDS.  Qs = q * scale, hence dq = scale * dS K and dk = dS^T Qs.
// Operands (Qs, k, v, dO, P, dS) are rounded to bf16 where they enter a fragment, fp32 accumulation; 7 HD / 16 products per unit.
// out: dq / dk / dv[cb] = 4 consecutive channels 16 cb + 4 lr .. of token li (query li for dq, key li for dk / dv).
template <int HD>
__device__ __forceinline__ void attn16_bwd_bf16(const float* qp, const float* kp, const float* vp, int ldq, const float* dop, int ldo,
                                                float scale, float4 (&dq)[HD / 16], float4 (&dk)[HD / 16], float4 (&dv)[HD / 16]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lr = lane >> 4;
  constexpr int KS = HD / 16;
  bf16x4_t qf[KS], kf[KS], vf[KS], of[KS];                // row fragments: token li, channels 16 ks + 4 lr ..
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float4 q4 = *reinterpret_cast<const float4*>(qp + li * ldq + 16 * ks + 4 * lr);
    const float4 k4 = *reinterpret_cast<const float4*>(kp + li * ldq + 16 * ks + 4 * lr);
    const float4 v4 = *reinterpret_cast<const float4*>(vp + li * ldq + 16 * ks + 4 * lr);
    const float4 o4 = *reinterpret_cast<const float4*>(dop + li * ldo + 16 * ks + 4 * lr);
    qf[ks] = pack4_bf16v(q4.x * scale, q4.y * scale, q4.z * scale, q4.w * scale);
    kf[ks] = pack4_bf16v(k4.x, k4.y, k4.z, k4.w);
    vf[ks] = pack4_bf16v(v4.x, v4.y, v4.z, v4.w);
    of[ks] = pack4_bf16v(o4.x, o4.y, o4.z, o4.w);
  }
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 s1 = z, p1 = z, s2 = z, p2 = z;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    s1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf[ks], qf[ks], s1, 0, 0, 0);     // S[query li][keys 4 lr ..]
    p1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[ks], of[ks], p1, 0, 0, 0);     // dP[query li][keys 4 lr ..]
    s2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qf[ks], kf[ks], s2, 0, 0, 0);     // S[queries 4 lr ..][key li]
    p2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(of[ks], vf[ks], p2, 0, 0, 0);     // dP[queries 4 lr ..][key li]
  }
  // (both layouts: the quad lies in the token's own window iff the two window indices agree)
  const bool valid = (lr >> 1) == (li >> 3);
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
  const bf16x4_t dsf1 = pack4_bf16v(ds1[0], ds1[1], ds1[2], ds1[3]);               // dS^T[keys 4 lr ..][query li]
  // layout 2: the statistics of query 4 lr + r live in the lanes (li = that query, lr in its own window's pair)
  float e2[4], ds2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = 4 * lr + r, src = qi + 16 * (2 * (qi >> 3));
    const float mq = __shfl(m, src, 64), iq = __shfl(inv, src, 64), dq_ = __shfl(dot, src, 64);
    e2[r] = valid ? expf(s2[r] - mq) * iq : 0.f;
    ds2[r] = valid ? e2[r] * (p2[r] - dq_) : 0.f;
  }
  const bf16x4_t pf2 = pack4_bf16v(e2[0], e2[1], e2[2], e2[3]);                    // P[queries 4 lr ..][key li]
  const bf16x4_t dsf2 = pack4_bf16v(ds2[0], ds2[1], ds2[2], ds2[3]);               // dS[queries 4 lr ..][key li]
#pragma unroll
  for (int cb = 0; cb < KS; ++cb) {
    // transposed fragments: rows = channels 16 cb + li, k = tokens 4 lr .. 4 lr + 3 (column reads of the LDS rows)
    const float* kb = kp + 4 * lr * ldq + 16 * cb + li;
    const float* qb = qp + 4 * lr * ldq + 16 * cb + li;
    const float* ob = dop + 4 * lr * ldo + 16 * cb + li;
    const bf16x4_t kt = pack4_bf16v(kb[0], kb[ldq], kb[2 * ldq], kb[3 * ldq]);
    const bf16x4_t qt = pack4_bf16v(qb[0] * scale, qb[ldq] * scale, qb[2 * ldq] * scale, qb[3 * ldq] * scale);
    const bf16x4_t ot = pack4_bf16v(ob[0], ob[ldo], ob[2 * ldo], ob[3 * ldo]);
    const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt, dsf1, z, 0, 0, 0);   // dQ^T[ch][query li] / scale
    const f32x4 b = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qt, dsf2, z, 0, 0, 0);   // dK^T[ch][key li]
    const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ot, pf2, z, 0, 0, 0);    // dV^T[ch][key li]
    dq[cb] = make_float4(a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale);
    dk[cb] = make_float4(b[0], b[1], b[2], b[3]);
    dv[cb] = make_float4(c[0], c[1], c[2], c[3]);
  }
}

}  // namespace micf
