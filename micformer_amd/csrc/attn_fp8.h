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
// where they enter a fragment, fp32 accumulation; the VALU form kept q / k / v / P in fp32 (MICF_ATTN_VALU=1 restores it).
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

}  // namespace micf
