// conv3_layout.h -- the re-laid-out copies of a few-output-channel 3x3x3 conv weight w[N][Cin][27] that the direct forward
// kernel streams (conv3_fwdx.hip), written either by that call's own re-layout launch or, once per step, by
// micf_conv3_weight_prep_grouped (conv3.hip).  One buffer, two parts:
//   fp32 part  [chunk][tap][16 n][16 c]                      (chunks * 27 * 256 floats)       A fragment of 4 k-steps = one float4
//   bf16 part  [chunk][tap pair][16 n][4 lr][8]              (chunks * 14 * 512 bf16)         A fragment of one 32-deep MFMA =
//              element e < 4: tap 2p, channel 4 lr + e; e >= 4: tap 2p + 1 (zero for the unpaired 27th tap), channel 4 lr + e - 4
#pragma once
#include <stdint.h>

namespace micf {
__host__ __device__ inline int64_t conv3_fwd_layout_f32(int Cin) { return (int64_t)((Cin + 15) / 16) * 27 * 256; }
__host__ __device__ inline int64_t conv3_fwd_layout_bf16(int Cin) { return (int64_t)((Cin + 15) / 16) * 14 * 512; }   // bf16 elements
__host__ __device__ inline int64_t conv3_fwd_layout_floats(int Cin) { return conv3_fwd_layout_f32(Cin) + conv3_fwd_layout_bf16(Cin) / 2; }
__host__ __device__ inline int64_t conv3_fwd_layout_items(int Cin) { return conv3_fwd_layout_f32(Cin) + conv3_fwd_layout_bf16(Cin); }

// item `id` of conv3_fwd_layout_items(Cin): one element of the fp32 part, then one element of the bf16 part
__device__ __forceinline__ void conv3_fwd_layout_write(const float* __restrict__ w, float* __restrict__ wt, int N, int Cin, int64_t id) {
  const int64_t nf = conv3_fwd_layout_f32(Cin);
  if (id < nf) {
    const int c = (int)(id & 15), n = (int)((id >> 4) & 15);
    const int tap = (int)((id >> 8) % 27), chunk = (int)((id >> 8) / 27);
    const int cc = chunk * 16 + c;
    wt[id] = (n < N && cc < Cin) ? w[((int64_t)n * Cin + cc) * 27 + tap] : 0.f;
    return;
  }
  const int64_t j = id - nf;
  const int e = (int)(j & 7), lr = (int)((j >> 3) & 3), n = (int)((j >> 5) & 15);
  const int p = (int)((j >> 9) % 14), chunk = (int)((j >> 9) / 14);
  const int tap = 2 * p + (e >> 2), cc = chunk * 16 + 4 * lr + (e & 3);
  const float v = (n < N && cc < Cin && tap < 27) ? w[((int64_t)n * Cin + cc) * 27 + tap] : 0.f;
  unsigned u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);                      // round-to-nearest-even, as pack_bf16
  reinterpret_cast<uint16_t*>(wt + nf)[j] = (uint16_t)(u >> 16);
}

// ---- data-gradient kernel (conv3_bwdx.hip), O = input channels (multiple of 16), N <= 16 dy channels:
//   fp32 part  [tap][O][16 n]                       (27 * O * 16 floats)
//   bf16 part  [tap pair][O][4 lr][8]               (14 * O * 32 bf16)   e < 4: tap 2p, n = 4 lr + e; e >= 4: tap 2p + 1 (zero for
//              the 27th), n = 4 lr + e - 4
__host__ __device__ inline int64_t conv3_bwd_layout_f32(int O) { return (int64_t)27 * O * 16; }
__host__ __device__ inline int64_t conv3_bwd_layout_bf16(int O) { return (int64_t)14 * O * 32; }
__host__ __device__ inline int64_t conv3_bwd_layout_floats(int O) { return conv3_bwd_layout_f32(O) + conv3_bwd_layout_bf16(O) / 2; }
__host__ __device__ inline int64_t conv3_bwd_layout_items(int O) { return conv3_bwd_layout_f32(O) + conv3_bwd_layout_bf16(O); }
__device__ __forceinline__ void conv3_bwd_layout_write(const float* __restrict__ w, float* __restrict__ wt, int N, int O, int64_t id) {
  const int64_t nf = conv3_bwd_layout_f32(O);
  if (id < nf) {
    const int n = (int)(id & 15);
    const int c = (int)((id >> 4) % O);
    const int tap = (int)((id >> 4) / O);
    wt[id] = n < N ? w[((int64_t)n * O + c) * 27 + tap] : 0.f;
    return;
  }
  const int64_t j = id - nf;
  const int e = (int)(j & 7), lr = (int)((j >> 3) & 3);
  const int c = (int)((j >> 5) % O), p = (int)((j >> 5) / O);
  const int tap = 2 * p + (e >> 2), n = 4 * lr + (e & 3);
  const float v = (n < N && tap < 27) ? w[((int64_t)n * O + c) * 27 + tap] : 0.f;
  unsigned u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  reinterpret_cast<uint16_t*>(wt + nf)[j] = (uint16_t)(u >> 16);
}
}  // namespace micf
