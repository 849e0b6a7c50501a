// block_wave.h -- shared pieces of the wave-private block kernels at C = 48 (block_wave_fwd.h, block_wave_bwd.h): the fragment
// addressing of the weights staged in LDS, the matrix-core wrappers, and a 48-wide token row in "layout P".
#pragma once

namespace micf {
namespace wave48 {

constexpr int C = 48, HID = 192, NWAVE = 8, NTHR = 64 * NWAVE;

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

// output row r of block j of a 48-wide (three blocks) / 192-wide (twelve blocks) product -> feature ("layout P": the quads of blocks
// 2c and 2c + 1 of lane group lr are features 32 c + 8 lr + 0..7; the odd block out of 48 keeps 32 + r)
__device__ __forceinline__ int row_p48(int j, int r) { return j < 2 ? 8 * (r >> 2) + 4 * j + (r & 3) : 32 + r; }
__device__ __forceinline__ int row_p192(int j, int r) { return 32 * (j >> 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3); }
// element offset of (row, k) in a K16-blocked [R, 16 KB] matrix (micf_weight_prep_grouped, bf16 = 2)
__device__ __forceinline__ int k16(int row, int k, int KB) { return ((row >> 4) * KB + (k >> 4)) * 256 + (row & 15) * 16 + (k & 15); }

__device__ __forceinline__ bf16x8 frag32(const char* lds, int base, int f, int lane) {
  return *reinterpret_cast<const bf16x8*>(lds + base + (f * 64 + lane) * 16);
}
__device__ __forceinline__ bf16x4_t frag16(const char* lds, int base, int f, int lane) {
  return *reinterpret_cast<const bf16x4_t*>(lds + base + (f * 64 + lane) * 8);
}
__device__ __forceinline__ f32x4 mfma32(const bf16x8& a, const bf16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(const bf16x4_t& a, const bf16x4_t& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
// A 16x16x16 product must not take the accumulator of the 16x16x32 product issued right before it: on gfx950 that back-to-back
// dependent pair of DIFFERENT shapes returned wrong values in the first two accumulator registers (measured: exactly the blocks the
// scheduler had left adjacent), and the compiler inserts no wait state for it.  A K = 48 product is therefore two INDEPENDENT
// products (k = 0..31 and k = 32..47) and one vector add; chains of equal shapes (proj, fc2) are the ordinary, interlocked case.
__device__ __forceinline__ f32x4 mfma48(const bf16x8& a32, const bf16x4_t& a16, const bf16x8& b32, const bf16x4_t& b16) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 p = mfma32(a32, b32, z), q = mfma16(a16, b16, z);
  return f32x4{p[0] + q[0], p[1] + q[1], p[2] + q[2], p[3] + q[3]};
}
__device__ __forceinline__ float sum4(float v) {          // over the 4 lane groups of a token
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
// PLAIN stores: a lane writes 8 or 16 bytes of a token row and the rest of the 128-byte line follows from other instructions of the
// same wave up to a head's worth of work later -- the write-back L2 merges them; streaming (nt) stores went out as partial lines
// (measured: 121 us instead of 51 us for the forward launch).  The weights live in LDS here, so there is nothing in L2 to protect.
__device__ __forceinline__ void st4u(void* p, const bf16x8& v) { *reinterpret_cast<u32x4v*>(p) = __builtin_bit_cast(u32x4v, v); }
__device__ __forceinline__ void st2u(void* p, const bf16x4_t& v) { *reinterpret_cast<u32x2v*>(p) = __builtin_bit_cast(u32x2v, v); }
__device__ __forceinline__ bf16x4_t pack4q(const f32x4& v) { return pack4_bf16v(v[0], v[1], v[2], v[3]); }

// a 48-wide fp32 token row in layout P: 12 values per lane (features 8 lr .. 8 lr + 7, then 32 + 4 lr .. + 3)
struct Row12 {
  float v[12];
  __device__ __forceinline__ void load(const float* row, int lr) {
    const float4 a = ld4g(row + 8 * lr), b = ld4g(row + 8 * lr + 4), c = ld4g(row + 32 + 4 * lr);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int e = 0; e < 12; ++e) v[e] = 0.f;
  }
  __device__ __forceinline__ void store(float* row, int lr) const {
    *reinterpret_cast<float4*>(row + 8 * lr) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(row + 8 * lr + 4) = make_float4(v[4], v[5], v[6], v[7]);
    *reinterpret_cast<float4*>(row + 32 + 4 * lr) = make_float4(v[8], v[9], v[10], v[11]);
  }
  __device__ __forceinline__ bf16x8 lo() const { return to_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7])); }
  __device__ __forceinline__ bf16x4_t hi() const { return pack4_bf16v(v[8], v[9], v[10], v[11]); }
};
// the same pieces of a parameter vector in LDS
__device__ __forceinline__ void vec12(const float* p, int lr, float (&o)[12]) {
  const float4 a = *reinterpret_cast<const float4*>(p + 8 * lr), b = *reinterpret_cast<const float4*>(p + 8 * lr + 4),
               c = *reinterpret_cast<const float4*>(p + 32 + 4 * lr);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; o[8] = c.x; o[9] = c.y; o[10] = c.z; o[11] = c.w;
}
// LayerNorm of a Row12 (statistics over the token's 48 features = this lane's 12 + the three other lane groups')
__device__ __forceinline__ void layernorm12(const Row12& x, const float* gam, const float* bet, int lr, float eps, Row12& y, float& mu, float& rs) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 12; e += 4) s += (x.v[e] + x.v[e + 1]) + (x.v[e + 2] + x.v[e + 3]);
  mu = sum4(s) * (1.0f / C);
  float qd = 0.f;
#pragma unroll
  for (int e = 0; e < 12; e += 4) {
    const float d0 = x.v[e] - mu, d1 = x.v[e + 1] - mu, d2 = x.v[e + 2] - mu, d3 = x.v[e + 3] - mu;
    qd += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  rs = 1.0f / sqrtf(sum4(qd) * (1.0f / C) + eps);
  float gm[12], bt[12];
  vec12(gam, lr, gm);
  vec12(bet, lr, bt);
#pragma unroll
  for (int e = 0; e < 12; ++e) y.v[e] = (x.v[e] - mu) * rs * gm[e] + bt[e];
}

}  // namespace wave48
}  // namespace micf
