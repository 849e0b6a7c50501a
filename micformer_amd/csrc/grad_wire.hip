// grad_wire.hip -- the bf16 wire format of the data-parallel gradient exchange: bf16 on the xGMI links,
// fp32 in every sum.  A slice of the flat fp32 gradient is (1) rounded to bf16 into N equal, zero-padded shards, (2) shard j of
// every rank travels to rank j (all-to-all, one hop on the point-to-point xGMI mesh), (3) rank j adds its N received shards in
// fp32 and rounds the SUM to bf16 once, (4) the reduced shards are all-gathered and (5) widened back into the flat buffer.
// Steps 1, 3, 5 are the three streaming kernels below; 2 and 4 are RCCL collectives issued by micformer_amd/dist.py.
#include "common.h"

namespace micf {

__device__ __forceinline__ uint32_t wire_pack2(float a, float b) {
  uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));   // gfx950: round-to-nearest-even, one instruction per pair
  return r;
}
__device__ __forceinline__ float wire_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float wire_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// dst[i] = bf16(src[i]) for i < n, 0 for n <= i < padded (padded % 8 == 0): 32 B in, 16 B out per thread and step
__global__ void __launch_bounds__(256) wire_pack_kernel(const float* __restrict__ src, int64_t n, uint4* __restrict__ dst, int64_t padded8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < padded8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 8;
    float v[8];
    if (e + 8 <= n) {
      const float4 a = *reinterpret_cast<const float4*>(src + e), b = *reinterpret_cast<const float4*>(src + e + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (e + k < n) ? src[e + k] : 0.f;
    }
    dst[i] = make_uint4(wire_pack2(v[0], v[1]), wire_pack2(v[2], v[3]), wire_pack2(v[4], v[5]), wire_pack2(v[6], v[7]));
  }
}

// out[i] = bf16( sum_r float(recv[r * shard + i]) ), the sum in fp32 in rank order (every rank adds in the same order)
__global__ void __launch_bounds__(256) wire_sum_kernel(const uint4* __restrict__ recv, int ranks, int64_t shard8, uint4* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < shard8; i += (int64_t)gridDim.x * blockDim.x) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < ranks; ++r) {
      const uint4 u = recv[(int64_t)r * shard8 + i];
      s[0] += wire_lo(u.x); s[1] += wire_hi(u.x); s[2] += wire_lo(u.y); s[3] += wire_hi(u.y);
      s[4] += wire_lo(u.z); s[5] += wire_hi(u.z); s[6] += wire_lo(u.w); s[7] += wire_hi(u.w);
    }
    out[i] = make_uint4(wire_pack2(s[0], s[1]), wire_pack2(s[2], s[3]), wire_pack2(s[4], s[5]), wire_pack2(s[6], s[7]));
  }
}

// dst[i] = float(src[i]) for i < n
__global__ void __launch_bounds__(256) wire_unpack_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t n8 = (n + 7) / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 8;
    if (e + 8 <= n) {
      const uint4 u = *reinterpret_cast<const uint4*>(src + e);
      *reinterpret_cast<float4*>(dst + e) = make_float4(wire_lo(u.x), wire_hi(u.x), wire_lo(u.y), wire_hi(u.y));
      *reinterpret_cast<float4*>(dst + e + 4) = make_float4(wire_lo(u.z), wire_hi(u.z), wire_lo(u.w), wire_hi(u.w));
    } else {
      for (int64_t k = e; k < n; ++k) dst[k] = __uint_as_float((uint32_t)src[k] << 16);
    }
  }
}

static inline int wire_blocks(int64_t units) {
  int64_t b = (units + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace micf

extern "C" int micf_grad_wire_pack(const float* src, int64_t n, void* dst, int64_t padded, micf_stream_t stream) {
  if (!src || !dst || n < 0 || padded < n || (padded & 7)) return MICF_EINVAL;
  if (!micf::aligned16(src) || !micf::aligned16(dst)) return MICF_EINVAL;
  if (padded == 0) return MICF_OK;
  hipLaunchKernelGGL(micf::wire_pack_kernel, dim3(micf::wire_blocks(padded / 8)), dim3(256), 0, (hipStream_t)stream, src, n,
                     static_cast<uint4*>(dst), padded / 8);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_grad_wire_sum(const void* recv, int ranks, int64_t shard, void* out, micf_stream_t stream) {
  if (!recv || !out || ranks < 1 || shard < 0 || (shard & 7)) return MICF_EINVAL;
  if (!micf::aligned16(recv) || !micf::aligned16(out)) return MICF_EINVAL;
  if (shard == 0) return MICF_OK;
  hipLaunchKernelGGL(micf::wire_sum_kernel, dim3(micf::wire_blocks(shard / 8)), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const uint4*>(recv), ranks, shard / 8, static_cast<uint4*>(out));
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_grad_wire_unpack(const void* src, float* dst, int64_t n, micf_stream_t stream) {
  if (!src || !dst || n < 0) return MICF_EINVAL;
  if (!micf::aligned16(src) || !micf::aligned16(dst)) return MICF_EINVAL;
  if (n == 0) return MICF_OK;
  hipLaunchKernelGGL(micf::wire_unpack_kernel, dim3(micf::wire_blocks((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const uint16_t*>(src), dst, n);
  MICF_RETURN_LAUNCH();
}
