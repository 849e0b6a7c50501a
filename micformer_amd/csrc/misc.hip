// misc.hip -- ABI bookkeeping + the layout helpers of the block wrappers: zero-pad / crop to window multiples
// (F.pad MS.py:349-350, 483; crop MS.py:399-400, 497-498) and the trilinear align_corners=True resize of the decoder's
// odd-size branch (F.interpolate MS.py:1018-1025) with its adjoint.
#include <hip/hip_fp16.h>

#include "common.h"
#include <string.h>

namespace micf {

__global__ void __launch_bounds__(256) pad3d_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int D, int H,
                                                    int W, int Dp, int Hp, int Wp, int C, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i / C; const int c = (int)(i - t * C);
    const int w = (int)(t % Wp); t /= Wp; const int h = (int)(t % Hp); t /= Hp; const int d = (int)(t % Dp); const int b = (int)(t / Dp);
    dst[i] = (d < D && h < H && w < W) ? src[((((int64_t)b * D + d) * H + h) * W + w) * C + c] : 0.f;
  }
}
__global__ void __launch_bounds__(256) crop3d_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int D, int H,
                                                     int W, int Dp, int Hp, int Wp, int C, int accumulate, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i / C; const int c = (int)(i - t * C);
    const int w = (int)(t % W); t /= W; const int h = (int)(t % H); t /= H; const int d = (int)(t % D); const int b = (int)(t / D);
    const float v = src[((((int64_t)b * Dp + d) * Hp + h) * Wp + w) * C + c];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

// align_corners=True: src = dst * (S_in - 1) / (S_out - 1)   (0 when S_out == 1)
__device__ __forceinline__ void ac_coord(int o, int Sin, int Sout, int& i0, int& i1, float& f) {
  const float scale = Sout > 1 ? (float)(Sin - 1) / (float)(Sout - 1) : 0.f;
  const float s = scale * (float)o;
  i0 = (int)s; if (i0 > Sin - 1) i0 = Sin - 1;
  i1 = i0 + (i0 < Sin - 1 ? 1 : 0);
  f = s - (float)i0;
}
template <bool BWD>
__global__ void __launch_bounds__(256) resize_kernel(const float* __restrict__ a, float* __restrict__ o, int B, int D, int H, int W,
                                                     int Do, int Ho, int Wo, int C, int64_t total) {
  // FWD: a = src [B,D,H,W,C], o = dst [B,Do,Ho,Wo,C];   BWD: a = d(dst), o = d(src) (atomic scatter, pre-zeroed)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i / C; const int c = (int)(i - t * C);
    const int w = (int)(t % Wo); t /= Wo; const int h = (int)(t % Ho); t /= Ho; const int d = (int)(t % Do); const int b = (int)(t / Do);
    int d0, d1, h0, h1, w0, w1; float fd, fh, fw;
    ac_coord(d, D, Do, d0, d1, fd); ac_coord(h, H, Ho, h0, h1, fh); ac_coord(w, W, Wo, w0, w1, fw);
    float acc = 0.f;
    const float gv = BWD ? a[i] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int dd = (q & 4) ? d1 : d0, hh = (q & 2) ? h1 : h0, ww = (q & 1) ? w1 : w0;
      const float wt = ((q & 4) ? fd : 1.f - fd) * ((q & 2) ? fh : 1.f - fh) * ((q & 1) ? fw : 1.f - fw);
      const int64_t s = ((((int64_t)b * D + dd) * H + hh) * W + ww) * C + c;
      if (BWD) atomicAdd(o + s, wt * gv); else acc += wt * a[s];
    }
    if (!BWD) o[i] = acc;
  }
}


// ---- sliding-window inference (utils.py:226-234: monai sliding_window_inference, mode="constant"): accumulate one window's
// logits into the full-volume fp32 sum and bump the per-voxel visit count; then out = sum / count.
__global__ void __launch_bounds__(256) sw_window_kernel(const float* __restrict__ vol, float* __restrict__ win, int C, int D, int H,
                                                        int W, int rd, int rh, int rw, int z0, int y0, int x0, int64_t total) {
  // win[c, z, y, x] = vol[c, z0+z, y0+y, x0+x]   (one batch element; NCDHW)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % rw); int64_t r = i / rw;
    const int y = (int)(r % rh); r /= rh;
    const int z = (int)(r % rd); const int c = (int)(r / rd);
    win[i] = vol[(((int64_t)c * D + z0 + z) * H + y0 + y) * W + x0 + x];
  }
}
__global__ void __launch_bounds__(256) sw_accumulate_kernel(const float* __restrict__ pred, float* __restrict__ out,
                                                            float* __restrict__ count, int K, int D, int H, int W, int rd, int rh,
                                                            int rw, int z0, int y0, int x0, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % rw); int64_t r = i / rw;
    const int y = (int)(r % rh); r /= rh;
    const int z = (int)(r % rd); const int k = (int)(r / rd);
    const int64_t v = ((int64_t)(z0 + z) * H + y0 + y) * W + x0 + x;
    out[(int64_t)k * D * H * W + v] += pred[i];
    if (k == 0) count[v] += 1.f;
  }
}
__global__ void __launch_bounds__(256) sw_normalize_kernel(float* __restrict__ out, const float* __restrict__ count, int64_t V,
                                                           int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] /= count[i % V];
}

}  // namespace micf
using namespace micf;

extern "C" int micf_abi_version(void) { return MICF_ABI_VERSION; }

micf::Options& micf::options() {
  static Options o;
  return o;
}

static int* option_slot(const char* name) {
  if (!name) return nullptr;
  Options& o = micf::options();
  const struct { const char* n; int* v; } table[] = {
      {"block_wave", &o.block_wave}, {"block_recompute_h", &o.block_recompute_h}, {"block_debug", &o.block_debug},
      {"sample_tile", &o.sample_tile}, {"sample_e", &o.sample_e}, {"cell_cap", &o.cell_cap},
      {"tile_cap_hits", &o.tile_cap_hits}, {"tile_cap_cell", &o.tile_cap_cell}, {"tile_cap_voxel", &o.tile_cap_voxel}};
  for (const auto& e : table)
    if (!strcmp(name, e.n)) return e.v;
  return nullptr;
}

extern "C" int micf_set_option(const char* name, int value) {
  int* slot = option_slot(name);
  if (!slot) return MICF_EINVAL;
  *slot = value;
  return MICF_OK;
}

extern "C" int micf_get_option(const char* name, int* value) {
  const int* slot = option_slot(name);
  if (!slot || !value) return MICF_EINVAL;
  *value = *slot;
  return MICF_OK;
}

extern "C" const char* micf_strerror(int code) {
  switch (code) {
    case MICF_OK: return "ok";
    case MICF_EINVAL: return "invalid argument (null pointer, bad size or misaligned buffer)";
    case MICF_EUNSUPPORTED: return "shape not supported by the HIP kernels";
    case MICF_ELAUNCH: return "HIP kernel launch failed";
    default: return "unknown micf error code";
  }
}

static int grid_for(int64_t total) { int64_t b = (total + 255) / 256; return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }

extern "C" int micf_pad3d(const float* src, float* dst, int B, int D, int H, int W, int Dp, int Hp, int Wp, int C,
                          micf_stream_t stream) {
  if (!src || !dst || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || Dp < D || Hp < H || Wp < W) return MICF_EINVAL;
  const int64_t total = (int64_t)B * Dp * Hp * Wp * C;
  hipLaunchKernelGGL(pad3d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, B, D, H, W, Dp, Hp, Wp, C, total);
  MICF_RETURN_LAUNCH();
}
extern "C" int micf_crop3d(const float* src, float* dst, int B, int D, int H, int W, int Dp, int Hp, int Wp, int C, int accumulate,
                           micf_stream_t stream) {
  if (!src || !dst || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || Dp < D || Hp < H || Wp < W) return MICF_EINVAL;
  const int64_t total = (int64_t)B * D * H * W * C;
  hipLaunchKernelGGL(crop3d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, B, D, H, W, Dp, Hp, Wp, C,
                     accumulate, total);
  MICF_RETURN_LAUNCH();
}
extern "C" int micf_resize_trilinear_fwd(const float* src, float* dst, int B, int D, int H, int W, int Do, int Ho, int Wo, int C,
                                         micf_stream_t stream) {
  if (!src || !dst || B <= 0 || D <= 0 || H <= 0 || W <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || C <= 0) return MICF_EINVAL;
  const int64_t total = (int64_t)B * Do * Ho * Wo * C;
  hipLaunchKernelGGL(resize_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, B, D, H, W, Do, Ho, Wo, C, total);
  MICF_RETURN_LAUNCH();
}
extern "C" int micf_resize_trilinear_bwd(const float* ddst, float* dsrc, int B, int D, int H, int W, int Do, int Ho, int Wo, int C,
                                         micf_stream_t stream) {
  if (!ddst || !dsrc || B <= 0 || D <= 0 || H <= 0 || W <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || C <= 0) return MICF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dsrc, 0, sizeof(float) * (size_t)B * D * H * W * C, s) != hipSuccess) return MICF_ELAUNCH;
  const int64_t total = (int64_t)B * Do * Ho * Wo * C;
  hipLaunchKernelGGL(resize_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, ddst, dsrc, B, D, H, W, Do, Ho, Wo, C, total);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_sw_window(const float* vol, float* win, int C, int D, int H, int W, int rd, int rh, int rw, int z0, int y0,
                              int x0, micf_stream_t stream) {
  if (!vol || !win || C <= 0 || rd <= 0 || rh <= 0 || rw <= 0 || z0 < 0 || y0 < 0 || x0 < 0 || z0 + rd > D || y0 + rh > H ||
      x0 + rw > W)
    return MICF_EINVAL;
  const int64_t total = (int64_t)C * rd * rh * rw;
  hipLaunchKernelGGL(sw_window_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, vol, win, C, D, H, W, rd, rh, rw, z0,
                     y0, x0, total);
  MICF_RETURN_LAUNCH();
}
extern "C" int micf_sw_accumulate(const float* pred, float* out, float* count, int K, int D, int H, int W, int rd, int rh, int rw,
                                  int z0, int y0, int x0, micf_stream_t stream) {
  if (!pred || !out || !count || K <= 0 || rd <= 0 || rh <= 0 || rw <= 0 || z0 < 0 || y0 < 0 || x0 < 0 || z0 + rd > D ||
      y0 + rh > H || x0 + rw > W)
    return MICF_EINVAL;
  const int64_t total = (int64_t)K * rd * rh * rw;
  hipLaunchKernelGGL(sw_accumulate_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, pred, out, count, K, D, H, W, rd,
                     rh, rw, z0, y0, x0, total);
  MICF_RETURN_LAUNCH();
}
extern "C" int micf_sw_normalize(float* out, const float* count, int K, int64_t V, micf_stream_t stream) {
  if (!out || !count || K <= 0 || V <= 0) return MICF_EINVAL;
  const int64_t total = (int64_t)K * V;
  hipLaunchKernelGGL(sw_normalize_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, out, count, V, total);
  MICF_RETURN_LAUNCH();
}

// ---- step plumbing
extern "C" int micf_zero(void* p, int64_t bytes, micf_stream_t stream) {
  if (!p || bytes < 0) return MICF_EINVAL;
  if (bytes == 0) return MICF_OK;
  return hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream) == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

namespace micf {
__device__ __forceinline__ uint64_t mix64(uint64_t z) {      // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) drop_path_draw_kernel(uint64_t* __restrict__ rng, const float* __restrict__ keep,
                                                             float* __restrict__ out, int n, int B) {
  const uint64_t seed = rng[0], ctr = rng[1];
  for (int e = threadIdx.x; e < n * B; e += 256) {
    const float k = keep[e / B];
    const uint64_t h = mix64(mix64(seed ^ mix64(ctr)) + (uint64_t)e);
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);          // 24 random bits -> [0, 1)
    out[e] = (u < k) ? 1.0f / k : 0.0f;
  }
  __syncthreads();
  if (threadIdx.x == 0) rng[1] = ctr + 1;
}
}  // namespace micf

extern "C" int micf_drop_path_draw(void* rng, const float* keep, float* out, int n, int B, micf_stream_t stream) {
  if (!rng || !keep || !out || n < 0 || B <= 0) return MICF_EINVAL;
  if (n == 0) return MICF_OK;
  hipLaunchKernelGGL(micf::drop_path_draw_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<uint64_t*>(rng), keep, out, n, B);
  MICF_RETURN_LAUNCH();
}

// ---- grouped weight preparation for the fused block kernels, refreshed once per step: per item ONE read of a row-major fp32
// matrix writes a same-orientation copy and / or a transposed copy, in fp32 or bf16; up to kPrepMax matrices per launch,
// 64 x 64 tiles through LDS, 16-byte global accesses when the shapes allow (all MicFormer weights: multiples of 16).
// bf16 = round-to-nearest-even, the same rounding the bf16 GEMM kernels apply at fragment read.
namespace micf {
constexpr int kPrepMax = 64;
struct PrepArgs {
  int n;
  int end[kPrepMax];                       // running total of 64 x 64 tiles
  const float* src[kPrepMax]; void* dst[kPrepMax]; void* dst_t[kPrepMax];
  int rows[kPrepMax], cols[kPrepMax], bf16[kPrepMax];
};
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // round-to-nearest-even, a in the low half (as gemm_dma.h)
  // gfx950 has the conversion in hardware: v_cvt_pk_bf16_f32 (round-to-nearest-even), ONE instruction for the pair -- the
  // integer emulation (add 0x7FFF + lsb, shift, merge) was ~7 VALU instructions per pair and the bound of every bf16 kernel
  typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
  typedef float f32x2_hw __attribute__((ext_vector_type(2)));
  const f32x2_hw f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_hw));
}
__device__ __forceinline__ void prep_store4(void* base, int64_t idx, bool bf16, bool vec, int valid, float4 v) {
  if (bf16) {
    uint16_t* d = static_cast<uint16_t*>(base) + idx;
    if (vec) { *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)); return; }
    const float e[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < valid; ++i) d[i] = (uint16_t)(pack_bf16(e[i], 0.f) & 0xFFFFu);
  } else {
    float* d = static_cast<float*>(base) + idx;
    if (vec) { *reinterpret_cast<float4*>(d) = v; return; }
    const float e[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < valid; ++i) d[i] = e[i];
  }
}
__global__ void __launch_bounds__(256) weight_prep_kernel(const PrepArgs a) {
  __shared__ float t[64][65];
  const int w = blockIdx.x;
  int k = 0;
  while (k < a.n - 1 && w >= a.end[k]) ++k;
  const int local = w - (k ? a.end[k - 1] : 0);
  const int rows = a.rows[k], cols = a.cols[k];
  const bool bf = a.bf16[k] == 1 || a.bf16[k] == 2, blocked = a.bf16[k] >= 2;
  // K16-blocked destination index of element (r, c) of an [R, Cn] matrix (4 consecutive c stay consecutive)
  auto at = [&](int r, int c, int Cn) {
    return blocked ? ((int64_t)(r >> 4) * (Cn >> 4) + (c >> 4)) * 256 + (r & 15) * 16 + (c & 15) : (int64_t)r * Cn + c;
  };
  const int tc = (cols + 63) >> 6;
  const int r0 = (local / tc) * 64, c0 = (local % tc) * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 float4 columns x 16 rows per pass
  const float* __restrict__ src = a.src[k];
  const bool vc = (cols & 3) == 0, vr = (rows & 3) == 0;
#pragma unroll
  for (int j = 0; j < 64; j += 16) {
    const int r = r0 + ty + j, c = c0 + 4 * tx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int valid = r < rows ? (cols - c < 4 ? cols - c : 4) : 0;
    if (valid == 4 && vc) v = *reinterpret_cast<const float4*>(src + (int64_t)r * cols + c);
    else if (valid > 0) {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < valid; ++i) e[i] = src[(int64_t)r * cols + c + i];
      v = make_float4(e[0], e[1], e[2], e[3]);
    }
    if (a.dst[k] && valid > 0) prep_store4(a.dst[k], at(r, c, cols), bf, valid == 4 && vc, valid, v);
    t[ty + j][4 * tx] = v.x; t[ty + j][4 * tx + 1] = v.y; t[ty + j][4 * tx + 2] = v.z; t[ty + j][4 * tx + 3] = v.w;
  }
  if (!a.dst_t[k]) return;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 64; j += 16) {
    const int c = c0 + ty + j, r = r0 + 4 * tx;                  // transposed row c, its columns r .. r + 3
    const int valid = c < cols ? (rows - r < 4 ? rows - r : 4) : 0;
    if (valid <= 0) continue;
    const float4 v = make_float4(t[4 * tx][ty + j], t[4 * tx + 1][ty + j], t[4 * tx + 2][ty + j], t[4 * tx + 3][ty + j]);
    prep_store4(a.dst_t[k], at(c, r, rows), bf, valid == 4 && vr, valid, v);
  }
}
}  // namespace micf

extern "C" int micf_weight_prep_grouped(const micf_weight_prep_item* items, int n, micf_stream_t stream) {
  if (n < 0 || (n > 0 && !items)) return MICF_EINVAL;
  for (int first = 0; first < n; first += micf::kPrepMax) {
    const int cnt = (n - first < micf::kPrepMax) ? n - first : micf::kPrepMax;
    micf::PrepArgs a;
    a.n = cnt;
    int blocks = 0;
    for (int k = 0; k < cnt; ++k) {
      const micf_weight_prep_item& it = items[first + k];
      const int align = (it.bf16 == 1 || it.bf16 == 2) ? 7 : 15;
      if (it.bf16 < 0 || it.bf16 > 3 || (it.bf16 >= 2 && ((it.rows | it.cols) & 15))) return MICF_EINVAL;
      if (!it.src || (!it.dst && !it.dst_t) || it.rows <= 0 || it.cols <= 0 || (reinterpret_cast<uintptr_t>(it.src) & 15) ||
          (reinterpret_cast<uintptr_t>(it.dst) & align) || (reinterpret_cast<uintptr_t>(it.dst_t) & align))
        return MICF_EINVAL;
      a.src[k] = it.src; a.dst[k] = it.dst; a.dst_t[k] = it.dst_t; a.rows[k] = it.rows; a.cols[k] = it.cols; a.bf16[k] = it.bf16;
      blocks += ((it.rows + 63) / 64) * ((it.cols + 63) / 64);
      a.end[k] = blocks;
    }
    hipLaunchKernelGGL(micf::weight_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
  }
  return MICF_OK;
}

// ---- sliding-window inference, batched: one launch crops `n` windows, one launch accumulates `n` predictions (windows of a
// batch may overlap: fp32 atomics; the sum over <= 8 visits is order-independent to an ulp)
namespace micf {
constexpr int kSwMax = 64;
struct SwCoords { int n; int b[kSwMax], z[kSwMax], y[kSwMax], x[kSwMax]; };
__global__ void __launch_bounds__(256) sw_window_batch_kernel(const float* __restrict__ vol, float* __restrict__ win, SwCoords c, int C,
                                                              int D, int H, int W, int rd, int rh, int rw, int64_t per) {
  const int n = blockIdx.y;
  const float* v = vol + (int64_t)c.b[n] * C * D * H * W;
  float* o = win + (int64_t)n * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % rw); int64_t r = i / rw;
    const int y = (int)(r % rh); r /= rh;
    const int z = (int)(r % rd); const int ch = (int)(r / rd);
    o[i] = v[(((int64_t)ch * D + c.z[n] + z) * H + c.y[n] + y) * W + c.x[n] + x];
  }
}
__global__ void __launch_bounds__(256) sw_accumulate_batch_kernel(const float* __restrict__ pred, float* __restrict__ out,
                                                                  float* __restrict__ count, SwCoords c, int K, int D, int H, int W,
                                                                  int rd, int rh, int rw, int64_t per) {
  const int n = blockIdx.y;
  const float* p = pred + (int64_t)n * per;
  float* o = out + (int64_t)c.b[n] * K * D * H * W;
  float* cn = count + (int64_t)c.b[n] * D * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % rw); int64_t r = i / rw;
    const int y = (int)(r % rh); r /= rh;
    const int z = (int)(r % rd); const int k = (int)(r / rd);
    const int64_t v = ((int64_t)(c.z[n] + z) * H + c.y[n] + y) * W + c.x[n] + x;
    atomicAdd(o + (int64_t)k * D * H * W + v, p[i]);
    if (k == 0) atomicAdd(cn + v, 1.f);
  }
}
static bool sw_coords(SwCoords& c, const int32_t* coords, int n, int B, int D, int H, int W, int rd, int rh, int rw) {
  if (!coords || n <= 0 || n > kSwMax) return false;
  c.n = n;
  for (int i = 0; i < n; ++i) {
    c.b[i] = coords[4 * i]; c.z[i] = coords[4 * i + 1]; c.y[i] = coords[4 * i + 2]; c.x[i] = coords[4 * i + 3];
    if (c.b[i] < 0 || c.b[i] >= B || c.z[i] < 0 || c.y[i] < 0 || c.x[i] < 0 || c.z[i] + rd > D || c.y[i] + rh > H || c.x[i] + rw > W) return false;
  }
  return true;
}
}  // namespace micf

extern "C" int micf_sw_window_batch(const float* vol, float* win, const int32_t* coords, int n, int B, int C, int D, int H, int W,
                                    int rd, int rh, int rw, micf_stream_t stream) {
  micf::SwCoords c;
  if (!vol || !win || C <= 0 || rd <= 0 || rh <= 0 || rw <= 0 || !micf::sw_coords(c, coords, n, B, D, H, W, rd, rh, rw)) return MICF_EINVAL;
  const int64_t per = (int64_t)C * rd * rh * rw;
  hipLaunchKernelGGL(micf::sw_window_batch_kernel, dim3(grid_for(per) > 2048 ? 2048 : grid_for(per), n), dim3(256), 0, (hipStream_t)stream,
                     vol, win, c, C, D, H, W, rd, rh, rw, per);
  MICF_RETURN_LAUNCH();
}
extern "C" int micf_sw_accumulate_batch(const float* pred, float* out, float* count, const int32_t* coords, int n, int B, int K, int D,
                                        int H, int W, int rd, int rh, int rw, micf_stream_t stream) {
  micf::SwCoords c;
  if (!pred || !out || !count || K <= 0 || rd <= 0 || rh <= 0 || rw <= 0 || !micf::sw_coords(c, coords, n, B, D, H, W, rd, rh, rw))
    return MICF_EINVAL;
  const int64_t per = (int64_t)K * rd * rh * rw;
  hipLaunchKernelGGL(micf::sw_accumulate_batch_kernel, dim3(grid_for(per) > 2048 ? 2048 : grid_for(per), n), dim3(256), 0,
                     (hipStream_t)stream, pred, out, count, c, K, D, H, W, rd, rh, rw, per);
  MICF_RETURN_LAUNCH();
}

// ---- input-pipeline tail (train.py:116-125: RandFlipd x3 on image + label, NormalizeIntensityd(nonzero, channel_wise),
// RandScaleIntensityd(0.1), RandShiftIntensityd(0.1)) and the float16 -> float32 cast of the loader (MMWHS.py:386, train.py:177)
namespace micf {
template <class T> __device__ __forceinline__ float ldv(const T* p, int64_t i);
template <> __device__ __forceinline__ float ldv<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldv<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }

// sums[(b*Cm + m)*3 + {0,1,2}] = {sum, sum of squares, count} of the NON-ZERO voxels of channel m of sample b (doubles)
template <class T>
__global__ void __launch_bounds__(256) intensity_stats_kernel(const T* __restrict__ vol, double* __restrict__ sums, int64_t V, int chunks) {
  const int plane = blockIdx.y;
  const int64_t per = (V + chunks - 1) / chunks;
  const int64_t v0 = blockIdx.x * per, v1 = (v0 + per < V) ? v0 + per : V;
  const T* p = vol + (int64_t)plane * V;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int64_t i = v0 + threadIdx.x; i < v1; i += 256) {
    const float x = ldv<T>(p, i);
    if (x != 0.f) { a += x; b += x * x; c += 1.f; }
  }
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  __shared__ float part[4][3];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { part[wave][0] = a; part[wave][1] = b; part[wave][2] = c; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int q = threadIdx.x;
    atomicAdd(sums + plane * 3 + q, (double)part[0][q] + (double)part[1][q] + (double)part[2][q] + (double)part[3][q]);
  }
}
// out[b, m, d, h, w] = ((raw[flip(d,h,w)] - mean)/std if raw != 0 else 0) * (1 + f_b) + o_b ; label_out = label_in[flip]
// params [B, 5] = {flipD, flipH, flipW (0 / 1), scale factor f, shift offset o}; params == NULL: no flips, f = o = 0 (validation)
template <class T>
__global__ void __launch_bounds__(256) input_prepare_kernel(const T* __restrict__ vol, const double* __restrict__ sums,
                                                            const float* __restrict__ params, float* __restrict__ out,
                                                            const uint8_t* __restrict__ lab_in, uint8_t* __restrict__ lab_out, int Cm, int D,
                                                            int H, int W) {
  const int b = blockIdx.y;
  const int64_t V = (int64_t)D * H * W;
  int fd = 0, fh = 0, fw = 0;
  float f = 0.f, o = 0.f;
  if (params) { fd = params[b * 5] != 0.f; fh = params[b * 5 + 1] != 0.f; fw = params[b * 5 + 2] != 0.f; f = params[b * 5 + 3]; o = params[b * 5 + 4]; }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < V; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % W); int64_t r = i / W;
    const int h = (int)(r % H); const int d = (int)(r / H);
    const int64_t src = ((int64_t)(fd ? D - 1 - d : d) * H + (fh ? H - 1 - h : h)) * W + (fw ? W - 1 - w : w);
    for (int m = 0; m < (out ? Cm : 0); ++m) {
      const double* sp = sums + ((int64_t)b * Cm + m) * 3;
      const double n = sp[2];
      const double mean = n > 0 ? sp[0] / n : 0.0;
      double var = n > 0 ? sp[1] / n - mean * mean : 0.0;
      if (var < 0) var = 0;
      double sd = sqrt(var);
      if (sd == 0.0) sd = 1.0;                                  // MONAI: a constant image is only shifted
      const float x = ldv<T>(vol, ((int64_t)b * Cm + m) * V + src);
      float y = x != 0.f ? (float)(((double)x - mean) / sd) : 0.f;
      out[((int64_t)b * Cm + m) * V + i] = y * (1.f + f) + o;
    }
    if (lab_in) lab_out[(int64_t)b * V + i] = lab_in[(int64_t)b * V + src];
  }
}
// The same arithmetic fused into patch embedding's gather (SURVEY 8(f) row 3 as written): rows[m][r, tap] of modality m =
// prepared value of the fine voxel (k zc + tz, k yc + ty, k xc + tx), r = ((b Dc + zc) Hc + yc) Wc + xc -- the [tokens, k^3] matrix
// the patch-embedding GEMM reads (patch.hip's space-to-depth layout), straight from the RAW volume: the flips are index
// arithmetic, normalise / scale / shift one affine map per (sample, channel); the float32 volume is never written.
// Voxels beyond the volume (right padding to a multiple of k, MS.py:866-872) are zero, as F.pad leaves them.
template <class T>
__global__ void __launch_bounds__(256) patch_rows_prepared_kernel(const T* __restrict__ vol, const double* __restrict__ sums,
                                                                  const float* __restrict__ params, float* __restrict__ rows0,
                                                                  float* __restrict__ rows1, int B, int Cm, int D, int H, int W, int k,
                                                                  int Dc, int Hc, int Wc, int64_t total4) {
  const int m = blockIdx.y;
  float* rows = m == 0 ? rows0 : rows1;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4 || !rows) return;
  const int K3 = k * k * k, q4 = K3 >> 2;
  const int t4 = (int)(i % q4);
  int64_t r = i / q4;
  const int xc = (int)(r % Wc); int64_t r2 = r / Wc;
  const int yc = (int)(r2 % Hc); r2 /= Hc;
  const int zc = (int)(r2 % Dc); const int b = (int)(r2 / Dc);
  int fd = 0, fh = 0, fw = 0;
  float f = 0.f, o = 0.f;
  if (params) { fd = params[b * 5] != 0.f; fh = params[b * 5 + 1] != 0.f; fw = params[b * 5 + 2] != 0.f; f = params[b * 5 + 3]; o = params[b * 5 + 4]; }
  const double* sp = sums + ((int64_t)b * Cm + m) * 3;
  const double n = sp[2];
  const double mean = n > 0 ? sp[0] / n : 0.0;
  double var = n > 0 ? sp[1] / n - mean * mean : 0.0;
  if (var < 0) var = 0;
  double sd = sqrt(var);
  if (sd == 0.0) sd = 1.0;
  const int64_t V = (int64_t)D * H * W;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tap = 4 * t4 + j;
    const int tz = tap / (k * k), ty = (tap / k) % k, tx = tap % k;
    const int z = zc * k + tz, y = yc * k + ty, x = xc * k + tx;
    if (z >= D || y >= H || x >= W) continue;
    const int64_t src = ((int64_t)(fd ? D - 1 - z : z) * H + (fh ? H - 1 - y : y)) * W + (fw ? W - 1 - x : x);
    const float raw = ldv<T>(vol, ((int64_t)b * Cm + m) * V + src);
    const float yv = raw != 0.f ? (float)(((double)raw - mean) / sd) : 0.f;
    v[j] = yv * (1.f + f) + o;
  }
  *reinterpret_cast<float4*>(rows + r * K3 + 4 * t4) = make_float4(v[0], v[1], v[2], v[3]);
}
}  // namespace micf

extern "C" int micf_patch_rows_prepared(const void* vol, int is_half, const double* sums, const float* params, float* rows0,
                                        float* rows1, int B, int Cm, int D, int H, int W, int k, micf_stream_t stream) {
  if (!vol || !sums || (!rows0 && !rows1) || B <= 0 || Cm < 1 || Cm > 2 || (Cm == 1 && rows1) || D <= 0 || H <= 0 || W <= 0 ||
      (k != 2 && k != 4))
    return MICF_EINVAL;
  if ((reinterpret_cast<uintptr_t>(rows0) | reinterpret_cast<uintptr_t>(rows1)) & 15) return MICF_EINVAL;
  const int Dc = (D + k - 1) / k, Hc = (H + k - 1) / k, Wc = (W + k - 1) / k;
  const int64_t total4 = (int64_t)B * Dc * Hc * Wc * (k * k * k / 4);
  const dim3 grid((unsigned)((total4 + 255) / 256), Cm);
  if (is_half) hipLaunchKernelGGL(micf::patch_rows_prepared_kernel<__half>, grid, dim3(256), 0, (hipStream_t)stream, static_cast<const __half*>(vol), sums, params, rows0, rows1, B, Cm, D, H, W, k, Dc, Hc, Wc, total4);
  else hipLaunchKernelGGL(micf::patch_rows_prepared_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(vol), sums, params, rows0, rows1, B, Cm, D, H, W, k, Dc, Hc, Wc, total4);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_intensity_stats(const void* vol, int is_half, double* sums, int B, int Cm, int64_t V, micf_stream_t stream) {
  if (!vol || !sums || B <= 0 || Cm <= 0 || V <= 0) return MICF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, sizeof(double) * 3 * B * Cm, s) != hipSuccess) return MICF_ELAUNCH;
  int chunks = (int)((V + 65535) / 65536);
  if (chunks > 256) chunks = 256;
  if (is_half) hipLaunchKernelGGL(micf::intensity_stats_kernel<__half>, dim3(chunks, B * Cm), dim3(256), 0, s, static_cast<const __half*>(vol), sums, V, chunks);
  else hipLaunchKernelGGL(micf::intensity_stats_kernel<float>, dim3(chunks, B * Cm), dim3(256), 0, s, static_cast<const float*>(vol), sums, V, chunks);
  MICF_RETURN_LAUNCH();
}
extern "C" int micf_input_prepare(const void* vol, int is_half, const double* sums, const float* params, float* out,
                                  const uint8_t* label_in, uint8_t* label_out, int B, int Cm, int D, int H, int W, micf_stream_t stream) {
  // (out == NULL with a label map: flip the labels only -- the image then goes through micf_patch_rows_prepared)
  if (!vol || !sums || (!out && !label_in) || B <= 0 || Cm <= 0 || D <= 0 || H <= 0 || W <= 0 || (label_in && !label_out)) return MICF_EINVAL;
  const int64_t V = (int64_t)D * H * W;
  const dim3 grid(grid_for(V) > 1024 ? 1024 : grid_for(V), B);
  if (is_half) hipLaunchKernelGGL(micf::input_prepare_kernel<__half>, grid, dim3(256), 0, (hipStream_t)stream, static_cast<const __half*>(vol), sums, params, out, label_in, label_out, Cm, D, H, W);
  else hipLaunchKernelGGL(micf::input_prepare_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(vol), sums, params, out, label_in, label_out, Cm, D, H, W);
  MICF_RETURN_LAUNCH();
}
