// patch.hip -- the stride == kernel convolutions of the encoder/decoder as per-patch GEMMs on the fp32 MFMA core:
//   PatchEmbed3D      Conv3d(1->E, k=s=4) on the raw volume, right zero-pad        (MS.py:854, 860-878)
//   PatchMerging      Conv3d(C->2C, k=s=2), odd dims zero-padded                    (MS.py:539, 548-557)
//   PatchExpand       ConvTranspose3d(C->C/2, k=s=2)                                (MS.py:568, 575-577)
//   reverse_patch_embedding  ConvTranspose3d(2E->E/2, k=s=4)                        (MS.py:990, 1037)
// Patch gather / pixel-shuffle scatter are accessor / epilogue index math on channels-last tensors; weights stay in
// the reference's state_dict layout and are read through strided accessors (they are small and L2-resident).
// Orientation (gemm_core.h): I = the channel axis that is contiguous in the output, J = coarse tokens.
#include "common.h"

namespace micf {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }

// ---- geometry of a k-strided patch grid: coarse (B, Dc, Hc, Wc) <-> fine (B, D, H, W), fine = coarse*k + tap
struct PatchGeo {
  int B, D, H, W;        // fine dims (actual, may be smaller than Dc*k: zero padding at the far end)
  int Dc, Hc, Wc, k;
  FastDiv fWc, fHc, fDc;
  __device__ __forceinline__ void cdecode(int t, int& b, int& d, int& h, int& w) const {
    uint32_t q, r;
    fWc.divmod((uint32_t)t, q, r); w = (int)r;
    fHc.divmod(q, q, r); h = (int)r;
    fDc.divmod(q, q, r); d = (int)r; b = (int)q;
  }
  // fine voxel/token index of (coarse token, tap) or -1 when it falls in the zero padding (k is 2 or 4)
  __device__ __forceinline__ int64_t fine(int tc, int tap) const {
    int b, d, h, w; cdecode(tc, b, d, h, w);
    int kd, kh, kw;
    if (k == 2) { kd = tap >> 2; kh = (tap >> 1) & 1; kw = tap & 1; }
    else { kd = tap >> 4; kh = (tap >> 2) & 3; kw = tap & 3; }
    const int fd = d * k + kd, fh = h * k + kh, fw = w * k + kw;
    if (fd >= D || fh >= H || fw >= W) return -1;
    return (((int64_t)b * D + fd) * H + fh) * W + fw;
  }
};

// ---------------- patch embed: vol [B, nmod, D, H, W]; (x = coarse token, r = tap) -> voxel
struct EmbedVol {
  const float* vol; int nmod, mod; PatchGeo g; int64_t DHW;
  __device__ __forceinline__ float operator()(int x, int r) const {
    const int64_t f = g.fine(x, r);
    if (f < 0) return 0.f;
    const int64_t b = f / DHW;       // fine() indexes (b, d, h, w); re-base onto the modality plane
    return vol[(b * nmod + mod) * DHW + (f - b * DHW)];
  }
};
struct EmbedVolT {   // (x = tap, r = coarse token)
  EmbedVol e;
  __device__ __forceinline__ float operator()(int x, int r) const { return e(r, x); }
};
struct StoreRowsEpi {   // out[token j, feature i .. i+3] (+ bias)
  const float* bias; float* y; int N, vec;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    float* p = y + (int64_t)j * N + i;
    if (vec && n == 4) {
      if (bias) { const float4 b = ld4(bias + i); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
      st4(p, v[0], v[1], v[2], v[3]);
    } else {
      MICF_FOR_N(n, e) p[e] = v[e] + (bias ? bias[i + e] : 0.f);
    }
  }
};
struct AtomicRowsEpi {  // out[j, i] += v
  float* out; int64_t ld;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    float* p = out + (int64_t)j * ld + i;
    MICF_FOR_N(n, e) atomicAdd(p + e, v[e]);
  }
};

// ---------------- conv_down (k = 2): x [B,D,H,W,C] -> y [B,Dc,Hc,Wc,N];  w [N][C][tap]
struct DownIn {       // (x = coarse token, r = tap*C + c) -> x_fine[token(tc,tap), c]
  const float* x; int C; PatchGeo g; FastDiv fC;
  __device__ __forceinline__ float operator()(int tc, int r) const {
    uint32_t tap, c; fC.divmod((uint32_t)r, tap, c);
    const int64_t f = g.fine(tc, (int)tap);
    return f < 0 ? 0.f : x[f * C + c];
  }
};
struct DownInT {      // (x = tap*C + c, r = coarse token)
  DownIn d;
  __device__ __forceinline__ float operator()(int x, int r) const { return d(r, x); }
};
struct DownW {        // (x = n, r = tap*C + c) -> w[n][c][tap]
  const float* w; int C, K3; FastDiv fC;
  __device__ __forceinline__ float operator()(int n, int r) const {
    uint32_t tap, c; fC.divmod((uint32_t)r, tap, c);
    return w[((int64_t)n * C + c) * K3 + tap];
  }
};
struct DownWT {       // (x = tap*C + c, r = n)
  DownW q;
  __device__ __forceinline__ float operator()(int x, int r) const { return q(r, x); }
};
struct DownDataEpi {  // (i = tap*C + c, j = coarse token) -> dx_fine[token(j, tap), c .. c+3]
  float* dx; int C; PatchGeo g; FastDiv fC; int vec;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    if (vec && n == 4) {             // C % 4 == 0: one tap for the 4 channels
      uint32_t tap, c; fC.divmod((uint32_t)i, tap, c);
      const int64_t f = g.fine(j, (int)tap);
      if (f >= 0) st4(dx + f * C + c, v[0], v[1], v[2], v[3]);
    } else {
      MICF_FOR_N(n, e) {
        uint32_t tap, c; fC.divmod((uint32_t)(i + e), tap, c);
        const int64_t f = g.fine(j, (int)tap);
        if (f >= 0) dx[f * C + c] = v[e];
      }
    }
  }
};
struct DownWgtEpi {   // (i = tap*C + c, j = n) -> dw[n][c][tap]
  float* dw; int C, K3; FastDiv fC;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    MICF_FOR_N(n, e) {
      uint32_t tap, c; fC.divmod((uint32_t)(i + e), tap, c);
      atomicAdd(dw + ((int64_t)j * C + c) * K3 + tap, v[e]);
    }
  }
};

// ---------------- conv_up (k in {2,4}): x [B,Dc,Hc,Wc,C] -> y [B,D,H,W,N] (D = k*Dc ...);  w [C][N][tap]
struct UpW {          // (x = tap*N + n, r = c) -> w[c][n][tap]
  const float* w; int N, K3; FastDiv fN;
  __device__ __forceinline__ float operator()(int x, int r) const {
    uint32_t tap, n; fN.divmod((uint32_t)x, tap, n);
    return w[((int64_t)r * N + n) * K3 + tap];
  }
};
struct UpWT {         // (x = c, r = tap*N + n)
  UpW q;
  __device__ __forceinline__ float operator()(int x, int r) const { return q(r, x); }
};
struct UpFwdEpi {     // (i = tap*N + n, j = coarse token) -> y_fine[token(j, tap), n .. n+3] + bias
  const float* bias; float* y; int N; PatchGeo g; FastDiv fN; int vec;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n4) const {
    if (vec && n4 == 4) {
      uint32_t tap, n; fN.divmod((uint32_t)i, tap, n);
      const int64_t f = g.fine(j, (int)tap);
      if (f < 0) return;
      if (bias) { const float4 b = ld4(bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
      st4(y + f * N + n, v[0], v[1], v[2], v[3]);
    } else {
      MICF_FOR_N(n4, e) {
        uint32_t tap, n; fN.divmod((uint32_t)(i + e), tap, n);
        const int64_t f = g.fine(j, (int)tap);
        if (f >= 0) y[f * N + n] = v[e] + (bias ? bias[n] : 0.f);
      }
    }
  }
};
struct UpDy {         // (x = coarse token, r = tap*N + n) -> dy_fine[token(tc,tap), n]
  const float* dy; int N; PatchGeo g; FastDiv fN;
  __device__ __forceinline__ float operator()(int tc, int r) const {
    uint32_t tap, n; fN.divmod((uint32_t)r, tap, n);
    const int64_t f = g.fine(tc, (int)tap);
    return f < 0 ? 0.f : dy[f * N + n];
  }
};
struct UpDyT {        // (x = tap*N + n, r = coarse token)
  UpDy d;
  __device__ __forceinline__ float operator()(int x, int r) const { return d(r, x); }
};
struct UpWgtEpi {     // (i = tap*N + n, j = c) -> dw[c][n][tap]
  float* dw; int N, K3; FastDiv fN;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n4) const {
    MICF_FOR_N(n4, e) {
      uint32_t tap, n; fN.divmod((uint32_t)(i + e), tap, n);
      atomicAdd(dw + ((int64_t)j * N + n) * K3 + tap, v[e]);
    }
  }
};

static bool mk_geo(PatchGeo& g, int B, int D, int H, int W, int k) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || (k != 2 && k != 4)) return false;
  const int Dc = (D + k - 1) / k, Hc = (H + k - 1) / k, Wc = (W + k - 1) / k;
  g = PatchGeo{B, D, H, W, Dc, Hc, Wc, k, FastDiv((uint32_t)Wc), FastDiv((uint32_t)Hc), FastDiv((uint32_t)Dc)};
  return (int64_t)B * D * H * W < (1LL << 31);
}

}  // namespace micf
using namespace micf;
#define S_(x) ((hipStream_t)(x))
#define RC(e) ((e) == hipSuccess ? MICF_OK : MICF_ELAUNCH)

extern "C" int micf_patch_embed_fwd(const float* vol, int nmod, int mod, const float* w, const float* bias, float* y, int B,
                                    int D, int H, int W, int E, int p, micf_stream_t stream) {
  PatchGeo g;
  if (!vol || !w || !y || E <= 0 || nmod <= 0 || mod < 0 || mod >= nmod || !mk_geo(g, B, D, H, W, p)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * g.Dc * g.Hc * g.Wc;
  const int K = p * p * p;
  // C[i = e, j = coarse token] = sum_tap w[e, tap] * patch[token, tap]
  auto qa = make_elem<true>(EmbedVol{vol, nmod, mod, g, (int64_t)D * H * W}, (int)Tc);
  const int vec = (E % 4 == 0) && aligned16(y) && (!bias || aligned16(bias));
  return RC(launch_gemm(rows_t(w, K, E, K), qa, StoreRowsEpi{bias, y, E, vec}, E, Tc, K, 1, S_(stream)));
}

extern "C" int micf_patch_embed_bwd_weight(const float* dy, const float* vol, int nmod, int mod, float* dw, float* dbias,
                                           int B, int D, int H, int W, int E, int p, micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !vol || !dw || E <= 0 || nmod <= 0 || mod < 0 || mod >= nmod || !mk_geo(g, B, D, H, W, p)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * g.Dc * g.Hc * g.Wc;
  const int K = p * p * p;
  // dW[e, tap] = sum_t dy[t, e] * patch[t, tap]:  C[i = tap, j = e]
  auto pa = make_elem<false>(EmbedVolT{EmbedVol{vol, nmod, mod, g, (int64_t)D * H * W}}, K);
  return RC(launch_gemm(pa, rows_d(dy, E, E), AtomicRowsEpi{dw, K}, K, E, (int)Tc, pick_splits(K, E, Tc), S_(stream), dbias,
                        dbias ? 2 : 0));
}

extern "C" int micf_conv_down_fwd(const float* x, const float* w, const float* bias, float* y, int B, int D, int H, int W,
                                  int C, int N, micf_stream_t stream) {
  PatchGeo g;
  if (!x || !w || !y || C <= 0 || N <= 0 || !mk_geo(g, B, D, H, W, 2)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * g.Dc * g.Hc * g.Wc;
  const FastDiv fC((uint32_t)C);
  // C[i = n, j = coarse token]
  auto pa = make_elem<false>(DownW{w, C, 8, fC}, N);
  auto qa = make_elem<true>(DownIn{x, C, g, fC}, (int)Tc);
  const int vec = (N % 4 == 0) && aligned16(y) && (!bias || aligned16(bias));
  return RC(launch_gemm(pa, qa, StoreRowsEpi{bias, y, N, vec}, N, Tc, 8 * C, 1, S_(stream)));
}

extern "C" int micf_conv_down_bwd_data(const float* dy, const float* w, float* dx, int B, int D, int H, int W, int C, int N,
                                       micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !w || !dx || C <= 0 || N <= 0 || !mk_geo(g, B, D, H, W, 2)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * g.Dc * g.Hc * g.Wc;
  const FastDiv fC((uint32_t)C);
  // dXcols[i = tap*C + c, j = tc] = sum_n w[n][c][tap] * dy[tc, n]  -> scattered to the fine grid (every fine token exactly once)
  auto pa = make_elem<false>(DownWT{DownW{w, C, 8, fC}}, 8 * C);
  const int vec = (C % 4 == 0) && aligned16(dx);
  return RC(launch_gemm(pa, rows_t(dy, N, (int)Tc, N), DownDataEpi{dx, C, g, fC, vec}, 8 * C, Tc, N, 1, S_(stream)));
}

extern "C" int micf_conv_down_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int B, int D, int H, int W,
                                         int C, int N, micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !x || !dw || C <= 0 || N <= 0 || !mk_geo(g, B, D, H, W, 2)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * g.Dc * g.Hc * g.Wc;
  const FastDiv fC((uint32_t)C);
  // dW^T[i = tap*C + c, j = n] = sum_tc x_fine[token(tc,tap), c] * dy[tc, n]
  auto pa = make_elem<false>(DownInT{DownIn{x, C, g, fC}}, 8 * C);
  return RC(launch_gemm(pa, rows_d(dy, N, N), DownWgtEpi{dw, C, 8, fC}, 8 * C, N, (int)Tc, pick_splits(8 * C, N, Tc), S_(stream),
                        dbias, dbias ? 2 : 0));
}

extern "C" int micf_conv_up_fwd(const float* x, const float* w, const float* bias, float* y, int B, int D, int H, int W, int C,
                                int N, int k, micf_stream_t stream) {
  // here (D, H, W) are the COARSE (input) dims; the output is exactly (kD, kH, kW)
  PatchGeo g;
  if (!x || !w || !y || C <= 0 || N <= 0 || (k != 2 && k != 4) || !mk_geo(g, B, D * k, H * k, W * k, k)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * D * H * W;
  const int K3 = k * k * k;
  const FastDiv fN((uint32_t)N);
  // C[i = tap*N + n, j = tc] = sum_c w[c][n][tap] * x[tc, c]
  auto pa = make_elem<false>(UpW{w, N, K3, fN}, K3 * N);
  const int vec = (N % 4 == 0) && aligned16(y) && (!bias || aligned16(bias));
  return RC(launch_gemm(pa, rows_t(x, C, (int)Tc, C), UpFwdEpi{bias, y, N, g, fN, vec}, K3 * N, Tc, C, 1, S_(stream)));
}

extern "C" int micf_conv_up_bwd_data(const float* dy, const float* w, float* dx, int B, int D, int H, int W, int C, int N, int k,
                                     micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !w || !dx || C <= 0 || N <= 0 || (k != 2 && k != 4) || !mk_geo(g, B, D * k, H * k, W * k, k)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * D * H * W;
  const int K3 = k * k * k;
  const FastDiv fN((uint32_t)N);
  // dx[i = c, j = tc] = sum_{tap,n} w[c][n][tap] * dy_fine[token(tc,tap), n]
  auto pa = make_elem<false>(UpWT{UpW{w, N, K3, fN}}, C);
  auto qa = make_elem<true>(UpDy{dy, N, g, fN}, (int)Tc);
  const int vec = (C % 4 == 0) && aligned16(dx);
  return RC(launch_gemm(pa, qa, StoreRowsEpi{nullptr, dx, C, vec}, C, Tc, K3 * N, 1, S_(stream)));
}

extern "C" int micf_conv_up_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int B, int D, int H, int W,
                                       int C, int N, int k, micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !x || !dw || C <= 0 || N <= 0 || (k != 2 && k != 4) || !mk_geo(g, B, D * k, H * k, W * k, k)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * D * H * W;
  const int K3 = k * k * k;
  const FastDiv fN((uint32_t)N);
  // dw^T[i = tap*N + n, j = c] = sum_tc dy_fine[token(tc,tap), n] * x[tc, c]
  auto pa = make_elem<false>(UpDyT{UpDy{dy, N, g, fN}}, K3 * N);
  if (launch_gemm(pa, rows_d(x, C, C), UpWgtEpi{dw, N, K3, fN}, K3 * N, C, (int)Tc, pick_splits(K3 * N, C, Tc), S_(stream)) != hipSuccess)
    return MICF_ELAUNCH;
  if (dbias) return colsum_atomic(dy, nullptr, 1, dbias, Tc * K3, N, S_(stream));
  return MICF_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Space-to-depth / depth-to-space: a convolution whose stride equals its kernel (patch embedding k = 4, PatchMerging k = 2,
// PatchExpand's transposed conv k = 2) is a plain GEMM once the k^3 voxels of every patch sit in one row.  Column order is
// (channel, tap), tap = (tz * k + ty) * k + tx -- the flattening of a Conv3d weight [N, C, k, k, k] -> [N, C k^3] and of a
// ConvTranspose3d weight [C, N, k, k, k] -> [C, N k^3], so the parameters and their gradients are used in place and the
// GEMMs are micf_linear_fwd / _bwd_data / the grouped weight gradient.  (The element-gather GEMMs above stay as the
// reference form of the entry points; at 6 TFLOP/s they were 0.9 ms of the critical chain of the base step.)
namespace micf {
struct S2dGeo {
  int B, D, H, W, C, k, K3;        // fine grid, channels, patch edge
  int Dc, Hc, Wc;                  // coarse grid = ceil(fine / k): out-of-range voxels read as zero / are not written
  int64_t batch_stride;            // elements between the samples of the FINE tensor (voxel stride = C)
};
template <bool TO_DEPTH>
__global__ void __launch_bounds__(256) s2d_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                  const float* __restrict__ bias, S2dGeo g, int64_t total4,
                                                  const float* addp = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int q4 = g.K3 >> 2;
  const int t4 = (int)(i % q4);
  int64_t r = i / q4;
  const int c = (int)(r % g.C); r /= g.C;
  const int xc = (int)(r % g.Wc); int64_t r2 = r / g.Wc;
  const int yc = (int)(r2 % g.Hc); r2 /= g.Hc;
  const int zc = (int)(r2 % g.Dc); const int b = (int)(r2 / g.Dc);
  float* mat = (TO_DEPTH ? dst : const_cast<float*>(src)) + (r * g.C + c) * g.K3 + 4 * t4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (!TO_DEPTH) { const float4 t = *reinterpret_cast<const float4*>(mat); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  const float bs = (!TO_DEPTH && bias) ? bias[c] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tap = 4 * t4 + j;
    const int tz = tap / (g.k * g.k), ty = (tap / g.k) % g.k, tx = tap % g.k;
    const int z = zc * g.k + tz, y = yc * g.k + ty, x = xc * g.k + tx;
    if (z >= g.D || y >= g.H || x >= g.W) continue;
    const int64_t a = (int64_t)b * g.batch_stride + (((int64_t)z * g.H + y) * g.W + x) * g.C + c;
    if (TO_DEPTH) v[j] = src[a];
    else dst[a] = v[j] + bs + (addp ? addp[a] : 0.f);
  }
  if (TO_DEPTH) *reinterpret_cast<float4*>(mat) = make_float4(v[0], v[1], v[2], v[3]);
}
static bool s2d_geo(S2dGeo& g, int B, int D, int H, int W, int C, int k, int64_t batch_stride) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || (k != 2 && k != 4)) return false;
  g.B = B; g.D = D; g.H = H; g.W = W; g.C = C; g.k = k; g.K3 = k * k * k;
  g.Dc = (D + k - 1) / k; g.Hc = (H + k - 1) / k; g.Wc = (W + k - 1) / k;
  g.batch_stride = batch_stride > 0 ? batch_stride : (int64_t)D * H * W * C;
  return true;
}
}  // namespace micf

extern "C" int micf_space_to_depth(const float* x, float* a, int B, int D, int H, int W, int C, int k, int64_t batch_stride,
                                   micf_stream_t stream) {
  S2dGeo g;
  if (!x || !a || !s2d_geo(g, B, D, H, W, C, k, batch_stride) || !aligned16(a)) return MICF_EINVAL;
  const int64_t total4 = (int64_t)B * g.Dc * g.Hc * g.Wc * C * (g.K3 / 4);
  hipLaunchKernelGGL(s2d_kernel<true>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, S_(stream), x, a, nullptr, g, total4);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_depth_to_space(const float* a, const float* bias, float* y, int B, int D, int H, int W, int C, int k,
                                   micf_stream_t stream) {
  S2dGeo g;
  if (!a || !y || !s2d_geo(g, B, D, H, W, C, k, 0) || !aligned16(a)) return MICF_EINVAL;
  const int64_t total4 = (int64_t)B * g.Dc * g.Hc * g.Wc * C * (g.K3 / 4);
  hipLaunchKernelGGL(s2d_kernel<false>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, S_(stream), a, y, bias, g, total4);
  MICF_RETURN_LAUNCH();
}

// ... + add[voxel, c]: a second gradient of the same tensor (the skip connection's) summed in the scatter, add may alias y
extern "C" int micf_depth_to_space_add(const float* a, const float* bias, const float* add, float* y, int B, int D, int H, int W,
                                       int C, int k, micf_stream_t stream) {
  S2dGeo g;
  if (!a || !y || !add || !s2d_geo(g, B, D, H, W, C, k, 0) || !aligned16(a)) return MICF_EINVAL;
  const int64_t total4 = (int64_t)B * g.Dc * g.Hc * g.Wc * C * (g.K3 / 4);
  hipLaunchKernelGGL(s2d_kernel<false>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, S_(stream), a, y, bias, g, total4, add);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_colsum(const float* x, float* out, int64_t M, int N, micf_stream_t stream) {
  if (!x || !out || M < 0 || N <= 0) return MICF_EINVAL;
  if (M == 0) return MICF_OK;
  return colsum_atomic(x, nullptr, 1, out, M, N, S_(stream));
}
