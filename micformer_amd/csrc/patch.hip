// patch.hip -- the stride == kernel convolutions of the encoder/decoder as per-patch GEMMs on the fp32 MFMA core:
//   PatchEmbed3D      Conv3d(1->E, k=s=4) on the raw volume, right zero-pad        (MS.py:854, 860-878)
//   PatchMerging      Conv3d(C->2C, k=s=2), odd dims zero-padded                    (MS.py:539, 548-557)
//   PatchExpand       ConvTranspose3d(C->C/2, k=s=2)                                (MS.py:568, 575-577)
//   reverse_patch_embedding  ConvTranspose3d(2E->E/2, k=s=4)                        (MS.py:990, 1037)
// Patch gather / pixel-shuffle scatter are accessor / epilogue index math on channels-last tensors; weights stay in
// the reference's state_dict layout and are read through strided accessors (they are small and L2-resident).
#include "common.h"

namespace micf {

// ---- geometry of a k-strided patch grid: coarse (B, Dc, Hc, Wc) <-> fine (B, D, H, W), fine = coarse*k + tap
struct PatchGeo {
  int B, D, H, W;        // fine dims (actual, may be smaller than Dc*k: zero padding at the far end)
  int Dc, Hc, Wc, k;
  __device__ __forceinline__ void cdecode(int t, int& b, int& d, int& h, int& w) const {
    w = t % Wc; t /= Wc; h = t % Hc; t /= Hc; d = t % Dc; b = t / Dc;
  }
  // fine voxel/token index of (coarse token, tap) or -1 when it falls in the zero padding
  __device__ __forceinline__ int64_t fine(int tc, int tap) const {
    int b, d, h, w; cdecode(tc, b, d, h, w);
    const int kd = tap / (k * k), kh = (tap / k) % k, kw = tap % k;
    const int fd = d * k + kd, fh = h * k + kh, fw = w * k + kw;
    if (fd >= D || fh >= H || fw >= W) return -1;
    return (((int64_t)b * D + fd) * H + fh) * W + fw;
  }
};

// ---------------- patch embed: vol [B, nmod, D, H, W]; (x = coarse token, r = tap) -> voxel
struct EmbedVol {
  const float* vol; int nmod, mod; PatchGeo g; int64_t DHW;
  __device__ __forceinline__ float operator()(int x, int r) const {
    int b, d, h, w; g.cdecode(x, b, d, h, w);
    const int k = g.k;
    const int fd = d * k + r / (k * k), fh = h * k + (r / k) % k, fw = w * k + r % k;
    if (fd >= g.D || fh >= g.H || fw >= g.W) return 0.f;
    return vol[((int64_t)b * nmod + mod) * DHW + ((int64_t)fd * g.H + fh) * g.W + fw];
  }
};
struct EmbedVolT {   // (x = tap, r = coarse token)
  EmbedVol e;
  __device__ __forceinline__ float operator()(int x, int r) const { return e(r, x); }
};
struct StoreRowsEpi {
  const float* bias; float* y; int N;
  __device__ __forceinline__ void operator()(int i, int j, float v) const { y[(int64_t)i * N + j] = v + (bias ? bias[j] : 0.f); }
};
struct AtomicRowsEpi {
  float* out; int64_t ld;
  __device__ __forceinline__ void operator()(int i, int j, float v) const { atomicAdd(out + (int64_t)i * ld + j, v); }
};

// ---------------- conv_down (k = 2): x [B,D,H,W,C] -> y [B,Dc,Hc,Wc,N];  w [N][C][tap]
struct DownIn {       // (x = coarse token, r = tap*C + c) -> x_fine[token(tc,tap), c]
  const float* x; int C; PatchGeo g;
  __device__ __forceinline__ float operator()(int tc, int r) const {
    const int tap = r / C, c = r - tap * C;
    const int64_t f = g.fine(tc, tap);
    return f < 0 ? 0.f : x[f * C + c];
  }
};
struct DownInT {      // (x = tap*C + c, r = coarse token)
  DownIn d;
  __device__ __forceinline__ float operator()(int x, int r) const { return d(r, x); }
};
struct DownW {        // (x = n, r = tap*C + c) -> w[n][c][tap]
  const float* w; int C, K3;
  __device__ __forceinline__ float operator()(int n, int r) const {
    const int tap = r / C, c = r - tap * C;
    return w[((int64_t)n * C + c) * K3 + tap];
  }
};
struct DownWT {       // (x = tap*C + c, r = n)
  DownW q;
  __device__ __forceinline__ float operator()(int x, int r) const { return q(r, x); }
};
struct DownDataEpi {  // (i = coarse token, j = tap*C + c) -> dx_fine
  float* dx; int C; PatchGeo g;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    const int tap = j / C, c = j - tap * C;
    const int64_t f = g.fine(i, tap);
    if (f >= 0) dx[f * C + c] = v;
  }
};
struct DownWgtEpi {   // (i = n, j = tap*C + c) -> dw[n][c][tap]
  float* dw; int C, K3;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    const int tap = j / C, c = j - tap * C;
    atomicAdd(dw + ((int64_t)i * C + c) * K3 + tap, v);
  }
};

// ---------------- conv_up (k in {2,4}): x [B,Dc,Hc,Wc,C] -> y [B,D,H,W,N] (D = k*Dc ...);  w [C][N][tap]
struct UpW {          // (x = tap*N + n, r = c) -> w[c][n][tap]
  const float* w; int N, K3;
  __device__ __forceinline__ float operator()(int x, int r) const {
    const int tap = x / N, n = x - tap * N;
    return w[((int64_t)r * N + n) * K3 + tap];
  }
};
struct UpWT {         // (x = c, r = tap*N + n)
  UpW q;
  __device__ __forceinline__ float operator()(int x, int r) const { return q(r, x); }
};
struct UpFwdEpi {     // (i = coarse token, j = tap*N + n) -> y_fine[token, n] + bias[n]
  const float* bias; float* y; int N; PatchGeo g;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    const int tap = j / N, n = j - tap * N;
    const int64_t f = g.fine(i, tap);
    if (f >= 0) y[f * N + n] = v + (bias ? bias[n] : 0.f);
  }
};
struct UpDy {         // (x = coarse token, r = tap*N + n) -> dy_fine[token(tc,tap), n]
  const float* dy; int N; PatchGeo g;
  __device__ __forceinline__ float operator()(int tc, int r) const {
    const int tap = r / N, n = r - tap * N;
    const int64_t f = g.fine(tc, tap);
    return f < 0 ? 0.f : dy[f * N + n];
  }
};
struct UpDyT {        // (x = tap*N + n, r = coarse token)
  UpDy d;
  __device__ __forceinline__ float operator()(int x, int r) const { return d(r, x); }
};
struct UpWgtEpi {     // (i = c, j = tap*N + n) -> dw[c][n][tap]
  float* dw; int N, K3;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    const int tap = j / N, n = j - tap * N;
    atomicAdd(dw + ((int64_t)i * N + n) * K3 + tap, v);
  }
};

static bool mk_geo(PatchGeo& g, int B, int D, int H, int W, int k) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || k <= 0) return false;
  g = PatchGeo{B, D, H, W, (D + k - 1) / k, (H + k - 1) / k, (W + k - 1) / k, k};
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
  auto pa = make_elem<true>(EmbedVol{vol, nmod, mod, g, (int64_t)D * H * W}, (int)Tc);
  RowsT qa{w, w, K, K, 1, E, nullptr, 1, 0, (K % 4 == 0) && aligned16(w)};
  return RC(launch_gemm(pa, qa, StoreRowsEpi{bias, y, E}, Tc, E, K, 1, S_(stream)));
}

extern "C" int micf_patch_embed_bwd_weight(const float* dy, const float* vol, int nmod, int mod, float* dw, float* dbias,
                                           int B, int D, int H, int W, int E, int p, micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !vol || !dw || E <= 0 || nmod <= 0 || mod < 0 || mod >= nmod || !mk_geo(g, B, D, H, W, p)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * g.Dc * g.Hc * g.Wc;
  const int K = p * p * p;
  // dW[e, tap] = sum_t dy[t, e] * patch[t, tap]
  RowsD pa{dy, dy, E, E, 1, E, nullptr, 1, 0, (E % 4 == 0) && aligned16(dy)};
  auto qa = make_elem<false>(EmbedVolT{EmbedVol{vol, nmod, mod, g, (int64_t)D * H * W}}, K);
  if (launch_gemm(pa, qa, AtomicRowsEpi{dw, K}, E, K, (int)Tc, pick_splits(E, K, Tc), S_(stream), dbias) != hipSuccess) return MICF_ELAUNCH;
  return MICF_OK;
}

extern "C" int micf_conv_down_fwd(const float* x, const float* w, const float* bias, float* y, int B, int D, int H, int W,
                                  int C, int N, micf_stream_t stream) {
  PatchGeo g;
  if (!x || !w || !y || C <= 0 || N <= 0 || !mk_geo(g, B, D, H, W, 2)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * g.Dc * g.Hc * g.Wc;
  auto pa = make_elem<true>(DownIn{x, C, g}, (int)Tc);
  auto qa = make_elem<true>(DownW{w, C, 8}, N);
  return RC(launch_gemm(pa, qa, StoreRowsEpi{bias, y, N}, Tc, N, 8 * C, 1, S_(stream)));
}

extern "C" int micf_conv_down_bwd_data(const float* dy, const float* w, float* dx, int B, int D, int H, int W, int C, int N,
                                       micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !w || !dx || C <= 0 || N <= 0 || !mk_geo(g, B, D, H, W, 2)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * g.Dc * g.Hc * g.Wc;
  // dXcols[tc, tap*C + c] = sum_n dy[tc, n] * w[n][c][tap]  -> scattered to the fine grid (every fine token exactly once)
  RowsT pa{dy, dy, N, N, 1, (int)Tc, nullptr, 1, 0, (N % 4 == 0) && aligned16(dy)};
  auto qa = make_elem<false>(DownWT{DownW{w, C, 8}}, 8 * C);
  return RC(launch_gemm(pa, qa, DownDataEpi{dx, C, g}, Tc, 8 * C, N, 1, S_(stream)));
}

extern "C" int micf_conv_down_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int B, int D, int H, int W,
                                         int C, int N, micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !x || !dw || C <= 0 || N <= 0 || !mk_geo(g, B, D, H, W, 2)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * g.Dc * g.Hc * g.Wc;
  RowsD pa{dy, dy, N, N, 1, N, nullptr, 1, 0, (N % 4 == 0) && aligned16(dy)};
  auto qa = make_elem<false>(DownInT{DownIn{x, C, g}}, 8 * C);
  if (launch_gemm(pa, qa, DownWgtEpi{dw, C, 8}, N, 8 * C, (int)Tc, pick_splits(N, 8 * C, Tc), S_(stream), dbias) != hipSuccess)
    return MICF_ELAUNCH;
  return MICF_OK;
}

extern "C" int micf_conv_up_fwd(const float* x, const float* w, const float* bias, float* y, int B, int D, int H, int W, int C,
                                int N, int k, micf_stream_t stream) {
  // here (D, H, W) are the COARSE (input) dims; the output is exactly (kD, kH, kW)
  PatchGeo g;
  if (!x || !w || !y || C <= 0 || N <= 0 || (k != 2 && k != 4) || !mk_geo(g, B, D * k, H * k, W * k, k)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * D * H * W;
  const int K3 = k * k * k;
  RowsT pa{x, x, C, C, 1, (int)Tc, nullptr, 1, 0, (C % 4 == 0) && aligned16(x)};
  auto qa = make_elem<false>(UpW{w, N, K3}, K3 * N);
  return RC(launch_gemm(pa, qa, UpFwdEpi{bias, y, N, g}, Tc, K3 * N, C, 1, S_(stream)));
}

extern "C" int micf_conv_up_bwd_data(const float* dy, const float* w, float* dx, int B, int D, int H, int W, int C, int N, int k,
                                     micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !w || !dx || C <= 0 || N <= 0 || (k != 2 && k != 4) || !mk_geo(g, B, D * k, H * k, W * k, k)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * D * H * W;
  const int K3 = k * k * k;
  // dx[tc, c] = sum_{tap,n} dy_fine[token(tc,tap), n] * w[c][n][tap]
  auto pa = make_elem<true>(UpDy{dy, N, g}, (int)Tc);
  auto qa = make_elem<true>(UpWT{UpW{w, N, K3}}, C);
  return RC(launch_gemm(pa, qa, StoreRowsEpi{nullptr, dx, C}, Tc, C, K3 * N, 1, S_(stream)));
}

extern "C" int micf_conv_up_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int B, int D, int H, int W,
                                       int C, int N, int k, micf_stream_t stream) {
  PatchGeo g;
  if (!dy || !x || !dw || C <= 0 || N <= 0 || (k != 2 && k != 4) || !mk_geo(g, B, D * k, H * k, W * k, k)) return MICF_EINVAL;
  const int64_t Tc = (int64_t)B * D * H * W;
  const int K3 = k * k * k;
  // dw[c][n][tap] = sum_tc x[tc, c] * dy_fine[token(tc,tap), n]
  RowsD pa{x, x, C, C, 1, C, nullptr, 1, 0, (C % 4 == 0) && aligned16(x)};
  auto qa = make_elem<false>(UpDyT{UpDy{dy, N, g}}, K3 * N);
  if (launch_gemm(pa, qa, UpWgtEpi{dw, N, K3}, C, K3 * N, (int)Tc, pick_splits(C, K3 * N, Tc), S_(stream)) != hipSuccess)
    return MICF_ELAUNCH;
  if (dbias) return colsum_atomic(dy, nullptr, 1, dbias, Tc * K3, N, S_(stream));
  return MICF_OK;
}
