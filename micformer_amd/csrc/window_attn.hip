// window_attn.hip -- softmax((q*scale) k^T) v per head inside non-overlapping 3-D windows of <= 8 tokens,
// forward and backward, on channels-last token grids.  Replaces window_partition -> bmm -> softmax -> bmm ->
// window_reverse of (Cross)WindowAttention3D (MS.py:37-50, 117-132, 193-200, 251-258): the window is index math,
// the 8x8 score matrix lives in registers, and nothing is permuted or copied.
// The QK^T / PV products are 8 x hd x 8 per head-window (0.86 % of the model's FLOPs): they run on the VALU; the
// kernel is bound by the q/k/v/o token traffic, which each token row is read for once from HBM (window mates hit L1).
#include "common.h"

namespace micf {

struct WinGeo {
  int B, D, H, W, wd, wh, ww, N, heads, hd;
  int nwd, nwh, nww;
  // token index of window `win`, in-window index i (order (wd, wh, ww) row-major, MS.py:47-49)
  __device__ __forceinline__ int token(int win, int i) const {
    int t = win;
    const int xw = t % nww; t /= nww;
    const int xh = t % nwh; t /= nwh;
    const int xd = t % nwd; const int b = t / nwd;
    const int iw = i % ww; const int ih = (i / ww) % wh; const int id = i / (ww * wh);
    return ((b * D + xd * wd + id) * H + xh * wh + ih) * W + xw * ww + iw;
  }
};

constexpr int kMaxWin = 8;

// thread = (window, head, query i); pair-major inside the block so a pair's N threads are adjacent.
template <int HD>   // HD > 0: compile-time head dim (multiple of 4, float4 loads); HD == 0: runtime, scalar loads
__global__ void __launch_bounds__(256) wattn_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                        const float* __restrict__ v, int ldkv, float* __restrict__ o,
                                                        int ldo, WinGeo g, float scale, int64_t npairs) {
  const int ppb = 256 / g.N;
  const int p_local = threadIdx.x / g.N, i = threadIdx.x % g.N;
  if (p_local >= ppb) return;
  const int64_t pair = (int64_t)blockIdx.x * ppb + p_local;
  if (pair >= npairs) return;
  const int head = (int)(pair % g.heads);
  const int win = (int)(pair / g.heads);
  const int hd = HD > 0 ? HD : g.hd;
  const int hoff = head * hd;
  const float* qp = q + (int64_t)g.token(win, i) * ldq + hoff;
  float s[kMaxWin];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < kMaxWin; ++j) {
    s[j] = -INFINITY;
    if (j < g.N) {
      const float* kp = k + (int64_t)g.token(win, j) * ldkv + hoff;
      float acc = 0.f;
      if constexpr (HD > 0) {
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          const float4 a = *reinterpret_cast<const float4*>(qp + d);
          const float4 b = *reinterpret_cast<const float4*>(kp + d);
          acc += (a.x * scale) * b.x + (a.y * scale) * b.y + (a.z * scale) * b.z + (a.w * scale) * b.w;
        }
      } else {
        for (int d = 0; d < hd; ++d) acc += (qp[d] * scale) * kp[d];
      }
      s[j] = acc;
      mx = fmaxf(mx, acc);
    }
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxWin; ++j) {
    s[j] = (j < g.N) ? expf(s[j] - mx) : 0.f;
    den += s[j];
  }
  const float inv = 1.0f / den;
  float* op = o + (int64_t)g.token(win, i) * ldo + hoff;
  if constexpr (HD > 0) {
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < kMaxWin; ++j) {
        if (j < g.N) {
          const float4 b = *reinterpret_cast<const float4*>(v + (int64_t)g.token(win, j) * ldkv + hoff + d);
          const float p = s[j] * inv;
          acc.x += p * b.x; acc.y += p * b.y; acc.z += p * b.z; acc.w += p * b.w;
        }
      }
      *reinterpret_cast<float4*>(op + d) = acc;
    }
  } else {
    for (int d = 0; d < hd; ++d) {
      float acc = 0.f;
      for (int j = 0; j < g.N; ++j) acc += s[j] * inv * v[(int64_t)g.token(win, j) * ldkv + hoff + d];
      op[d] = acc;
    }
  }
}

// Backward.  Phase 1: thread (pair, i) recomputes row i of P, dP = do v^T, dS = P (dP - sum_j P dP) and writes
// dq_i = scale * dS k; P and dS rows go to LDS.  Phase 2: thread (pair, j) forms dk_j = scale * dS[:, j]^T q and
// dv_j = P[:, j]^T do.
template <int HD>
__global__ void __launch_bounds__(256) wattn_bwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                        const float* __restrict__ v, int ldkv,
                                                        const float* __restrict__ d_o, int ldo, float* __restrict__ dq,
                                                        int lddq, float* __restrict__ dk, float* __restrict__ dv,
                                                        int lddkv, WinGeo g, float scale, int64_t npairs) {
  __shared__ float Pm[256][kMaxWin + 1];
  __shared__ float Sm[256][kMaxWin + 1];
  const int ppb = 256 / g.N;
  const int p_local = threadIdx.x / g.N, i = threadIdx.x % g.N;
  const int64_t pair = (int64_t)blockIdx.x * ppb + p_local;
  const bool active = (p_local < ppb) && (pair < npairs);
  const int head = active ? (int)(pair % g.heads) : 0;
  const int win = active ? (int)(pair / g.heads) : 0;
  const int hd = HD > 0 ? HD : g.hd;
  const int hoff = head * hd;
  if (active) {
    const int ti = g.token(win, i);
    const float* qp = q + (int64_t)ti * ldq + hoff;
    const float* dop = d_o + (int64_t)ti * ldo + hoff;
    float s[kMaxWin], dp[kMaxWin];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) {
      s[j] = -INFINITY; dp[j] = 0.f;
      if (j < g.N) {
        const int tj = g.token(win, j);
        const float* kp = k + (int64_t)tj * ldkv + hoff;
        const float* vp = v + (int64_t)tj * ldkv + hoff;
        float a = 0.f, b = 0.f;
        for (int d = 0; d < hd; ++d) { a += (qp[d] * scale) * kp[d]; b += dop[d] * vp[d]; }
        s[j] = a; dp[j] = b;
        mx = fmaxf(mx, a);
      }
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) { s[j] = (j < g.N) ? expf(s[j] - mx) : 0.f; den += s[j]; }
    const float inv = 1.0f / den;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) { s[j] *= inv; dot += s[j] * dp[j]; }
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) {
      const float ds = s[j] * (dp[j] - dot);
      Pm[threadIdx.x][j] = s[j];
      Sm[threadIdx.x][j] = ds;
      dp[j] = ds;
    }
    float* dqp = dq + (int64_t)ti * lddq + hoff;
    for (int d = 0; d < hd; ++d) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxWin; ++j)
        if (j < g.N) acc += dp[j] * k[(int64_t)g.token(win, j) * ldkv + hoff + d];
      dqp[d] = acc * scale;
    }
  }
  __syncthreads();
  if (active) {
    const int j = i;                       // this thread now owns key/value row j of its pair
    const int tj = g.token(win, j);
    const int base = p_local * g.N;
    float* dkp = dk + (int64_t)tj * lddkv + hoff;
    float* dvp = dv + (int64_t)tj * lddkv + hoff;
    for (int d = 0; d < hd; ++d) {
      float ak = 0.f, av = 0.f;
      for (int ii = 0; ii < g.N; ++ii) {
        const int t2 = g.token(win, ii);
        ak += Sm[base + ii][j] * q[(int64_t)t2 * ldq + hoff + d];
        av += Pm[base + ii][j] * d_o[(int64_t)t2 * ldo + hoff + d];
      }
      dkp[d] = ak * scale;
      dvp[d] = av;
    }
  }
}

static int make_geo(WinGeo& g, int B, int D, int H, int W, int C, int heads, int wd, int wh, int ww) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || wd <= 0 || wh <= 0 || ww <= 0) return MICF_EINVAL;
  if (C % heads != 0) return MICF_EINVAL;
  if (D % wd || H % wh || W % ww) return MICF_EINVAL;       // the host pads to window multiples first
  if (wd * wh * ww > kMaxWin) return MICF_EUNSUPPORTED;
  g = WinGeo{B, D, H, W, wd, wh, ww, wd * wh * ww, heads, C / heads, D / wd, H / wh, W / ww};
  return MICF_OK;
}

}  // namespace micf
using namespace micf;

extern "C" int micf_window_attn_fwd(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo,
                                    int B, int D, int H, int W, int C, int heads, int wd, int wh, int ww, float scale,
                                    micf_stream_t stream) {
  if (!q || !k || !v || !o) return MICF_EINVAL;
  WinGeo g;
  int rc = make_geo(g, B, D, H, W, C, heads, wd, wh, ww);
  if (rc) return rc;
  const int64_t npairs = (int64_t)B * g.nwd * g.nwh * g.nww * heads;
  const int ppb = 256 / g.N;
  const dim3 grid(ceil_div(npairs, ppb));
  const bool vec = (g.hd % 4 == 0) && (ldq % 4 == 0) && (ldkv % 4 == 0) && (ldo % 4 == 0) && aligned16(q) &&
                   aligned16(k) && aligned16(v) && aligned16(o);
  hipStream_t s = (hipStream_t)stream;
  if (vec && g.hd == 16) hipLaunchKernelGGL(wattn_fwd_kernel<16>, grid, dim3(256), 0, s, q, ldq, k, v, ldkv, o, ldo, g, scale, npairs);
  else if (vec && g.hd == 8) hipLaunchKernelGGL(wattn_fwd_kernel<8>, grid, dim3(256), 0, s, q, ldq, k, v, ldkv, o, ldo, g, scale, npairs);
  else if (vec && g.hd == 32) hipLaunchKernelGGL(wattn_fwd_kernel<32>, grid, dim3(256), 0, s, q, ldq, k, v, ldkv, o, ldo, g, scale, npairs);
  else hipLaunchKernelGGL(wattn_fwd_kernel<0>, grid, dim3(256), 0, s, q, ldq, k, v, ldkv, o, ldo, g, scale, npairs);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_window_attn_bwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* d_o,
                                    int ldo, float* dq, int lddq, float* dk, float* dv, int lddkv, int B, int D, int H,
                                    int W, int C, int heads, int wd, int wh, int ww, float scale, micf_stream_t stream) {
  if (!q || !k || !v || !d_o || !dq || !dk || !dv) return MICF_EINVAL;
  WinGeo g;
  int rc = make_geo(g, B, D, H, W, C, heads, wd, wh, ww);
  if (rc) return rc;
  const int64_t npairs = (int64_t)B * g.nwd * g.nwh * g.nww * heads;
  const int ppb = 256 / g.N;
  const dim3 grid(ceil_div(npairs, ppb));
  hipLaunchKernelGGL(wattn_bwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, q, ldq, k, v, ldkv, d_o, ldo, dq, lddq,
                     dk, dv, lddkv, g, scale, npairs);
  MICF_RETURN_LAUNCH();
}
