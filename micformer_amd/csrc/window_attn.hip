// window_attn.hip -- softmax((q*scale) k^T) v per head inside non-overlapping 3-D windows of <= 8 tokens,
// forward and backward, on channels-last token grids.  Replaces window_partition -> bmm -> softmax -> bmm ->
// window_reverse of (Cross)WindowAttention3D (MS.py:37-50, 117-132, 193-200, 251-258): the window is index math,
// the 8x8 score matrix lives in registers, and nothing is permuted or copied.
// The QK^T / PV products are 8 x hd x 8 per head-window (0.86 % of the model's FLOPs): they run on the VALU; the
// kernel is bound by the q/k/v/o token traffic, which each token row is read for once from HBM (window mates hit L1).
#include "common.h"
#include "attn_fp8.h"

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

// Thread = (window, in-window token i, head) with HEAD FASTEST, then the token: the 64-byte head slices of a token row and
// the two w-adjacent tokens of a window are contiguous in HBM, so consecutive lanes read consecutive 16-byte pieces.
// A workgroup owns wpb = 256 / (N * heads) whole windows.
struct WinThread {
  bool active; int win, i, head, local;   // local = (window-in-block * N + i) * heads + head
};
__device__ __forceinline__ WinThread win_thread(const WinGeo& g, int64_t nwin) {
  WinThread t;
  const int per = g.N * g.heads;
  const int wpb = 256 / per;
  const int wl = threadIdx.x / per, rem = threadIdx.x % per;
  t.i = rem / g.heads; t.head = rem % g.heads; t.local = threadIdx.x;
  const int64_t w = (int64_t)blockIdx.x * wpb + wl;
  t.active = (wl < wpb) && (w < nwin);
  t.win = t.active ? (int)w : 0;
  return t;
}

template <int HD>
__device__ __forceinline__ void load_row(float* dst, const float* src) {
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4 a = *reinterpret_cast<const float4*>(src + d);
    dst[d] = a.x; dst[d + 1] = a.y; dst[d + 2] = a.z; dst[d + 3] = a.w;
  }
}

// Q8 (MICF_DTYPE_BF16_ATTN_FP8 on the shapes the block kernels do not take): the operands of both products -- q * scale, k, v and
// P -- are rounded to e4m3 where the matrix-core path rounds them (attn_fp8.h); the products of two e4m3 values are exact in fp32,
// so this is the same arithmetic up to the order of the fp32 sums.
template <int HD, bool Q8 = false>   // HD > 0: compile-time head dim (multiple of 4, float4 loads); HD == 0: runtime, scalar loads
__global__ void __launch_bounds__(256) wattn_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                        const float* __restrict__ v, int ldkv, float* __restrict__ o,
                                                        int ldo, WinGeo g, float scale, int64_t nwin) {
  const WinThread t = win_thread(g, nwin);
  if (!t.active) return;
  const int hd = HD > 0 ? HD : g.hd;
  const int hoff = t.head * hd;
  int tok[kMaxWin];
#pragma unroll
  for (int j = 0; j < kMaxWin; ++j) tok[j] = (j < g.N) ? g.token(t.win, j) : 0;
  const int ti = g.token(t.win, t.i);
  float s[kMaxWin];
  float mx = -INFINITY;
  if constexpr (HD > 0) {
    float qr[HD];
    load_row<HD>(qr, q + (int64_t)ti * ldq + hoff);
#pragma unroll
    for (int d = 0; d < HD; ++d) qr[d] = Q8 ? round_fp8(qr[d] * scale) : qr[d] * scale;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) {
      s[j] = -INFINITY;
      if (j < g.N) {
        float kr[HD];
        load_row<HD>(kr, k + (int64_t)tok[j] * ldkv + hoff);
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc += qr[d] * (Q8 ? round_fp8(kr[d]) : kr[d]);
        s[j] = acc;
        mx = fmaxf(mx, acc);
      }
    }
  } else {
    const float* qp = q + (int64_t)ti * ldq + hoff;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) {
      s[j] = -INFINITY;
      if (j < g.N) {
        const float* kp = k + (int64_t)tok[j] * ldkv + hoff;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc += Q8 ? round_fp8(qp[d] * scale) * round_fp8(kp[d]) : (qp[d] * scale) * kp[d];
        s[j] = acc;
        mx = fmaxf(mx, acc);
      }
    }
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxWin; ++j) {
    s[j] = (j < g.N) ? expf(s[j] - mx) : 0.f;
    den += s[j];
  }
  const float inv = 1.0f / den;
  float* op = o + (int64_t)ti * ldo + hoff;
  if constexpr (HD > 0) {
    float acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) {
      if (j < g.N) {
        float vr[HD];
        load_row<HD>(vr, v + (int64_t)tok[j] * ldkv + hoff);
        const float p = Q8 ? round_fp8(s[j] * inv) : s[j] * inv;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] += p * (Q8 ? round_fp8(vr[d]) : vr[d]);
      }
    }
#pragma unroll
    for (int d = 0; d < HD; d += 4) *reinterpret_cast<float4*>(op + d) = make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]);
  } else {
    for (int d = 0; d < hd; ++d) {
      float acc = 0.f;
      for (int j = 0; j < g.N; ++j)
        acc += Q8 ? round_fp8(s[j] * inv) * round_fp8(v[(int64_t)tok[j] * ldkv + hoff + d]) : s[j] * inv * v[(int64_t)tok[j] * ldkv + hoff + d];
      op[d] = acc;
    }
  }
}

// Backward.  Phase 1: thread (window, i, head) recomputes row i of P, dP = do v^T, dS = P (dP - sum_j P dP) and writes
// dq_i = scale * dS k; P and dS rows go to LDS.  Phase 2: the same thread, now as key/value row j = i, forms
// dk_j = scale * dS[:, j]^T q and dv_j = P[:, j]^T do from the LDS rows of its window mates.
template <int HD>
__global__ void __launch_bounds__(256) wattn_bwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                        const float* __restrict__ v, int ldkv,
                                                        const float* __restrict__ d_o, int ldo, float* __restrict__ dq,
                                                        int lddq, float* __restrict__ dk, float* __restrict__ dv,
                                                        int lddkv, WinGeo g, float scale, int64_t nwin) {
  __shared__ float Pm[256][kMaxWin + 1];
  __shared__ float Sm[256][kMaxWin + 1];
  const WinThread t = win_thread(g, nwin);
  const int hd = HD > 0 ? HD : g.hd;
  const int hoff = t.head * hd;
  int tok[kMaxWin];
#pragma unroll
  for (int j = 0; j < kMaxWin; ++j) tok[j] = (t.active && j < g.N) ? g.token(t.win, j) : 0;
  if (t.active) {
    const int ti = g.token(t.win, t.i);
    float s[kMaxWin], dp[kMaxWin];
    float mx = -INFINITY;
    if constexpr (HD > 0) {
      float qr[HD], dor[HD];
      load_row<HD>(qr, q + (int64_t)ti * ldq + hoff);
      load_row<HD>(dor, d_o + (int64_t)ti * ldo + hoff);
#pragma unroll
      for (int d = 0; d < HD; ++d) qr[d] *= scale;
#pragma unroll
      for (int j = 0; j < kMaxWin; ++j) {
        s[j] = -INFINITY; dp[j] = 0.f;
        if (j < g.N) {
          float kr[HD], vr[HD];
          load_row<HD>(kr, k + (int64_t)tok[j] * ldkv + hoff);
          load_row<HD>(vr, v + (int64_t)tok[j] * ldkv + hoff);
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) { a += qr[d] * kr[d]; b += dor[d] * vr[d]; }
          s[j] = a; dp[j] = b;
          mx = fmaxf(mx, a);
        }
      }
    } else {
      const float* qp = q + (int64_t)ti * ldq + hoff;
      const float* dop = d_o + (int64_t)ti * ldo + hoff;
#pragma unroll
      for (int j = 0; j < kMaxWin; ++j) {
        s[j] = -INFINITY; dp[j] = 0.f;
        if (j < g.N) {
          const float* kp = k + (int64_t)tok[j] * ldkv + hoff;
          const float* vp = v + (int64_t)tok[j] * ldkv + hoff;
          float a = 0.f, b = 0.f;
          for (int d = 0; d < hd; ++d) { a += (qp[d] * scale) * kp[d]; b += dop[d] * vp[d]; }
          s[j] = a; dp[j] = b;
          mx = fmaxf(mx, a);
        }
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
      Pm[t.local][j] = s[j];
      Sm[t.local][j] = ds;
      dp[j] = ds;
    }
    float* dqp = dq + (int64_t)ti * lddq + hoff;
    if constexpr (HD > 0) {
      float acc[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxWin; ++j) {
        if (j < g.N) {
          float kr[HD];
          load_row<HD>(kr, k + (int64_t)tok[j] * ldkv + hoff);
#pragma unroll
          for (int d = 0; d < HD; ++d) acc[d] += dp[j] * kr[d];
        }
      }
#pragma unroll
      for (int d = 0; d < HD; d += 4)
        *reinterpret_cast<float4*>(dqp + d) = make_float4(acc[d] * scale, acc[d + 1] * scale, acc[d + 2] * scale, acc[d + 3] * scale);
    } else {
      for (int d = 0; d < hd; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxWin; ++j)
          if (j < g.N) acc += dp[j] * k[(int64_t)tok[j] * ldkv + hoff + d];
        dqp[d] = acc * scale;
      }
    }
  }
  __syncthreads();
  if (t.active) {
    const int j = t.i;                      // this thread now owns key/value row j of its (window, head)
    const int base = t.local - t.i * g.heads;          // LDS row of (window, i = 0, head); row of i = base + i*heads
    float* dkp = dk + (int64_t)g.token(t.win, j) * lddkv + hoff;
    float* dvp = dv + (int64_t)g.token(t.win, j) * lddkv + hoff;
    if constexpr (HD > 0) {
      float ak[HD], av[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) { ak[d] = 0.f; av[d] = 0.f; }
#pragma unroll
      for (int ii = 0; ii < kMaxWin; ++ii) {
        if (ii < g.N) {
          float qr[HD], dor[HD];
          load_row<HD>(qr, q + (int64_t)tok[ii] * ldq + hoff);
          load_row<HD>(dor, d_o + (int64_t)tok[ii] * ldo + hoff);
          const float sd = Sm[base + ii * g.heads][j], pp = Pm[base + ii * g.heads][j];
#pragma unroll
          for (int d = 0; d < HD; ++d) { ak[d] += sd * qr[d]; av[d] += pp * dor[d]; }
        }
      }
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        *reinterpret_cast<float4*>(dkp + d) = make_float4(ak[d] * scale, ak[d + 1] * scale, ak[d + 2] * scale, ak[d + 3] * scale);
        *reinterpret_cast<float4*>(dvp + d) = make_float4(av[d], av[d + 1], av[d + 2], av[d + 3]);
      }
    } else {
      for (int d = 0; d < hd; ++d) {
        float ak = 0.f, av = 0.f;
        for (int ii = 0; ii < g.N; ++ii) {
          ak += Sm[base + ii * g.heads][j] * q[(int64_t)tok[ii] * ldq + hoff + d];
          av += Pm[base + ii * g.heads][j] * d_o[(int64_t)tok[ii] * ldo + hoff + d];
        }
        dkp[d] = ak * scale;
        dvp[d] = av;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Coalesced BACKWARD (head_dim 8 / 16 / 32 with 16-byte aligned rows): LPH = head_dim / 4 LANES share one (token, head) row, each
// holding 4 of its dims.  A head row is then one contiguous 16-byte-per-lane access (the kernel above issues head_dim / 4
// separate 16-byte loads per lane, 64 bytes apart between lanes), dot products finish with log2(LPH) lane shuffles, and the
// register footprint drops ~4x (130 -> ~60 VGPRs).  Measured 68 -> 60 us at the 32^3 x 2 stage and 18 -> 12.5 us at 8^3 / 4^3;
// the same mapping made the FORWARD slower (25 -> 36 us at 32^3 x 2), so forward keeps the one-thread-per-row kernel.  Unit of work = (window, chunk of hc heads), U = N * hc * LPH threads; a workgroup holds
// floor(256 / U) units and is launched with exactly that many threads.
template <int LPH>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int m = 1; m < LPH; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
__device__ __forceinline__ float4 ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct UnitMap { int hc, nchunks, U; int64_t nunits; };

template <int HD>
__global__ void __launch_bounds__(256) wattn_bwd4_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                         const float* __restrict__ v, int ldkv, const float* __restrict__ d_o,
                                                         int ldo, float* __restrict__ dq, int lddq, float* __restrict__ dk,
                                                         float* __restrict__ dv, int lddkv, WinGeo g, float scale, UnitMap um) {
  constexpr int LPH = HD / 4;
  __shared__ float Pm[256 / LPH][kMaxWin + 1];
  __shared__ float Sm[256 / LPH][kMaxWin + 1];
  const int ul = threadIdx.x / um.U, r = threadIdx.x % um.U;
  const int64_t u = (int64_t)blockIdx.x * (blockDim.x / um.U) + ul;
  const bool active = u < um.nunits;
  const int win = active ? (int)(u / um.nchunks) : 0, chunk = active ? (int)(u % um.nchunks) : 0;
  const int ql = r % LPH, hl = (r / LPH) % um.hc, i = r / (LPH * um.hc);
  const int hoff = (chunk * um.hc + hl) * HD + 4 * ql;
  const int row = threadIdx.x / LPH;                       // LDS row of this (unit, token i, head)
  int tok[kMaxWin];
#pragma unroll
  for (int j = 0; j < kMaxWin; ++j) tok[j] = (j < g.N) ? g.token(win, j) : 0;
  const int tki = g.token(win, i);
  if (active) {
    float4 q4 = ldf4(q + (int64_t)tki * ldq + hoff);
    q4.x *= scale; q4.y *= scale; q4.z *= scale; q4.w *= scale;
    const float4 do4 = ldf4(d_o + (int64_t)tki * ldo + hoff);
    float4 k4[kMaxWin];
    float s[kMaxWin], dp[kMaxWin];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) {
      s[j] = -INFINITY; dp[j] = 0.f; k4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < g.N) {
        k4[j] = ldf4(k + (int64_t)tok[j] * ldkv + hoff);
        const float4 v4 = ldf4(v + (int64_t)tok[j] * ldkv + hoff);
        s[j] = row_sum<LPH>(dot4(q4, k4[j]));
        dp[j] = row_sum<LPH>(dot4(do4, v4));
        mx = fmaxf(mx, s[j]);
      }
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) { s[j] = (j < g.N) ? expf(s[j] - mx) : 0.f; den += s[j]; }
    const float inv = 1.0f / den;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) { s[j] *= inv; dot += s[j] * dp[j]; }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) {
      const float ds = s[j] * (dp[j] - dot);
      if (ql == 0) { Pm[row][j] = s[j]; Sm[row][j] = ds; }
      acc.x += ds * k4[j].x; acc.y += ds * k4[j].y; acc.z += ds * k4[j].z; acc.w += ds * k4[j].w;
    }
    *reinterpret_cast<float4*>(dq + (int64_t)tki * lddq + hoff) = make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
  }
  __syncthreads();
  if (active) {
    // this lane group now owns key / value row j = i of its (window, head): rows of (ii, head) = base + ii * hc
    const int base = ul * (um.U / LPH) + hl;
    float4 ak = make_float4(0.f, 0.f, 0.f, 0.f), av = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ii = 0; ii < kMaxWin; ++ii) {
      if (ii < g.N) {
        const float4 q4 = ldf4(q + (int64_t)tok[ii] * ldq + hoff);
        const float4 do4 = ldf4(d_o + (int64_t)tok[ii] * ldo + hoff);
        const float sd = Sm[base + ii * um.hc][i], pp = Pm[base + ii * um.hc][i];
        ak.x += sd * q4.x; ak.y += sd * q4.y; ak.z += sd * q4.z; ak.w += sd * q4.w;
        av.x += pp * do4.x; av.y += pp * do4.y; av.z += pp * do4.z; av.w += pp * do4.w;
      }
    }
    *reinterpret_cast<float4*>(dk + (int64_t)tki * lddkv + hoff) = make_float4(ak.x * scale, ak.y * scale, ak.z * scale, ak.w * scale);
    *reinterpret_cast<float4*>(dv + (int64_t)tki * lddkv + hoff) = av;
  }
}

// heads per unit: the largest divisor of `heads` whose unit (N tokens x hc heads x LPH lanes) fits a 256-thread workgroup
static bool unit_map(const WinGeo& g, int lph, int64_t nwin, UnitMap& um, int& threads, int& blocks) {
  um.hc = 0;
  for (int d = g.heads; d >= 1; --d)
    if (g.heads % d == 0 && g.N * d * lph <= 256) { um.hc = d; break; }
  if (!um.hc) return false;
  um.nchunks = g.heads / um.hc;
  um.U = g.N * um.hc * lph;
  um.nunits = nwin * um.nchunks;
  const int upw = 256 / um.U;
  threads = upw * um.U;
  blocks = (int)((um.nunits + upw - 1) / upw);
  return true;
}

static int make_geo(WinGeo& g, int B, int D, int H, int W, int C, int heads, int wd, int wh, int ww) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || wd <= 0 || wh <= 0 || ww <= 0) return MICF_EINVAL;
  if (C % heads != 0) return MICF_EINVAL;
  if (D % wd || H % wh || W % ww) return MICF_EINVAL;       // the host pads to window multiples first
  if (wd * wh * ww > kMaxWin || wd * wh * ww * heads > 256) return MICF_EUNSUPPORTED;
  g = WinGeo{B, D, H, W, wd, wh, ww, wd * wh * ww, heads, C / heads, D / wd, H / wh, W / ww};
  return MICF_OK;
}

}  // namespace micf
using namespace micf;

static int window_attn_fwd_impl(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo, int B, int D, int H,
                                int W, int C, int heads, int wd, int wh, int ww, float scale, bool q8, micf_stream_t stream);

extern "C" int micf_window_attn_fwd(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo,
                                    int B, int D, int H, int W, int C, int heads, int wd, int wh, int ww, float scale,
                                    micf_stream_t stream) {
  return window_attn_fwd_impl(q, ldq, k, v, ldkv, o, ldo, B, D, H, W, C, heads, wd, wh, ww, scale, false, stream);
}

extern "C" int micf_window_attn_fwd_fp8(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo,
                                        int B, int D, int H, int W, int C, int heads, int wd, int wh, int ww, float scale,
                                        micf_stream_t stream) {
  return window_attn_fwd_impl(q, ldq, k, v, ldkv, o, ldo, B, D, H, W, C, heads, wd, wh, ww, scale, true, stream);
}

static int window_attn_fwd_impl(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo, int B, int D, int H,
                                int W, int C, int heads, int wd, int wh, int ww, float scale, bool q8, micf_stream_t stream) {
  if (!q || !k || !v || !o) return MICF_EINVAL;
  WinGeo g;
  int rc = make_geo(g, B, D, H, W, C, heads, wd, wh, ww);
  if (rc) return rc;
  const int64_t nwin = (int64_t)B * g.nwd * g.nwh * g.nww;
  const dim3 grid(ceil_div(nwin, 256 / (g.N * heads)));
  const bool vec = (g.hd % 4 == 0) && (ldq % 4 == 0) && (ldkv % 4 == 0) && (ldo % 4 == 0) && aligned16(q) &&
                   aligned16(k) && aligned16(v) && aligned16(o);
  hipStream_t s = (hipStream_t)stream;
#define MICF_WF(HD_) do { if (q8) hipLaunchKernelGGL((wattn_fwd_kernel<HD_, true>), grid, dim3(256), 0, s, q, ldq, k, v, ldkv, o, ldo, g, scale, nwin); \
                         else hipLaunchKernelGGL((wattn_fwd_kernel<HD_, false>), grid, dim3(256), 0, s, q, ldq, k, v, ldkv, o, ldo, g, scale, nwin); } while (0)
  if (vec && g.hd == 16) MICF_WF(16);
  else if (vec && g.hd == 8) MICF_WF(8);
  else if (vec && g.hd == 32) MICF_WF(32);
  else MICF_WF(0);
#undef MICF_WF
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_window_attn_bwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* d_o,
                                    int ldo, float* dq, int lddq, float* dk, float* dv, int lddkv, int B, int D, int H,
                                    int W, int C, int heads, int wd, int wh, int ww, float scale, micf_stream_t stream) {
  if (!q || !k || !v || !d_o || !dq || !dk || !dv) return MICF_EINVAL;
  WinGeo g;
  int rc = make_geo(g, B, D, H, W, C, heads, wd, wh, ww);
  if (rc) return rc;
  const int64_t nwin = (int64_t)B * g.nwd * g.nwh * g.nww;
  const dim3 grid(ceil_div(nwin, 256 / (g.N * heads)));
  const bool vec = (g.hd % 4 == 0) && (ldq % 4 == 0) && (ldkv % 4 == 0) && (ldo % 4 == 0) && (lddq % 4 == 0) &&
                   (lddkv % 4 == 0) && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(d_o) && aligned16(dq) &&
                   aligned16(dk) && aligned16(dv);
  hipStream_t s = (hipStream_t)stream;
  UnitMap um; int threads = 0, blocks = 0;
  if (vec && (g.hd == 8 || g.hd == 16 || g.hd == 32) && unit_map(g, g.hd / 4, nwin, um, threads, blocks)) {
#define MICF_BWD4(HD_) hipLaunchKernelGGL(wattn_bwd4_kernel<HD_>, dim3(blocks), dim3(threads), 0, s, q, ldq, k, v, ldkv, d_o, ldo, dq, lddq, dk, dv, lddkv, g, scale, um)
    if (g.hd == 16) MICF_BWD4(16); else if (g.hd == 8) MICF_BWD4(8); else MICF_BWD4(32);
#undef MICF_BWD4
    MICF_RETURN_LAUNCH();
  }
#define MICF_BWD(HD_) hipLaunchKernelGGL(wattn_bwd_kernel<HD_>, grid, dim3(256), 0, s, q, ldq, k, v, ldkv, d_o, ldo, dq, lddq, dk, dv, lddkv, g, scale, nwin)
  if (vec && g.hd == 16) MICF_BWD(16);
  else if (vec && g.hd == 8) MICF_BWD(8);
  else if (vec && g.hd == 32) MICF_BWD(32);
  else MICF_BWD(0);
#undef MICF_BWD
  MICF_RETURN_LAUNCH();
}
