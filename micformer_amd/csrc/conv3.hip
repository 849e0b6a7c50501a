// conv3.hip -- 3x3x3 / stride 1 / zero-pad 1 convolution, forward + both gradients, as implicit GEMM on the fp32
// MFMA core: the halo gather, the concatenation of the two input modalities and the NCDHW <-> channels-last
// conversion of the logits are accessor index math (no im2col, no torch.cat, no permute copies).
// Replaces conv_offset[0] on cat[LN(x), xa] (MS.py:314, 354-356) and Head.out_conv (MS.py:1046, 1053).
#include "common.h"

namespace micf {

// P operand of the forward: (x = token, r = tap*Cin + c) -> input[nbr(token, tap), c], zero outside the volume.
// sign = +1: nbr = token + (tap - 1)   (forward / weight gradient)
struct Conv3In {
  const float* x1; const float* x2; int c1, c2, Cin; Geo g; int X; int vec;
  __device__ __forceinline__ bool nbr(int t, int tap, int& tn) const {
    int b, d, h, w; g.decode(t, b, d, h, w);
    const int dd = d + tap / 9 - 1, hh = h + (tap / 3) % 3 - 1, ww = w + tap % 3 - 1;
    if ((unsigned)dd >= (unsigned)g.D || (unsigned)hh >= (unsigned)g.H || (unsigned)ww >= (unsigned)g.W) return false;
    tn = g.token(b, dd, hh, ww);
    return true;
  }
  __device__ __forceinline__ float at(int tn, int c) const {
    return c < c1 ? x1[(int64_t)tn * c1 + c] : x2[(int64_t)tn * c2 + (c - c1)];
  }
  // 4 consecutive channels of one (token, tap); zero-filled when outside
  __device__ __forceinline__ void load4(int t, int r, int r_lim, float* v) const {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (vec) {
      if (r >= r_lim) return;
      const int tap = r / Cin, c = r - tap * Cin;
      int tn;
      if (!nbr(t, tap, tn)) return;
      const float4 q = c < c1 ? *reinterpret_cast<const float4*>(x1 + (int64_t)tn * c1 + c)
                              : *reinterpret_cast<const float4*>(x2 + (int64_t)tn * c2 + (c - c1));
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rr = r + e;
        if (rr >= r_lim) continue;
        const int tap = rr / Cin, c = rr - tap * Cin;
        int tn;
        if (nbr(t, tap, tn)) v[e] = at(tn, c);
      }
    }
  }
};

struct Conv3FwdP {   // T mapping: x = token, r = tap*Cin + c
  Conv3In in;
  template <int BX> using Stage = StageT<BX>;
  template <int BX>
  __device__ __forceinline__ void fetch(Stage<BX>& st, int x0, int r0, int r_end, int tid) const {
    st.fetch(x0, r0, tid, [&](int x, int r, float* v) {
      if (x < in.X) in.load4(x, r, r_end, v); else v[0] = v[1] = v[2] = v[3] = 0.f;
    });
  }
};

struct Conv3WgtQ {   // D mapping: x = tap*Cin + c (4 consecutive c), r = token
  Conv3In in; int J;
  template <int BX> using Stage = StageD<BX>;
  template <int BX>
  __device__ __forceinline__ void fetch(Stage<BX>& st, int x0, int r0, int r_end, int tid) const {
    st.fetch(x0, r0, tid, [&](int x, int r, float* v) {
      if (r < r_end) in.load4(r, x, J, v); else v[0] = v[1] = v[2] = v[3] = 0.f;
    });
  }
};

// weights w[n][c][tap] seen as (x = n, r = tap*Cin + c)  [forward]   or   (x = c, r = tap*N + n)  [data gradient]
struct Conv3WFwd {
  const float* w; int Cin;
  __device__ __forceinline__ float operator()(int x, int r) const {
    const int tap = r / Cin, c = r - tap * Cin;
    return w[((int64_t)x * Cin + c) * 27 + tap];
  }
};
struct Conv3WBwd {
  const float* w; int Cin, N;
  __device__ __forceinline__ float operator()(int x, int r) const {
    const int tap = r / N, n = r - tap * N;
    return w[((int64_t)n * Cin + x) * 27 + tap];
  }
};

// dy seen from the INPUT token: (x = token, r = tap*N + n) -> dy[token - (tap - 1), n]
struct Conv3DyGather {
  const float* dy; int layout, N; Geo g; int64_t DHW;
  __device__ __forceinline__ float operator()(int x, int r) const {
    const int tap = r / N, n = r - tap * N;
    int b, d, h, w; g.decode(x, b, d, h, w);
    const int dd = d - (tap / 9 - 1), hh = h - ((tap / 3) % 3 - 1), ww = w - (tap % 3 - 1);
    if ((unsigned)dd >= (unsigned)g.D || (unsigned)hh >= (unsigned)g.H || (unsigned)ww >= (unsigned)g.W) return 0.f;
    const int64_t vox = ((int64_t)dd * g.H + hh) * g.W + ww;
    return layout == 0 ? dy[((int64_t)b * DHW + vox) * N + n] : dy[((int64_t)b * N + n) * DHW + vox];
  }
};
// dy as (x = n, r = token)
struct Conv3DyT {
  const float* dy; int layout, N; int64_t DHW;
  __device__ __forceinline__ float operator()(int x, int r) const {
    if (layout == 0) return dy[(int64_t)r * N + x];
    const int64_t b = r / DHW, vox = r - b * DHW;
    return dy[(b * N + x) * DHW + vox];
  }
};

struct Conv3FwdEpi {
  const float* bias; float* y; int layout, N; int64_t DHW;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    if (bias) v += bias[j];
    if (layout == 0) y[(int64_t)i * N + j] = v;
    else { const int64_t b = i / DHW, vox = i - b * DHW; y[(b * N + j) * DHW + vox] = v; }
  }
};
struct Conv3FwdSplitEpi {   // y pre-zeroed; split 0 adds the bias; channels-last only
  const float* bias; float* y; int N;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    if (bias && blockIdx.z == 0) v += bias[j];
    atomicAdd(y + (int64_t)i * N + j, v);
  }
};
struct Conv3DataEpi {
  float* d1; float* d2; int c1, c2, acc1, acc2;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    if (j < c1) { if (d1) { float* p = d1 + (int64_t)i * c1 + j; *p = acc1 ? *p + v : v; } }
    else if (d2) { float* p = d2 + (int64_t)i * c2 + (j - c1); *p = acc2 ? *p + v : v; }
  }
};
struct Conv3WgtEpi {   // (i = tap*Cin + c, j = n) -> dw[n][c][tap]
  float* dw; int Cin;
  __device__ __forceinline__ void operator()(int i, int j, float v) const {
    const int tap = i / Cin, c = i - tap * Cin;
    atomicAdd(dw + ((int64_t)j * Cin + c) * 27 + tap, v);
  }
};

// per-channel sums of an NCDHW tensor [B, N, V] accumulated into out[N]
__global__ void __launch_bounds__(256) plane_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int N,
                                                        int64_t V, int chunks) {
  const int plane = blockIdx.y;                       // b*N + n
  const int64_t per = (V + chunks - 1) / chunks;
  const int64_t v0 = blockIdx.x * per, v1 = (v0 + per < V) ? v0 + per : V;
  float s = 0.f;
  for (int64_t i = v0 + threadIdx.x; i < v1; i += 256) s += x[(int64_t)plane * V + i];
  s = wave_sum(s);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + plane % N, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace micf
using namespace micf;

static bool conv3_args_ok(int B, int D, int H, int W, int N, int c1, int c2) {
  return B > 0 && D > 0 && H > 0 && W > 0 && N > 0 && c1 > 0 && c2 >= 0 && (int64_t)B * D * H * W < (1LL << 31) &&
         (int64_t)27 * (c1 + c2) * (int64_t)(N > c1 + c2 ? N : c1 + c2) < (1LL << 31);
}

extern "C" int micf_conv3_fwd(const float* x1, int c1, const float* x2, int c2, const float* w, const float* bias,
                              float* y, int y_layout, int B, int D, int H, int W, int N, micf_stream_t stream) {
  if (!x1 || !w || !y || (c2 > 0 && !x2) || !conv3_args_ok(B, D, H, W, N, c1, c2)) return MICF_EINVAL;
  const Geo g{B, D, H, W};
  const int Cin = c1 + c2;
  const int64_t T = g.tokens();
  const int vec = (c1 % 4 == 0) && (c2 % 4 == 0) && aligned16(x1) && (!x2 || aligned16(x2));
  Conv3FwdP pa{Conv3In{x1, x2 ? x2 : x1, c1, c2 > 0 ? c2 : 1, Cin, g, (int)T, vec}};
  auto qa = make_elem<false>(Conv3WFwd{w, Cin}, N);
  hipStream_t s = (hipStream_t)stream;
  const int splits = (y_layout == 0) ? pick_splits(T, N, 27 * Cin) : 1;
  if (splits > 1) {   // small token grids (8^3, 4^3 stages): split the 27*Cin reduction over workgroups
    if (hipMemsetAsync(y, 0, sizeof(float) * (size_t)T * N, s) != hipSuccess) return MICF_ELAUNCH;
    return launch_gemm(pa, qa, Conv3FwdSplitEpi{bias, y, N}, T, N, 27 * Cin, splits, s) == hipSuccess ? MICF_OK : MICF_ELAUNCH;
  }
  Conv3FwdEpi epi{bias, y, y_layout, N, (int64_t)D * H * W};
  return launch_gemm(pa, qa, epi, T, N, 27 * Cin, 1, s) == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

extern "C" int micf_conv3_bwd_data(const float* dy, int dy_layout, const float* w, float* dx1, int c1, int acc1,
                                   float* dx2, int c2, int acc2, int B, int D, int H, int W, int N,
                                   micf_stream_t stream) {
  if (!dy || !w || (!dx1 && !dx2) || !conv3_args_ok(B, D, H, W, N, c1, c2)) return MICF_EINVAL;
  const Geo g{B, D, H, W};
  const int Cin = c1 + c2;
  const int64_t T = g.tokens();
  auto pa = make_elem<true>(Conv3DyGather{dy, dy_layout, N, g, (int64_t)D * H * W}, (int)T);
  auto qa = make_elem<false>(Conv3WBwd{w, Cin, N}, Cin);
  Conv3DataEpi epi{dx1, dx2, c1, c2 > 0 ? c2 : 1, acc1, acc2};
  return launch_gemm(pa, qa, epi, T, Cin, 27 * N, 1, (hipStream_t)stream) == hipSuccess ? MICF_OK : MICF_ELAUNCH;
}

extern "C" int micf_conv3_bwd_weight(const float* dy, int dy_layout, const float* x1, int c1, const float* x2, int c2,
                                     float* dw, float* dbias, int B, int D, int H, int W, int N, micf_stream_t stream) {
  if (!dy || !x1 || !dw || (c2 > 0 && !x2) || !conv3_args_ok(B, D, H, W, N, c1, c2)) return MICF_EINVAL;
  const Geo g{B, D, H, W};
  const int Cin = c1 + c2;
  const int64_t T = g.tokens();
  const int64_t DHW = (int64_t)D * H * W;
  const int vec = (c1 % 4 == 0) && (c2 % 4 == 0) && aligned16(x1) && (!x2 || aligned16(x2));
  // dW^T[tap*Cin + c, n] = sum_t in[nbr(t, tap), c] * dy[t, n]: the long 27*Cin axis is the tile's I side (full MFMA rows),
  // the narrow N (16 / 8) is its J side; the reduction over tokens is split across workgroups (atomic epilogue).
  Conv3WgtQ pa{Conv3In{x1, x2 ? x2 : x1, c1, c2 > 0 ? c2 : 1, Cin, g, (int)T, vec}, 27 * Cin};
  Conv3WgtEpi epi{dw, Cin};
  hipStream_t s = (hipStream_t)stream;
  const int splits = pick_splits(27 * Cin, N, T);
  hipError_t e;
  if (dy_layout == 0) {
    RowsD qa{dy, dy, N, N, 1, N, nullptr, 1, 0, (N % 4 == 0) && aligned16(dy)};
    e = launch_gemm(pa, qa, epi, 27 * Cin, N, (int)T, splits, s);
  } else {
    auto qa = make_elem<true>(Conv3DyT{dy, dy_layout, N, DHW}, N);
    e = launch_gemm(pa, qa, epi, 27 * Cin, N, (int)T, splits, s);
  }
  if (e != hipSuccess) return MICF_ELAUNCH;
  if (dbias) {
    if (dy_layout == 0) return colsum_atomic(dy, nullptr, 1, dbias, T, N, s);
    int chunks = (int)((DHW + 65535) / 65536);
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(plane_sum_kernel, dim3(chunks, B * N), dim3(256), 0, s, dy, dbias, N, DHW, chunks);
    MICF_RETURN_LAUNCH();
  }
  return MICF_OK;
}
