// conv3.hip -- 3x3x3 / stride 1 / zero-pad 1 convolution, forward + both gradients, as implicit GEMM on the fp32
// MFMA core: the halo gather, the concatenation of the two input modalities and the NCDHW <-> channels-last
// conversion of the logits are accessor index math (no im2col, no torch.cat, no permute copies).
// Replaces conv_offset[0] on cat[LN(x), xa] (MS.py:314, 354-356) and Head.out_conv (MS.py:1046, 1053).
// Orientation (gemm_core.h): the tile's I side is whatever is contiguous in the OUTPUT -- the 16 hidden channels for the
// channels-last offset conv, the voxels for the NCDHW logits, the input channels for d(input).
#include "common.h"
#include "conv3_layout.h"
#include "gemm_dma.h"

namespace micf {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }

// input [x1 (c1) | x2 (c2)] at the 27 neighbours of a token; zero outside the volume
struct Conv3In {
  const float* x1; const float* x2; int c1, c2, Cin; Geo g; int X; int vec; FastDiv fCin;
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
  // 4 consecutive channels of one (token, tap) at reduction index r = tap*Cin + c; zero-filled when outside
  __device__ __forceinline__ void load4(int t, int r, int r_lim, float* v) const {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (vec) {
      if (r >= r_lim) return;
      uint32_t tap, c; fCin.divmod((uint32_t)r, tap, c);
      int tn;
      if (!nbr(t, (int)tap, tn)) return;
      const float4 q = (int)c < c1 ? ld4(x1 + (int64_t)tn * c1 + c) : ld4(x2 + (int64_t)tn * c2 + (c - c1));
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rr = r + e;
        if (rr >= r_lim) continue;
        uint32_t tap, c; fCin.divmod((uint32_t)rr, tap, c);
        int tn;
        if (nbr(t, (int)tap, tn)) v[e] = at(tn, (int)c);
      }
    }
  }
};

struct Conv3TokT {   // T mapping: x = token, r = tap*Cin + c
  Conv3In in;
  template <int BX, int BR> using Stage = StageT<BX, BR>;
  template <int BX, int BR>
  __device__ __forceinline__ void fetch(Stage<BX, BR>& st, int x0, int r0, int r_end, int tid) const {
    st.fetch(x0, r0, tid, [&](int x, int r, float* v) {
      if (x < in.X) in.load4(x, r, r_end, v); else v[0] = v[1] = v[2] = v[3] = 0.f;
    });
  }
};

struct Conv3ColD {   // D mapping: x = tap*Cin + c (4 consecutive c), r = token
  Conv3In in; int J;
  template <int BX, int BR> using Stage = StageD<BX, BR>;
  template <int BX, int BR>
  __device__ __forceinline__ void fetch(Stage<BX, BR>& st, int x0, int r0, int r_end, int tid) const {
    st.fetch(x0, r0, tid, [&](int x, int r, float* v) {
      if (r < r_end) in.load4(r, x, J, v); else v[0] = v[1] = v[2] = v[3] = 0.f;
    });
  }
};

// weights w[n][c][tap] seen as (x = n, r = tap*Cin + c)  [forward]   or   (x = c, r = tap*N + n)  [data gradient]
struct Conv3WFwd {
  const float* w; int Cin; FastDiv fCin;
  __device__ __forceinline__ float operator()(int x, int r) const {
    uint32_t tap, c; fCin.divmod((uint32_t)r, tap, c);
    return w[((int64_t)x * Cin + c) * 27 + tap];
  }
};
struct Conv3WBwd {
  const float* w; int Cin; FastDiv fN;
  __device__ __forceinline__ float operator()(int x, int r) const {
    uint32_t tap, n; fN.divmod((uint32_t)r, tap, n);
    return w[((int64_t)n * Cin + x) * 27 + tap];
  }
};

// dy seen from the INPUT token: (x = token, r = tap*N + n) -> dy[token - (tap - 1), n]
struct Conv3DyGather {
  const float* dy; int layout, N; Geo g; int64_t DHW; FastDiv fN;
  __device__ __forceinline__ float operator()(int x, int r) const {
    uint32_t tap, n; fN.divmod((uint32_t)r, tap, n);
    int b, d, h, w; g.decode(x, b, d, h, w);
    const int dd = d - ((int)tap / 9 - 1), hh = h - (((int)tap / 3) % 3 - 1), ww = w - ((int)tap % 3 - 1);
    if ((unsigned)dd >= (unsigned)g.D || (unsigned)hh >= (unsigned)g.H || (unsigned)ww >= (unsigned)g.W) return 0.f;
    const int64_t vox = ((int64_t)dd * g.H + hh) * g.W + ww;
    return layout == 0 ? dy[((int64_t)b * DHW + vox) * N + n] : dy[((int64_t)b * N + n) * DHW + vox];
  }
};
// NCDHW dy as (x = n, r = token)
struct Conv3DyPlanes {
  const float* dy; int N; int64_t DHW; FastDiv fDHW;
  __device__ __forceinline__ float operator()(int x, int r) const {
    uint32_t b, vox; fDHW.divmod((uint32_t)r, b, vox);
    return dy[((int64_t)b * N + x) * DHW + vox];
  }
};

struct Conv3FwdClEpi {      // channels-last y[token j, n = i .. i+3]
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
struct Conv3FwdClSplitEpi {  // y pre-zeroed; reduction split 0 adds the bias
  const float* bias; float* y; int N;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    float* p = y + (int64_t)j * N + i;
    MICF_FOR_N(n, e) atomicAdd(p + e, v[e] + ((bias && blockIdx.y == 0) ? bias[i + e] : 0.f));
  }
};
struct Conv3FwdPlanesEpi {   // NCDHW y[b, n = j, voxel i .. i+3]
  const float* bias; float* y; int N; int64_t DHW; FastDiv fDHW; int vec;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    const float bb = bias ? bias[j] : 0.f;
    uint32_t b, vox; fDHW.divmod((uint32_t)i, b, vox);
    if (vec && n == 4) {           // DHW % 4 == 0: the 4 voxels stay inside one sample
      st4(y + ((int64_t)b * N + j) * DHW + vox, v[0] + bb, v[1] + bb, v[2] + bb, v[3] + bb);
    } else {
      MICF_FOR_N(n, e) { fDHW.divmod((uint32_t)(i + e), b, vox); y[((int64_t)b * N + j) * DHW + vox] = v[e] + bb; }
    }
  }
};
struct Conv3DataEpi {        // dx[token j, c = i .. i+3] split over the two sources at c1
  float* d1; float* d2; int c1, c2, acc1, acc2, vec;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    if (vec && n == 4) {
      float* p; int acc;
      if (i < c1) { if (!d1) return; p = d1 + (int64_t)j * c1 + i; acc = acc1; }
      else { if (!d2) return; p = d2 + (int64_t)j * c2 + (i - c1); acc = acc2; }
      if (acc) { const float4 o = ld4(p); v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
      st4(p, v[0], v[1], v[2], v[3]);
    } else {
      MICF_FOR_N(n, e) {
        const int c = i + e;
        if (c < c1) { if (d1) { float* p = d1 + (int64_t)j * c1 + c; *p = acc1 ? *p + v[e] : v[e]; } }
        else if (d2) { float* p = d2 + (int64_t)j * c2 + (c - c1); *p = acc2 ? *p + v[e] : v[e]; }
      }
    }
  }
};
struct Conv3WgtEpi {   // (i = tap*Cin + c, j = n) -> dw[n][c][tap]
  float* dw; int Cin; FastDiv fCin;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v, int n) const {
    MICF_FOR_N(n, e) {
      uint32_t tap, c; fCin.divmod((uint32_t)(i + e), tap, c);
      atomicAdd(dw + ((int64_t)j * Cin + c) * 27 + tap, v[e]);
    }
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

int conv3_fwd_direct(const float* x1, int c1, const float* x2, int c2, const float* w, const float* bias, float* y,
                     int y_layout, int B, int D, int H, int W, int N, hipStream_t stream);                 // conv3_direct.hip
int conv3_wgrad_direct(const float* dy, int dy_layout, const float* x1, int c1, const float* x2, int c2, float* dw,
                       float* dbias, int B, int D, int H, int W, int N, hipStream_t stream);   // conv3_wgrad.hip

}  // namespace micf
using namespace micf;
#define RC(e) ((e) == hipSuccess ? MICF_OK : MICF_ELAUNCH)

static bool conv3_args_ok(int B, int D, int H, int W, int N, int c1, int c2) {
  return B > 0 && D > 0 && H > 0 && W > 0 && N > 0 && c1 > 0 && c2 >= 0 && (int64_t)B * D * H * W < (1LL << 31) &&
         (int64_t)27 * (c1 + c2) * (int64_t)(N > c1 + c2 ? N : c1 + c2) < (1LL << 31);
}

static Conv3In make_in(const float* x1, int c1, const float* x2, int c2, const Geo& g) {
  const int vec = (c1 % 4 == 0) && (c2 % 4 == 0) && aligned16(x1) && (!x2 || aligned16(x2));
  return Conv3In{x1, x2 ? x2 : x1, c1, c2 > 0 ? c2 : 1, c1 + c2, g, (int)g.tokens(), vec, FastDiv((uint32_t)(c1 + c2))};
}

extern "C" int64_t micf_conv3_fwd_workspace(int N, int c1, int c2) { return conv3_fwdx_workspace(N, c1, c2); }

// ---- both re-laid-out copies of a list of few-output-channel conv weights in ONE launch (once per step: the weights only change
// in Adam).  fwd = [chunk][tap][16 n][16 c] (conv3_fwdx.hip), bwd = [tap][c][16 n] (conv3_bwdx.hip).
namespace micf {
constexpr int kC3PrepMax = 64;
struct C3PrepArgs {
  const float* w[kC3PrepMax]; float* fwd[kC3PrepMax]; float* bwd[kC3PrepMax];
  int N[kC3PrepMax], Cin[kC3PrepMax], end[kC3PrepMax];          // end: running total of 16-channel chunks (= workgroups)
  int n;
};
// Workgroup = (weight, 16-channel chunk).  For every output channel n the chunk's 16 x 27 weights are CONTIGUOUS in w
// ([N][Cin][27]): they come in with coalesced loads, sit in LDS as [n][c][tap], and every element of the four layouts that
// belongs to this chunk -- each a contiguous run of the destination -- is written from there (the per-element gather this
// replaces read w with a stride of 27 floats once per layout: 0.43 ms per step).
__global__ void __launch_bounds__(256) conv3_weight_prep_kernel(const C3PrepArgs a) {
  __shared__ float ws[16][16 * 27 + 1];
  const int wgi = blockIdx.x;
  int k = 0;
  while (k < a.n - 1 && wgi >= a.end[k]) ++k;
  const int chunk = wgi - (k ? a.end[k - 1] : 0);
  const float* __restrict__ w = a.w[k];
  const int N = a.N[k], Cin = a.Cin[k], chunks = (Cin + 15) / 16;
  const int c0 = chunk * 16, nc = min(16, Cin - c0);
  for (int i = threadIdx.x; i < 16 * 432; i += 256) {
    const int n = i / 432, r = i % 432;                               // r = c * 27 + tap
    ws[n][r] = (n < N && r < nc * 27) ? w[((int64_t)n * Cin + c0) * 27 + r] : 0.f;
  }
  __syncthreads();
  auto bf = [](float v) { return (uint16_t)(pack_bf16(v, 0.f) & 0xFFFFu); };
  if (float* fwd = a.fwd[k]) {
    float* f32p = fwd + (int64_t)chunk * 27 * 256;                    // [tap][16 n][16 c]
    for (int i = threadIdx.x; i < 27 * 256; i += 256) {
      const int c = i & 15, n = (i >> 4) & 15, tap = i >> 8;
      f32p[i] = ws[n][c * 27 + tap];
    }
    uint16_t* hp = reinterpret_cast<uint16_t*>(fwd + conv3_fwd_layout_f32(Cin)) + (int64_t)chunk * 14 * 512;   // [pair][16 n][4 lr][8]
    for (int i = threadIdx.x; i < 14 * 512; i += 256) {
      const int e = i & 7, lr = (i >> 3) & 3, n = (i >> 5) & 15, p = i >> 9;
      const int tap = 2 * p + (e >> 2), c = 4 * lr + (e & 3);
      hp[i] = bf(tap < 27 ? ws[n][c * 27 + tap] : 0.f);
    }
  }
  if (float* bwd = a.bwd[k]) {
    const int O = Cin;                                                 // (multiple of 16 for the direct kernel; tails are masked)
    for (int i = threadIdx.x; i < 27 * 256; i += 256) {                // [tap][O][16 n]: this chunk's run of 16 c x 16 n per tap
      const int n = i & 15, c = (i >> 4) & 15, tap = i >> 8;
      if (c < nc) bwd[((int64_t)tap * O + c0 + c) * 16 + n] = ws[n][c * 27 + tap];
    }
    uint16_t* hb = reinterpret_cast<uint16_t*>(bwd + conv3_bwd_layout_f32(O));                                  // [pair][O][4 lr][8]
    for (int i = threadIdx.x; i < 14 * 512; i += 256) {
      const int e = i & 7, lr = (i >> 3) & 3, c = (i >> 5) & 15, p = i >> 9;
      const int tap = 2 * p + (e >> 2), n = 4 * lr + (e & 3);
      if (c < nc) hb[(((int64_t)p * O + c0 + c) * 4 + lr) * 8 + e] = bf(tap < 27 ? ws[n][c * 27 + tap] : 0.f);
    }
  }
  (void)chunks;
}
}  // namespace micf

extern "C" int micf_conv3_weight_prep_grouped(const micf_conv3_prep_item* items, int n, micf_stream_t stream) {
  if (n < 0 || (n > 0 && !items)) return MICF_EINVAL;
  for (int first = 0; first < n; first += kC3PrepMax) {
    const int cnt = (n - first < kC3PrepMax) ? n - first : kC3PrepMax;
    C3PrepArgs a;
    a.n = cnt;
    int blocks = 0;
    for (int k = 0; k < cnt; ++k) {
      const micf_conv3_prep_item& it = items[first + k];
      if (!it.w || (!it.fwd && !it.bwd) || it.N <= 0 || it.N > 16 || it.Cin <= 0) return MICF_EINVAL;
      a.w[k] = it.w; a.fwd[k] = it.fwd; a.bwd[k] = it.bwd; a.N[k] = it.N; a.Cin[k] = it.Cin;
      blocks += (it.Cin + 15) / 16;
      a.end[k] = blocks;
    }
    hipLaunchKernelGGL(conv3_weight_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
  }
  return MICF_OK;
}

extern "C" int micf_conv3_fwd(const float* x1, int c1, const float* x2, int c2, const float* w, const float* bias,
                              float* y, int y_layout, int B, int D, int H, int W, int N, float* workspace,
                              int64_t workspace_floats, int prepared, int dtype, micf_stream_t stream) {
  if (!x1 || !w || !y || (c2 > 0 && !x2) || !conv3_args_ok(B, D, H, W, N, c1, c2)) return MICF_EINVAL;
  if (y_layout == 0 && workspace && workspace_floats >= conv3_fwdx_workspace(N, c1, c2)) {
    // few output channels, channels-last: direct convolution with pre-transposed weights streamed from L2 (conv3_fwdx.hip)
    const int rc = conv3_fwd_x(x1, c1, x2, c2, w, bias, y, workspace, B, D, H, W, N, (hipStream_t)stream, dtype, prepared);
    if (rc != MICF_EUNSUPPORTED) return rc;
  }
  if (y_layout == 0) {   // LDS-halo direct convolution on the matrix cores: measured faster than the implicit GEMM for the
                         // channels-last offset conv (126 vs 224 us at 32^3 x 96 ch, batch 2); the NCDHW out_conv stays on the GEMM
    const int rc = conv3_fwd_direct(x1, c1, x2, c2, w, bias, y, y_layout, B, D, H, W, N, (hipStream_t)stream);
    if (rc != MICF_EUNSUPPORTED) return rc;
  }
  const Geo g{B, D, H, W};
  const int Cin = c1 + c2;
  const int64_t T = g.tokens();
  const int64_t DHW = (int64_t)D * H * W;
  hipStream_t s = (hipStream_t)stream;
  Conv3TokT tok{make_in(x1, c1, x2, c2, g)};
  auto wgt = make_elem<false>(Conv3WFwd{w, Cin, FastDiv((uint32_t)Cin)}, N);
  if (y_layout == 0) {          // C[i = n, j = token]
    const int splits = pick_splits(N, T, 27 * Cin);
    if (splits > 1) {           // small token grids (8^3, 4^3 stages): split the 27*Cin reduction over workgroups
      if (hipMemsetAsync(y, 0, sizeof(float) * (size_t)T * N, s) != hipSuccess) return MICF_ELAUNCH;
      return RC(launch_gemm(wgt, tok, Conv3FwdClSplitEpi{bias, y, N}, N, T, 27 * Cin, splits, s));
    }
    const int vec = (N % 4 == 0) && aligned16(y) && (!bias || aligned16(bias));
    return RC(launch_gemm(wgt, tok, Conv3FwdClEpi{bias, y, N, vec}, N, T, 27 * Cin, 1, s));
  }
  // NCDHW: C[i = token (voxels contiguous), j = n]
  const int vec = (DHW % 4 == 0) && aligned16(y);
  return RC(launch_gemm(tok, wgt, Conv3FwdPlanesEpi{bias, y, N, DHW, FastDiv((uint32_t)DHW), vec}, (int)T, N, 27 * Cin, 1, s));
}

extern "C" int64_t micf_conv3_bwd_data_workspace(int N, int c1, int c2) {
  return (N > 0 && N <= 16 && c1 + c2 > 0) ? conv3_bwd_layout_floats(c1 + c2) : 0;
}

extern "C" int micf_conv3_bwd_data(const float* dy, int dy_layout, const float* w, float* dx1, int c1, int acc1,
                                   float* dx2, int c2, int acc2, int B, int D, int H, int W, int N, float* workspace,
                                   int64_t workspace_floats, int prepared, int dtype, micf_stream_t stream) {
  if (!dy || !w || (!dx1 && !dx2) || !conv3_args_ok(B, D, H, W, N, c1, c2)) return MICF_EINVAL;
  if (dy_layout == 0 && workspace && workspace_floats >= micf_conv3_bwd_data_workspace(N, c1, c2)) {
    // few dy channels, channels-last: direct convolution with pre-transposed weights (conv3_bwdx.hip)
    const int rc = conv3_bwd_data_x(dy, w, workspace, dx1, c1, acc1, dx2, c2, acc2, B, D, H, W, N, (hipStream_t)stream, dtype, prepared);
    if (rc != MICF_EUNSUPPORTED) return rc;
  }
  const Geo g{B, D, H, W};
  const int Cin = c1 + c2;
  const int64_t T = g.tokens();
  // C[i = c, j = token] = sum_{tap, n} w[n][c][tap] * dy[token - (tap - 1), n]
  auto pa = make_elem<false>(Conv3WBwd{w, Cin, FastDiv((uint32_t)N)}, Cin);
  const Conv3DyGather gat{dy, dy_layout, N, g, (int64_t)D * H * W, FastDiv((uint32_t)N)};
  const int vec = (c1 % 4 == 0) && (c2 % 4 == 0) && (!dx1 || aligned16(dx1)) && (!dx2 || aligned16(dx2));
  Conv3DataEpi epi{dx1, dx2, c1, c2 > 0 ? c2 : 1, acc1, acc2, vec};
  hipStream_t s = (hipStream_t)stream;
  if (dy_layout == 0) return RC(launch_gemm(pa, make_elem<true>(gat, (int)T), epi, Cin, T, 27 * N, 1, s));   // n contiguous
  return RC(launch_gemm(pa, make_elem<false>(gat, (int)T), epi, Cin, T, 27 * N, 1, s));                      // voxels contiguous
}

// Several offset-conv layers of ONE shape (the two modalities' heads of every cross pair a flush hands over) in one launch + one
// reduce; shapes the MFMA kernel does not take run item by item through micf_conv3_bwd_weight with the same workspace.
extern "C" int64_t micf_conv3_bwd_weight_grouped_workspace(int n, int B, int D, int H, int W, int N, int c1, int c2) {
  if (n <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || c1 < 0 || c2 < 0) return 0;
  const int64_t one = conv3_wgradx_workspace(B, D, H, W, N, c1, c2);
  const int64_t all = conv3_wgradx_workspace(B, D, H, W, N, c1, c2, n);
  return all > one ? all : one;
}

extern "C" int64_t micf_conv3_bwd_weight_workspace(int B, int D, int H, int W, int N, int c1, int c2) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || c1 < 0 || c2 < 0) return 0;
  return conv3_wgradx_workspace(B, D, H, W, N, c1, c2);
}

extern "C" int micf_conv3_bwd_weight(const float* dy, int dy_layout, const float* x1, int c1, const float* x2, int c2,
                                     float* dw, float* dbias, int B, int D, int H, int W, int N, float* workspace,
                                     int64_t workspace_floats, int dtype, micf_stream_t stream) {
  if (!dy || !x1 || !dw || (c2 > 0 && !x2) || !conv3_args_ok(B, D, H, W, N, c1, c2)) return MICF_EINVAL;
  if (dy_layout == 0 && workspace) {   // 16 dy channels, channels-last: register-resident MFMA kernel (conv3_wgradx.hip)
    const int rc = conv3_wgradx(dy, x1, c1, x2, c2, dw, dbias, B, D, H, W, N, workspace, workspace_floats, (hipStream_t)stream, dtype);
    if (rc != MICF_EUNSUPPORTED) return rc;
  }
  {   // direct LDS-tiled kernel for the shapes of the model (N = 16 offset conv, N = 8 out_conv)
    const int rc = conv3_wgrad_direct(dy, dy_layout, x1, c1, x2, c2, dw, dbias, B, D, H, W, N, (hipStream_t)stream);
    if (rc != MICF_EUNSUPPORTED) return rc;
  }
  const Geo g{B, D, H, W};
  const int Cin = c1 + c2;
  const int64_t T = g.tokens();
  const int64_t DHW = (int64_t)D * H * W;
  // dW^T[i = tap*Cin + c, j = n] = sum_t in[nbr(t, tap), c] * dy[t, n]: the long 27*Cin axis is the tile's I side, the narrow
  // N (16 / 8) its J side; the reduction over tokens is split across workgroups (atomic epilogue).
  Conv3ColD pa{make_in(x1, c1, x2, c2, g), 27 * Cin};
  Conv3WgtEpi epi{dw, Cin, FastDiv((uint32_t)Cin)};
  hipStream_t s = (hipStream_t)stream;
  const int splits = pick_splits(27 * Cin, N, T);
  hipError_t e;
  if (dy_layout == 0) {
    // dbias = column sums of the dy slab in LDS (colsum side 2)
    e = launch_gemm(pa, rows_d(dy, N, N), epi, 27 * Cin, N, (int)T, splits, s, dbias, dbias ? 2 : 0);
    return RC(e);
  }
  e = launch_gemm(pa, make_elem<true>(Conv3DyPlanes{dy, N, DHW, FastDiv((uint32_t)DHW)}, N), epi, 27 * Cin, N, (int)T, splits, s,
                  dbias, dbias ? 2 : 0);
  return RC(e);
}

extern "C" int micf_conv3_bwd_weight_grouped(const micf_conv3_wgrad_item* items, int n, int c1, int c2, int B, int D, int H, int W,
                                             int N, float* workspace, int64_t workspace_floats, int dtype, micf_stream_t stream) {
  if (!items || n <= 0 || !conv3_args_ok(B, D, H, W, N, c1, c2)) return MICF_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!items[i].dy || !items[i].x1 || !items[i].dw || (c2 > 0 && !items[i].x2)) return MICF_EINVAL;
  constexpr int kMax = 12;
  if (workspace && n > 1) {
    bool all_ok = true;
    for (int base = 0; base < n && all_ok; base += kMax) {
      const int m = n - base < kMax ? n - base : kMax;
      const float* dy[kMax]; const float* x1[kMax]; const float* x2[kMax]; float* dw[kMax]; float* db[kMax];
      for (int i = 0; i < m; ++i) {
        const micf_conv3_wgrad_item& it = items[base + i];
        dy[i] = it.dy; x1[i] = it.x1; x2[i] = c2 > 0 ? it.x2 : nullptr; dw[i] = it.dw; db[i] = it.dbias;
      }
      const int rc = conv3_wgradx_items(dy, x1, x2, dw, db, m, c1, c2, B, D, H, W, N, workspace, workspace_floats,
                                        (hipStream_t)stream, dtype);
      if (rc == MICF_EUNSUPPORTED && base == 0) { all_ok = false; break; }    // (nothing launched yet: item by item below)
      if (rc != MICF_OK) return rc;
    }
    if (all_ok) return MICF_OK;
  }
  for (int i = 0; i < n; ++i) {
    const int rc = micf_conv3_bwd_weight(items[i].dy, 0, items[i].x1, c1, items[i].x2, c2, items[i].dw, items[i].dbias, B, D, H, W, N,
                                         workspace, workspace_floats, dtype, stream);
    if (rc != MICF_OK) return rc;
  }
  return MICF_OK;
}
