// head_tail.hip -- reverse_patch_embedding followed by Head.out_conv, as ONE linear map on the coarse token grid.
// Reference ops replaced: nn.ConvTranspose3d(2E -> E/2, k = s = P) (MS.py:1037) immediately followed by
// nn.Conv3d(E/2 -> classes, 3, padding=1) (MS.py:1053) -- there is no norm or activation between them.
//
// z[u, c] = b_up[c] + sum_k W_up[k, c, u mod P] x[u div P, k]            (fine voxel u, E/2 channels: 400 MB at 128^3, batch 2)
// y[u, o] = b_out[o] + sum_{t, c} W_out[o, c, t] z[u + t - 1, c]         (zero padding of z outside the volume)
// Both are linear, so y restricted to the (P+2)^3 fine voxels around coarse voxel q is a linear function of x[q]:
//   T[q, (f, o)] = Bf[f, o] + sum_k Wb[f, o, k] x[q, k],        f in [0, P+2)^3  <->  fine voxel u = P*q - 1 + f
//   Wb[f, o, k] = sum_{(p, t): p - t + 2 = f per axis} sum_c W_out[o, c, t] W_up[k, c, p],   Bf likewise with b_up[c]
//   y[u, o]     = b_out[o] + sum of the (1, 2, 4 or 8) T entries of the in-volume coarse voxels whose patch touches u
// which needs 3.4x fewer flops than the two convolutions, never materialises z, and turns the work into one plain GEMM
// [tokens, 2E] x [2E, (P+2)^3 * classes] (the LDS-DMA core, linear.hip) plus a gather ("col2im").  Backward is the
// transpose: U = im2col(dy), dx = U Wb (GEMM), dWb = U^T x and dBf = colsum(U) (GEMM), and the chain rule through the
// composition gives dW_up, db_up, dW_out, db_out (tiny contractions).  Summation order differs from the reference's two
// convolutions only by fp32 re-association.
#include "common.h"

namespace micf {

// one thread per (f, o, k): k fastest.  k == Ci computes the bias composite Bf[f, o] (W_up row replaced by b_up).
__global__ void __launch_bounds__(256) tail_compose_kernel(const float* __restrict__ w_up, const float* __restrict__ b_up,
                                                           const float* __restrict__ w_out, float* __restrict__ wb,
                                                           float* __restrict__ bf, int Ci, int Cm, int Co, int P,
                                                           const float* __restrict__ w_up_t) {
  const int F = P + 2, P3 = P * P * P;
  const int64_t total = (int64_t)F * F * F * Co * (Ci + 1);
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int k = (int)(id % (Ci + 1));
  int64_t r = id / (Ci + 1);
  const int o = (int)(r % Co); r /= Co;
  const int fw = (int)(r % F); r /= F;
  const int fh = (int)(r % F);
  const int fd = (int)(r / F);
  float acc = 0.f;
  for (int pd = max(fd - 2, 0); pd <= min(fd, P - 1); ++pd)
    for (int ph = max(fh - 2, 0); ph <= min(fh, P - 1); ++ph)
      for (int pw = max(fw - 2, 0); pw <= min(fw, P - 1); ++pw) {
        const int t = ((pd + 2 - fd) * 3 + (ph + 2 - fh)) * 3 + (pw + 2 - fw);
        const int p = (pd * P + ph) * P + pw;
#pragma unroll 24       // (every (p, t) pair was three dependent L2 round trips at 8 loads in flight: the whole c loop of the reference's Cm = 24 at once)
        for (int c = 0; c < Cm; ++c) {
          // (w_up_t = w_up as [Cm * P^3][Ci]: the lanes of a wave are consecutive k -> one coalesced load instead of 64 lines)
          const float wu = (k < Ci) ? (w_up_t ? w_up_t[((int64_t)c * P3 + p) * Ci + k] : w_up[((int64_t)k * Cm + c) * P3 + p]) : b_up[c];
          acc += w_out[((int64_t)o * Cm + c) * 27 + t] * wu;
        }
      }
  const int64_t row = (((int64_t)fd * F + fh) * F + fw) * Co + o;
  if (k < Ci) wb[row * Ci + k] = acc;
  else bf[row] = acc;
}

// y[b, o, u] = b_out[o] + sum of T over the coarse voxels whose (P+2)^3 patch covers u.  One thread per fine voxel.
// sw != nullptr: sliding-window inference (utils.py:226-234) -- batch entry b is WINDOW b of a volume: its logits are ADDED into the
// fp32 volume accumulator y [VB, Co, VD, VH, VW] at the window's origin sw[4 b ..] = {volume sample, z0, y0, x0} (device
// memory, so a captured predictor graph can be replayed with new coordinates) and the visit count of the voxel is bumped: the
// accumulate / count epilogue of the reference's inferer fused into the logits store (windows of one batch may overlap: fp32 atomics).
template <int CO>
__global__ void __launch_bounds__(256) tail_col2im_kernel(const float* __restrict__ T, const float* __restrict__ b_out,
                                                          float* __restrict__ y, int B, int Dc, int Hc, int Wc, int Co_rt, int P,
                                                          const int32_t* __restrict__ sw, float* __restrict__ count, int VD, int VH,
                                                          int VW) {
  const int Co = CO ? CO : Co_rt;
  const int F = P + 2, Df = Dc * P, Hf = Hc * P, Wf = Wc * P;
  const int64_t plane = (int64_t)Df * Hf * Wf;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)B * plane) return;
  const int b = (int)(id / plane);
  int64_t r = id % plane;
  const int uw = (int)(r % Wf); r /= Wf;
  const int uh = (int)(r % Hf);
  const int ud = (int)(r / Hf);
  // per axis: the owning coarse voxel (f = residue + 1) and possibly one neighbour
  int qd[2], fd[2], nd = 1, qh[2], fh[2], nh = 1, qw[2], fw[2], nw = 1;
  { const int q = ud / P, rr = ud % P; qd[0] = q; fd[0] = rr + 1;
    if (rr == 0 && q > 0) { qd[1] = q - 1; fd[1] = P + 1; nd = 2; } else if (rr == P - 1 && q < Dc - 1) { qd[1] = q + 1; fd[1] = 0; nd = 2; } }
  { const int q = uh / P, rr = uh % P; qh[0] = q; fh[0] = rr + 1;
    if (rr == 0 && q > 0) { qh[1] = q - 1; fh[1] = P + 1; nh = 2; } else if (rr == P - 1 && q < Hc - 1) { qh[1] = q + 1; fh[1] = 0; nh = 2; } }
  { const int q = uw / P, rr = uw % P; qw[0] = q; fw[0] = rr + 1;
    if (rr == 0 && q > 0) { qw[1] = q - 1; fw[1] = P + 1; nw = 2; } else if (rr == P - 1 && q < Wc - 1) { qw[1] = q + 1; fw[1] = 0; nw = 2; } }
  const int64_t ldT = (int64_t)F * F * F * Co;
  constexpr int MAXCO = CO ? CO : 32;
  float acc[MAXCO];
#pragma unroll
  for (int o = 0; o < MAXCO; ++o) acc[o] = (o < Co) ? b_out[o] : 0.f;
  for (int a = 0; a < nd; ++a)
    for (int c = 0; c < nh; ++c)
      for (int e = 0; e < nw; ++e) {
        const int64_t q = (((int64_t)b * Dc + qd[a]) * Hc + qh[c]) * Wc + qw[e];
        const float* src = T + q * ldT + (((int64_t)fd[a] * F + fh[c]) * F + fw[e]) * Co;
        if (CO && CO % 4 == 0) {
#pragma unroll
          for (int o = 0; o < MAXCO; o += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + o);
            acc[o] += v.x; acc[o + 1] += v.y; acc[o + 2] += v.z; acc[o + 3] += v.w;
          }
        } else {
#pragma unroll
          for (int o = 0; o < MAXCO; ++o) if (o < Co) acc[o] += src[o];
        }
      }
  if (sw) {
    const int vb = sw[4 * b], z = sw[4 * b + 1] + ud, yy = sw[4 * b + 2] + uh, x = sw[4 * b + 3] + uw;
    const int64_t vplane = (int64_t)VD * VH * VW, v = ((int64_t)z * VH + yy) * VW + x;
    float* dst = y + (int64_t)vb * Co * vplane + v;
#pragma unroll
    for (int o = 0; o < MAXCO; ++o) if (o < Co) atomicAdd(dst + (int64_t)o * vplane, acc[o]);
    atomicAdd(count + (int64_t)vb * vplane + v, 1.f);
    return;
  }
  float* dst = y + (int64_t)b * Co * plane + ((int64_t)ud * Hf + uh) * Wf + uw;
#pragma unroll
  for (int o = 0; o < MAXCO; ++o) if (o < Co) dst[(int64_t)o * plane] = acc[o];
}

// U[q, (f, o)] = dy[b, o, P*q - 1 + f] (0 outside the volume).  One thread per (q, f).
template <int CO>
__global__ void __launch_bounds__(256) tail_im2col_kernel(const float* __restrict__ dy, float* __restrict__ U, int B, int Dc,
                                                          int Hc, int Wc, int Co_rt, int P) {
  const int Co = CO ? CO : Co_rt;
  const int F = P + 2, F3 = F * F * F, Df = Dc * P, Hf = Hc * P, Wf = Wc * P;
  const int64_t plane = (int64_t)Df * Hf * Wf;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nq = (int64_t)B * Dc * Hc * Wc;
  if (id >= nq * F3) return;
  const int f = (int)(id % F3);
  int64_t q = id / F3;
  const int fw = f % F, fh = (f / F) % F, fd = f / (F * F);
  const int qw = (int)(q % Wc); int64_t r = q / Wc;
  const int qh = (int)(r % Hc); r /= Hc;
  const int qd = (int)(r % Dc);
  const int b = (int)(r / Dc);
  const int ud = qd * P - 1 + fd, uh = qh * P - 1 + fh, uw = qw * P - 1 + fw;
  const bool in = ud >= 0 && ud < Df && uh >= 0 && uh < Hf && uw >= 0 && uw < Wf;
  const float* src = dy + (int64_t)b * Co * plane + ((int64_t)ud * Hf + uh) * Wf + uw;
  float* dst = U + id * Co;
  constexpr int MAXCO = CO ? CO : 32;
  float v[MAXCO];
#pragma unroll
  for (int o = 0; o < MAXCO; ++o) v[o] = (in && o < Co) ? src[(int64_t)o * plane] : 0.f;
  if (CO && CO % 4 == 0) {
#pragma unroll
    for (int o = 0; o < MAXCO; o += 4) *reinterpret_cast<float4*>(dst + o) = make_float4(v[o], v[o + 1], v[o + 2], v[o + 3]);
  } else {
#pragma unroll
    for (int o = 0; o < MAXCO; ++o) if (o < Co) dst[o] = v[o];
  }
}

// The same for 8 classes, patch 4 and rows of TW = 8 coarse voxels along W: the fine region the 8 patches cover
// ([8 classes][6][6][34] floats = 38 KB) goes through LDS, so dy is read in 34-float runs instead of the 6-float runs of the
// per-(q, f) gather, and U is written as one contiguous 54 KB block.
constexpr int kImTW = 8, kImP = 4, kImF = kImP + 2, kImCO = 8, kImX = kImTW * kImP + 2;
__global__ void __launch_bounds__(256) tail_im2col_rows_kernel(const float* __restrict__ dy, float* __restrict__ U, int B, int Dc,
                                                               int Hc, int Wc) {
  __shared__ float s[kImCO * kImF * kImF * kImX];
  const int Df = Dc * kImP, Hf = Hc * kImP, Wf = Wc * kImP;
  const int64_t plane = (int64_t)Df * Hf * Wf;
  const int wt = Wc / kImTW;
  int r = blockIdx.x;
  const int qw0 = (r % wt) * kImTW; r /= wt;
  const int qh = r % Hc; r /= Hc;
  const int qd = r % Dc;
  const int b = r / Dc;
  const float* src = dy + (int64_t)b * kImCO * plane;
  for (int i = threadIdx.x; i < kImCO * kImF * kImF * kImX; i += 256) {
    const int x = i % kImX;
    int t = i / kImX;
    const int fh = t % kImF; t /= kImF;
    const int fd = t % kImF;
    const int o = t / kImF;
    const int ud = qd * kImP - 1 + fd, uh = qh * kImP - 1 + fh, uw = qw0 * kImP - 1 + x;
    const bool in = ud >= 0 && ud < Df && uh >= 0 && uh < Hf && uw >= 0 && uw < Wf;
    s[i] = in ? src[(int64_t)o * plane + ((int64_t)ud * Hf + uh) * Wf + uw] : 0.f;
  }
  __syncthreads();
  constexpr int F3 = kImF * kImF * kImF;
  const int64_t q0 = (((int64_t)b * Dc + qd) * Hc + qh) * Wc + qw0;
  float* dst = U + q0 * F3 * kImCO;
  for (int i = threadIdx.x; i < kImTW * F3 * (kImCO / 4); i += 256) {       // one float4 (4 classes) per item, contiguous in U
    const int oh = i & 1;
    const int f = (i >> 1) % F3, tq = (i >> 1) / F3;
    const int fw = f % kImF, fh = (f / kImF) % kImF, fd = f / (kImF * kImF);
    const float* p = s + ((4 * oh * kImF + fd) * kImF + fh) * kImX + tq * kImP + fw;
    constexpr int OS = kImF * kImF * kImX;
    typedef float f32x4_nt __attribute__((ext_vector_type(4)));     // 453 MB at 128^3: streamed past the caches
    __builtin_nontemporal_store(f32x4_nt{p[0], p[OS], p[2 * OS], p[3 * OS]}, reinterpret_cast<f32x4_nt*>(dst + 4 * (int64_t)i));
  }
}

// dW_up[k, c, p] += sum_{t, o} dWb[f(p,t), o, k] W_out[o, c, t]      (k == Ci: db_up[c] += sum_p of the same with dBf)
// one thread per (k, c, p), p fastest
__global__ void __launch_bounds__(256) tail_dwup_kernel(const float* __restrict__ dwb, const float* __restrict__ dbf,
                                                        const float* __restrict__ w_out, float* __restrict__ dw_up,
                                                        float* __restrict__ db_up, int Ci, int Cm, int Co, int P) {
  const int F = P + 2, P3 = P * P * P;
  const int64_t total = (int64_t)(Ci + 1) * Cm * P3;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int k = (int)(id % (Ci + 1));            // k fastest: the dWb rows are read coalesced (the single store per thread is strided)
  const int c = (int)((id / (Ci + 1)) % Cm);
  const int p = (int)(id / ((int64_t)(Ci + 1) * Cm));
  const int pw = p % P, ph = (p / P) % P, pd = p / (P * P);
  float acc = 0.f;
  if (Co == 8) {                                  // (the model's head: 8 independent loads per tap in flight, three taps unrolled)
#pragma unroll 3
    for (int t = 0; t < 27; ++t) {
      const int tw = t % 3, th = (t / 3) % 3, td = t / 9;
      const int64_t row0 = (((int64_t)(pd - td + 2) * F + (ph - th + 2)) * F + (pw - tw + 2)) * 8;
      float g[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) g[o] = (k < Ci) ? dwb[(row0 + o) * Ci + k] : dbf[row0 + o];
#pragma unroll
      for (int o = 0; o < 8; ++o) acc += g[o] * w_out[((int64_t)o * Cm + c) * 27 + t];
    }
  } else
  for (int t = 0; t < 27; ++t) {
    const int tw = t % 3, th = (t / 3) % 3, td = t / 9;
    const int64_t row0 = (((int64_t)(pd - td + 2) * F + (ph - th + 2)) * F + (pw - tw + 2)) * Co;
    for (int o = 0; o < Co; ++o) {
      const float g = (k < Ci) ? dwb[(row0 + o) * Ci + k] : dbf[row0 + o];
      acc += g * w_out[((int64_t)o * Cm + c) * 27 + t];
    }
  }
  if (k < Ci) dw_up[((int64_t)k * Cm + c) * P3 + p] += acc;
  else atomicAdd(db_up + c, acc);
}

// dW_out[o, c, t] += sum_p ( sum_k dWb[f(p,t), o, k] W_up[k, c, p] + dBf[f(p,t), o] b_up[c] ):  one WAVE per (o, c, t)
// db_out[o] += sum over the owned f (1 <= f_a <= P) of dBf[f, o]:  the extra blocks after the (o, c, t) ones
__global__ void __launch_bounds__(256) tail_dwout_kernel(const float* __restrict__ dwb, const float* __restrict__ dbf,
                                                         const float* __restrict__ w_up, const float* __restrict__ b_up,
                                                         float* __restrict__ dw_out, float* __restrict__ db_out, int Ci, int Cm,
                                                         int Co, int P, const float* __restrict__ w_up_t) {
  const int F = P + 2, P3 = P * P * P;
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nout = (int64_t)Co * Cm * 27;
  if (wid < nout) {
    const int t = (int)(wid % 27);
    const int c = (int)((wid / 27) % Cm);
    const int o = (int)(wid / (27 * Cm));
    const int tw = t % 3, th = (t / 3) % 3, td = t / 9;
    float acc = 0.f;
    for (int p = 0; p < P3; ++p) {
      const int pw = p % P, ph = (p / P) % P, pd = p / (P * P);
      const int64_t row = (((int64_t)(pd - td + 2) * F + (ph - th + 2)) * F + (pw - tw + 2)) * Co + o;
      for (int k = lane; k < Ci; k += 64)
        acc += dwb[row * Ci + k] * (w_up_t ? w_up_t[((int64_t)c * P3 + p) * Ci + k] : w_up[((int64_t)k * Cm + c) * P3 + p]);
      if (lane == 0) acc += dbf[row] * b_up[c];
    }
    acc = wave_sum(acc);
    if (lane == 0) dw_out[wid] += acc;
  } else if (wid < nout + Co) {
    const int o = (int)(wid - nout);
    float acc = 0.f;
    for (int p = lane; p < P3; p += 64) {
      const int pw = p % P, ph = (p / P) % P, pd = p / (P * P);
      acc += dbf[((((int64_t)(pd + 1) * F + (ph + 1)) * F + (pw + 1))) * Co + o];
    }
    acc = wave_sum(acc);
    if (lane == 0) db_out[o] += acc;
  }
}

// The same as a split-K GEMM: dW_out[(o, t), c] = sum over (p, k) of A[(o, t), (p, k)] B[(p, k), c], A = dWb gathered, B = W_up (k == Ci:
// the dBf / b_up column).  Block = (patch voxel p, 32-wide k chunk): its 27 Co x 32 slice of A and Cm x 32 slice of B go through LDS
// (each read from L2 ONCE per block: the wave-per-output form above re-reads them 24 / 216 times, 255 MB of 4-byte loads), every
// thread accumulates ~20 of the 27 Co Cm outputs and adds them atomically.  db_out rides in the k-chunk-0 blocks.
constexpr int kDwoK = 32, kDwoP = 8;
__global__ void __launch_bounds__(256) tail_dwout_gemm_kernel(const float* __restrict__ dwb, const float* __restrict__ dbf,
                                                              const float* __restrict__ w_up, const float* __restrict__ b_up,
                                                              float* __restrict__ dw_out, float* __restrict__ db_out, int Ci,
                                                              int Cm, int Co, int P, const float* __restrict__ w_up_t) {
  // block = (group of kDwoP patch voxels, k chunk, class o): A = 27 rows (taps) x 32, B = Cm x 32 per patch voxel; a thread owns
  // up to 3 of the 27 Cm outputs of its class across the group's voxels and adds them atomically once (32-way contention, not 256)
  __shared__ float As[27 * (kDwoK + 1)];
  __shared__ float Bs[64 * (kDwoK + 1)];
  const int F = P + 2, P3 = P * P * P;
  const int kc = blockIdx.y, k0 = kc * kDwoK, o = blockIdx.z;
  float acc[3] = {0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int p = blockIdx.x * kDwoP; p < min(P3, (int)(blockIdx.x + 1) * kDwoP); ++p) {
    const int pw = p % P, ph = (p / P) % P, pd = p / (P * P);
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * kDwoK; i += 256) {
      const int t = i / kDwoK, kk = i % kDwoK, k = k0 + kk;
      const int tw = t % 3, th = (t / 3) % 3, td = t / 9;
      const int64_t row = (((int64_t)(pd - td + 2) * F + (ph - th + 2)) * F + (pw - tw + 2)) * Co + o;
      As[t * (kDwoK + 1) + kk] = k < Ci ? dwb[row * Ci + k] : (k == Ci ? dbf[row] : 0.f);
    }
    for (int i = threadIdx.x; i < Cm * kDwoK; i += 256) {
      const int c = i / kDwoK, kk = i % kDwoK, k = k0 + kk;
      Bs[c * (kDwoK + 1) + kk] = k < Ci ? (w_up_t ? w_up_t[((int64_t)c * P3 + p) * Ci + k] : w_up[((int64_t)k * Cm + c) * P3 + p])
                                        : (k == Ci ? b_up[c] : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int j = threadIdx.x + 256 * u;
      if (j < 27 * Cm) {
        const float* ap = As + (j / Cm) * (kDwoK + 1);
        const float* bp = Bs + (j % Cm) * (kDwoK + 1);
        float a = 0.f;
#pragma unroll
        for (int kk = 0; kk < kDwoK; ++kk) a += ap[kk] * bp[kk];
        acc[u] += a;
      }
    }
    if (kc == 0 && threadIdx.x == 0) bsum += dbf[((((int64_t)(pd + 1) * F + (ph + 1)) * F + (pw + 1))) * Co + o];
  }
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int j = threadIdx.x + 256 * u;
    if (j < 27 * Cm) atomicAdd(dw_out + ((int64_t)o * Cm + j % Cm) * 27 + j / Cm, acc[u]);
  }
  if (kc == 0 && threadIdx.x == 0) atomicAdd(db_out + o, bsum);
}

static bool dims_ok(int Ci, int Cm, int Co, int P) { return Ci > 0 && Cm > 0 && Co > 0 && Co <= 32 && P >= 2 && P <= 8; }

}  // namespace micf

using namespace micf;

extern "C" int micf_head_tail_compose(const float* w_up, const float* b_up, const float* w_out, float* wb, float* bf, int Ci,
                                      int Cm, int Co, int P, const float* w_up_t, micf_stream_t stream) {
  if (!w_up || !b_up || !w_out || !wb || !bf) return MICF_EINVAL;
  if (!dims_ok(Ci, Cm, Co, P)) return MICF_EUNSUPPORTED;
  const int F = P + 2;
  const int64_t total = (int64_t)F * F * F * Co * (Ci + 1);
  hipLaunchKernelGGL(tail_compose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_up, b_up,
                     w_out, wb, bf, Ci, Cm, Co, P, w_up_t);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_head_tail_col2im(const float* t, const float* b_out, float* y, int B, int Dc, int Hc, int Wc, int Co, int P,
                                     micf_stream_t stream) {
  if (!t || !b_out || !y || B <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0) return MICF_EINVAL;
  if (!dims_ok(1, 1, Co, P)) return MICF_EUNSUPPORTED;
  const int64_t total = (int64_t)B * Dc * Hc * Wc * P * P * P;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (Co == 8 && aligned16(t)) hipLaunchKernelGGL(tail_col2im_kernel<8>, dim3(blocks), dim3(256), 0, s, t, b_out, y, B, Dc, Hc, Wc, Co, P, nullptr, nullptr, 0, 0, 0);
  else hipLaunchKernelGGL(tail_col2im_kernel<0>, dim3(blocks), dim3(256), 0, s, t, b_out, y, B, Dc, Hc, Wc, Co, P, nullptr, nullptr, 0, 0, 0);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_head_tail_col2im_sw(const float* t, const float* b_out, float* out, float* count, const int32_t* coords, int n,
                                        int Dc, int Hc, int Wc, int Co, int P, int VB, int VD, int VH, int VW, micf_stream_t stream) {
  if (!t || !b_out || !out || !count || !coords || n <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0 || VB <= 0) return MICF_EINVAL;
  if (Dc * P > VD || Hc * P > VH || Wc * P > VW) return MICF_EINVAL;      // (the window must fit the volume; origins are the caller's)
  if (!dims_ok(1, 1, Co, P)) return MICF_EUNSUPPORTED;
  const int64_t total = (int64_t)n * Dc * Hc * Wc * P * P * P;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (Co == 8 && aligned16(t)) hipLaunchKernelGGL(tail_col2im_kernel<8>, dim3(blocks), dim3(256), 0, s, t, b_out, out, n, Dc, Hc, Wc, Co, P, coords, count, VD, VH, VW);
  else hipLaunchKernelGGL(tail_col2im_kernel<0>, dim3(blocks), dim3(256), 0, s, t, b_out, out, n, Dc, Hc, Wc, Co, P, coords, count, VD, VH, VW);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_head_tail_im2col(const float* dy, float* u, int B, int Dc, int Hc, int Wc, int Co, int P,
                                     micf_stream_t stream) {
  if (!dy || !u || B <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0) return MICF_EINVAL;
  if (!dims_ok(1, 1, Co, P)) return MICF_EUNSUPPORTED;
  const int F = P + 2;
  const int64_t total = (int64_t)B * Dc * Hc * Wc * F * F * F;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (Co == kImCO && P == kImP && Wc % kImTW == 0 && aligned16(u))
    hipLaunchKernelGGL(tail_im2col_rows_kernel, dim3((unsigned)((int64_t)B * Dc * Hc * (Wc / kImTW))), dim3(256), 0, s, dy, u, B, Dc, Hc, Wc);
  else if (Co == 8 && aligned16(u)) hipLaunchKernelGGL(tail_im2col_kernel<8>, dim3(blocks), dim3(256), 0, s, dy, u, B, Dc, Hc, Wc, Co, P);
  else hipLaunchKernelGGL(tail_im2col_kernel<0>, dim3(blocks), dim3(256), 0, s, dy, u, B, Dc, Hc, Wc, Co, P);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_head_tail_decompose(const float* dwb, const float* dbf, const float* w_up, const float* b_up,
                                        const float* w_out, float* dw_up, float* db_up, float* dw_out, float* db_out, int Ci,
                                        int Cm, int Co, int P, const float* w_up_t, micf_stream_t stream) {
  if (!dwb || !dbf || !w_up || !b_up || !w_out || !dw_up || !db_up || !dw_out || !db_out) return MICF_EINVAL;
  if (!dims_ok(Ci, Cm, Co, P)) return MICF_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n1 = (int64_t)(Ci + 1) * Cm * P * P * P;
  hipLaunchKernelGGL(tail_dwup_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, dwb, dbf, w_out, dw_up, db_up, Ci, Cm, Co, P);
  if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
  if (Cm <= 64 && 27 * Cm <= 3 * 256) {
    hipLaunchKernelGGL(tail_dwout_gemm_kernel, dim3((P * P * P + kDwoP - 1) / kDwoP, (Ci + 1 + kDwoK - 1) / kDwoK, Co), dim3(256), 0, s,
                       dwb, dbf, w_up, b_up, dw_out, db_out, Ci, Cm, Co, P, w_up_t);
    MICF_RETURN_LAUNCH();
  }
  const int64_t n2 = (int64_t)Co * Cm * 27 + Co;
  hipLaunchKernelGGL(tail_dwout_kernel, dim3((unsigned)((n2 + 3) / 4)), dim3(256), 0, s, dwb, dbf, w_up, b_up, dw_out, db_out, Ci, Cm, Co, P, w_up_t);
  MICF_RETURN_LAUNCH();
}
