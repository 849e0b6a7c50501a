// loss_optim.hip -- MDiceLoss forward/backward (dice.py:130-166), argmax + meandice (train.py:305, 392-407) and the
// fused Adam + cosine-LR step (train.py:114, 148, 200-207) as HBM-streaming kernels.
#include "common.h"
#include "loss_terms.h"

namespace micf {

// ---- loss forward: per (b, channel) plane partial sums {sum p t, sum p^2, sum t^2, sum bce} -> double atomics
// Target: one-hot float planes [B, K, V] (what train.py:177 feeds) or, LABEL, the uint8 class map [B, V] it was expanded from
// (t = label == channel): 8x fewer target bytes in HBM and over PCIe.
template <bool LABEL>
__global__ void __launch_bounds__(256) dice_bce_partial_kernel(const float* __restrict__ z, const void* __restrict__ tv,
                                                               double* __restrict__ sums, int K, int64_t V, int chunks) {
  const int plane = blockIdx.y;             // b*K + ch
  const int ch = plane % K;
  const int64_t per = (V + chunks - 1) / chunks;
  const int64_t v0 = blockIdx.x * per, v1 = (v0 + per < V) ? v0 + per : V;
  const float* zp = z + (int64_t)plane * V;
  const float* tp = static_cast<const float*>(tv) + (int64_t)plane * V;
  const uint8_t* lp8 = static_cast<const uint8_t*>(tv) + (int64_t)(plane / K) * V;
  // (the terms on the hardware exp / log / rcp, loss_terms.h: the libm forms made this reduction VALU-bound -- 71 us for 268 MB at
  //  128^3 x 8 x 2 -- and two independent 16-byte load pairs per iteration keep more than one HBM round trip in flight)
  LossAcc la{0.f, 0.f, 0.f, 0.f};
  // 16-byte accesses where the plane and the chunk allow it (V and the chunk length multiples of 4: every training shape)
  const bool vec = !LABEL && (V & 3) == 0 && (per & 3) == 0 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(tv)) & 15) == 0;
  if (vec) {
    int64_t i = v0 + 4 * threadIdx.x;
    for (; i + 1024 < v1; i += 2048) {
      const float4 z4 = *reinterpret_cast<const float4*>(zp + i), t4 = *reinterpret_cast<const float4*>(tp + i);
      const float4 y4 = *reinterpret_cast<const float4*>(zp + i + 1024), u4 = *reinterpret_cast<const float4*>(tp + i + 1024);
      la.term(z4.x, t4.x); la.term(z4.y, t4.y); la.term(z4.z, t4.z); la.term(z4.w, t4.w);
      la.term(y4.x, u4.x); la.term(y4.y, u4.y); la.term(y4.z, u4.z); la.term(y4.w, u4.w);
    }
    for (; i < v1; i += 1024) {
      const float4 z4 = *reinterpret_cast<const float4*>(zp + i), t4 = *reinterpret_cast<const float4*>(tp + i);
      la.term(z4.x, t4.x); la.term(z4.y, t4.y); la.term(z4.z, t4.z); la.term(z4.w, t4.w);
    }
  } else {
    for (int64_t i = v0 + threadIdx.x; i < v1; i += 256) la.term(zp[i], LABEL ? (lp8[i] == ch ? 1.f : 0.f) : tp[i]);
  }
  float a = la.a, b = la.b, c = la.c, d = la.d;
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
  __shared__ float part[4][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { part[wave][0] = a; part[wave][1] = b; part[wave][2] = c; part[wave][3] = d; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    atomicAdd(sums + ch * 4 + q, (double)part[0][q] + (double)part[1][q] + (double)part[2][q] + (double)part[3][q]);
  }
}

__global__ void dice_bce_final_kernel(const double* __restrict__ sums, float* __restrict__ loss, int K, double count) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double dice = 0.0, ce = 0.0;
  for (int i = 0; i < K; ++i) {
    const double I = sums[i * 4 + 0], P2 = sums[i * 4 + 1], T2 = sums[i * 4 + 2], S = sums[i * 4 + 3];
    dice += 1.0 - (2.0 * I + 1.0) / (P2 + T2 + 1.0);
    ce += S / count;
  }
  *loss = (float)((0.7 * dice + 0.3 * ce) / K);
}

template <bool LABEL>
__global__ void __launch_bounds__(256) dice_bce_bwd_kernel(const float* __restrict__ z, const void* __restrict__ tv,
                                                           const double* __restrict__ sums, const float* __restrict__ gout,
                                                           float* __restrict__ dz, int K, int64_t V, double count) {
  const int plane = blockIdx.y;
  const int ch = plane % K;
  const float I2 = (float)(2.0 * sums[ch * 4 + 0] + 1.0);
  const float den = (float)(sums[ch * 4 + 1] + sums[ch * 4 + 2] + 1.0);
  const float g = gout ? *gout : 1.0f;
  const float cd = g * 0.7f / (float)K, cb = g * 0.3f / (float)K / (float)count;
  const float inv_den2 = 1.0f / (den * den);
  const int64_t base = (int64_t)plane * V;
  const float* t = static_cast<const float*>(tv);
  const uint8_t* l8 = static_cast<const uint8_t*>(tv) + (int64_t)(plane / K) * V;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < V; i += (int64_t)gridDim.x * 256) {
    const float p = 1.0f / (1.0f + expf(-z[base + i]));
    const float tt = LABEL ? (l8[i] == ch ? 1.f : 0.f) : t[base + i];
    // d(1 - (2I+1)/den)/dp = -(2 t den - (2I+1) 2 p)/den^2 ;  dBCE/dp = (p - t)/max(p(1-p), 1e-12)  (ATen)
    const float ddice = -(2.0f * tt * den - I2 * 2.0f * p) * inv_den2;
    const float dbce = (p - tt) / fmaxf((1.0f - p) * p, 1e-12f);
    dz[base + i] = (cd * ddice + cb * dbce) * ((1.0f - p) * p);
  }
}

// ---- argmax over K channels + per-class {pred, label, intersection} counts for meandice
__global__ void __launch_bounds__(256) argmax_count_kernel(const float* __restrict__ z, const uint8_t* __restrict__ label,
                                                           uint8_t* __restrict__ mask, unsigned long long* __restrict__ counts,
                                                           int K, int64_t V, int64_t total) {
  __shared__ unsigned int sc[3 * 32];
  for (int i = threadIdx.x; i < 3 * 32; i += 256) sc[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / V, vx = i - b * V;
    const float* zp = z + b * K * V + vx;
    float best = zp[0]; int arg = 0;
    for (int c = 1; c < K; ++c) { const float v = zp[(int64_t)c * V]; if (v > best) { best = v; arg = c; } }   // first max wins (torch.argmax)
    if (mask) mask[i] = (uint8_t)arg;
    if (label) {
      const int l = label[i];             // labels >= K (e.g. a 255 ignore index) belong to no class: counted nowhere
      atomicAdd(&sc[arg], 1u);
      if (l < K) atomicAdd(&sc[32 + l], 1u);
      if (l == arg) atomicAdd(&sc[64 + arg], 1u);
    }
  }
  __syncthreads();
  if (label)
    for (int i = threadIdx.x; i < 3 * K; i += 256) {
      const unsigned int v = sc[(i / K) * 32 + i % K];
      if (v) atomicAdd(counts + i, (unsigned long long)v);
    }
}
__global__ void meandice_final_kernel(const unsigned long long* __restrict__ counts, double* __restrict__ out, int K) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int c = 1; c < K; ++c)
    s += (2.0 * (double)counts[2 * K + c] + 1e-6) / ((double)counts[c] + (double)counts[K + c] + 1e-6);
  *out = s / (K - 1);
}

// ---- MDiceLoss(_Val).metric (dice.py:168-175 / 223-230, binary_dice metric_mode): per (sample, class) plane the Dice of the
// thresholded prediction sigmoid(z) > 0.5 (i.e. z > 0): 2 sum(p t) / (sum p + sum t); 1 / 0 when the target plane is empty and
// the prediction is / is not.
template <bool LABEL>
__global__ void __launch_bounds__(256) dice_metric_partial_kernel(const float* __restrict__ z, const void* __restrict__ tv,
                                                                  double* __restrict__ sums, int K, int64_t V, int chunks) {
  const int plane = blockIdx.y, ch = plane % K;
  const int64_t per = (V + chunks - 1) / chunks;
  const int64_t v0 = blockIdx.x * per, v1 = (v0 + per < V) ? v0 + per : V;
  const float* zp = z + (int64_t)plane * V;
  const float* tp = static_cast<const float*>(tv) + (int64_t)plane * V;
  const uint8_t* lp8 = static_cast<const uint8_t*>(tv) + (int64_t)(plane / K) * V;
  float a = 0.f, b = 0.f, c = 0.f;                      // sum p t, sum p, sum t
  for (int64_t i = v0 + threadIdx.x; i < v1; i += 256) {
    const float p = zp[i] > 0.f ? 1.f : 0.f;
    const float tt = LABEL ? (lp8[i] == ch ? 1.f : 0.f) : tp[i];
    a += p * tt; b += p; c += tt;
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
__global__ void dice_metric_final_kernel(const double* __restrict__ sums, float* __restrict__ out, int planes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes) return;
  const double I = sums[i * 3], P = sums[i * 3 + 1], T = sums[i * 3 + 2];
  out[i] = T == 0.0 ? (P == 0.0 ? 1.f : 0.f) : (float)((2.0 * I) / (P + T));
}

// ---- Adam
struct AdamState { long long step; double lr; };

__global__ void adam_tick_kernel(AdamState* st, double base_lr, double eta_min, long long t_max) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->step += 1;
  // lr in effect for optimizer step number `step` (1-based) = closed-form cosine at scheduler epoch step-1
  st->lr = eta_min + (base_lr - eta_min) * (1.0 + cos(3.14159265358979323846 * (double)(st->step - 1) / (double)t_max)) / 2.0;
}

__device__ __forceinline__ unsigned bf16_pair(float a, float b) {   // round-to-nearest-even, a in the low half (as gemm_dma.h)
  // gfx950 has the conversion in hardware: v_cvt_pk_bf16_f32 (round-to-nearest-even), ONE instruction for the pair -- the
  // integer emulation (add 0x7FFF + lsb, shift, merge) was ~7 VALU instructions per pair and the bound of every bf16 kernel
  typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
  typedef float f32x2_hw __attribute__((ext_vector_type(2)));
  const f32x2_hw f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_hw));
}

__global__ void __launch_bounds__(256) adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        const AdamState* __restrict__ st, float beta1, float beta2, float eps,
                                                        float gscale, uint16_t* __restrict__ p16) {
  const double step = (double)st->step;
  const float bc1 = (float)(1.0 - pow((double)beta1, step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
  const float step_size = (float)(st->lr / (double)bc1);
  // two float4 per thread and round trip: the loads of both are issued before the first store (the kernel is pure HBM streaming,
  // 30 bytes per parameter; bytes in flight per CU are what bounds it)
  const int64_t stride = (int64_t)gridDim.x * 1024;
  for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += 2 * stride) {
    const int64_t i1 = i0 + stride;
    const bool full0 = i0 + 3 < n, full1 = i1 + 3 < n;
    float4 P[2], M[2], V[2], G[2];
    if (full0) { P[0] = *reinterpret_cast<float4*>(p + i0); M[0] = *reinterpret_cast<float4*>(m + i0); V[0] = *reinterpret_cast<float4*>(v + i0); G[0] = *reinterpret_cast<const float4*>(g + i0); }
    if (full1) { P[1] = *reinterpret_cast<float4*>(p + i1); M[1] = *reinterpret_cast<float4*>(m + i1); V[1] = *reinterpret_cast<float4*>(v + i1); G[1] = *reinterpret_cast<const float4*>(g + i1); }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t i = h ? i1 : i0;
      if (i >= n) continue;
    if (h ? full1 : full0) {
      float* pp = &P[h].x; float* mm = &M[h].x; float* vv = &V[h].x; float* gg = &G[h].x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gg[e] *= gscale;                                                  // data parallel: sum over ranks -> mean (1 on one GPU)
        mm[e] = mm[e] + (gg[e] - mm[e]) * (1.0f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
        vv[e] = vv[e] * beta2 + (1.0f - beta2) * gg[e] * gg[e];           // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
        pp[e] = pp[e] - step_size * (mm[e] / denom);                      // param.addcdiv_(exp_avg, denom, -step_size)
      }
      *reinterpret_cast<float4*>(p + i) = P[h]; *reinterpret_cast<float4*>(m + i) = M[h]; *reinterpret_cast<float4*>(v + i) = V[h];
      if (p16) *reinterpret_cast<uint2*>(p16 + i) = make_uint2(bf16_pair(P[h].x, P[h].y), bf16_pair(P[h].z, P[h].w));   // bf16 mirror (RNE)
    } else {
      for (int64_t j = i; j < n; ++j) {
        const float gj = g[j] * gscale;
        m[j] = m[j] + (gj - m[j]) * (1.0f - beta1);
        v[j] = v[j] * beta2 + (1.0f - beta2) * gj * gj;
        p[j] = p[j] - step_size * (m[j] / (sqrtf(v[j]) / bc2_sqrt + eps));
        if (p16) p16[j] = (uint16_t)(bf16_pair(p[j], 0.f) & 0xFFFFu);
      }
    }
    }
  }
}

}  // namespace micf
using namespace micf;

static int dice_fwd(const float* logits, const void* target, bool label, double* sums, float* loss, int B, int K, int64_t V,
                    hipStream_t s) {
  if (!logits || !target || !sums || !loss || B <= 0 || K <= 0 || V <= 0) return MICF_EINVAL;
  if (hipMemsetAsync(sums, 0, sizeof(double) * 4 * K, s) != hipSuccess) return MICF_ELAUNCH;
  int chunks = (int)((V + 32767) / 32768);
  if (label) hipLaunchKernelGGL(dice_bce_partial_kernel<true>, dim3(chunks, B * K), dim3(256), 0, s, logits, target, sums, K, V, chunks);
  else hipLaunchKernelGGL(dice_bce_partial_kernel<false>, dim3(chunks, B * K), dim3(256), 0, s, logits, target, sums, K, V, chunks);
  hipLaunchKernelGGL(dice_bce_final_kernel, dim3(1), dim3(64), 0, s, sums, loss, K, (double)B * (double)V);
  MICF_RETURN_LAUNCH();
}
static int dice_bwd(const float* logits, const void* target, bool label, const double* sums, const float* grad_out, float* dlogits,
                    int B, int K, int64_t V, hipStream_t s) {
  if (!logits || !target || !sums || !dlogits || B <= 0 || K <= 0 || V <= 0) return MICF_EINVAL;
  int bx = (int)((V + 256 * 8 - 1) / (256 * 8));
  if (bx > 4096) bx = 4096;
  if (label) hipLaunchKernelGGL(dice_bce_bwd_kernel<true>, dim3(bx, B * K), dim3(256), 0, s, logits, target, sums, grad_out, dlogits, K, V, (double)B * (double)V);
  else hipLaunchKernelGGL(dice_bce_bwd_kernel<false>, dim3(bx, B * K), dim3(256), 0, s, logits, target, sums, grad_out, dlogits, K, V, (double)B * (double)V);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_dice_bce_fwd(const float* logits, const float* target, double* sums, float* loss, int B, int K, int64_t V,
                                 micf_stream_t stream) {
  return dice_fwd(logits, target, false, sums, loss, B, K, V, (hipStream_t)stream);
}
extern "C" int micf_dice_bce_bwd(const float* logits, const float* target, const double* sums, const float* grad_out,
                                 float* dlogits, int B, int K, int64_t V, micf_stream_t stream) {
  return dice_bwd(logits, target, false, sums, grad_out, dlogits, B, K, V, (hipStream_t)stream);
}
extern "C" int micf_dice_bce_label_fwd(const float* logits, const uint8_t* label, double* sums, float* loss, int B, int K, int64_t V,
                                       micf_stream_t stream) {
  if (K > 255) return MICF_EUNSUPPORTED;
  return dice_fwd(logits, label, true, sums, loss, B, K, V, (hipStream_t)stream);
}
extern "C" int micf_dice_bce_label_bwd(const float* logits, const uint8_t* label, const double* sums, const float* grad_out,
                                       float* dlogits, int B, int K, int64_t V, micf_stream_t stream) {
  if (K > 255) return MICF_EUNSUPPORTED;
  return dice_bwd(logits, label, true, sums, grad_out, dlogits, B, K, V, (hipStream_t)stream);
}

extern "C" int micf_argmax_meandice(const float* logits, const uint8_t* label, uint8_t* mask_out, int64_t* counts, double* out,
                                    int B, int K, int64_t V, micf_stream_t stream) {
  if (!logits || B <= 0 || K <= 0 || K > 32 || V <= 0 || (label && (!counts || !out))) return MICF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (label && hipMemsetAsync(counts, 0, sizeof(int64_t) * 3 * K, s) != hipSuccess) return MICF_ELAUNCH;
  const int64_t total = (int64_t)B * V;
  int blocks = (int)((total + 256 * 8 - 1) / (256 * 8));
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(argmax_count_kernel, dim3(blocks), dim3(256), 0, s, logits, label, mask_out,
                     reinterpret_cast<unsigned long long*>(counts), K, V, total);
  if (label) hipLaunchKernelGGL(meandice_final_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned long long*>(counts), out, K);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_adam_tick(void* state, double base_lr, double eta_min, int64_t t_max, micf_stream_t stream) {
  if (!state || t_max <= 0) return MICF_EINVAL;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (AdamState*)state, base_lr, eta_min, (long long)t_max);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const void* state, float beta1,
                              float beta2, float eps, float grad_scale, void* p_bf16, micf_stream_t stream) {
  if (!p || !g || !m || !v || !state || n < 0) return MICF_EINVAL;
  if (n == 0) return MICF_OK;
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || (reinterpret_cast<uintptr_t>(p_bf16) & 7)) return MICF_EINVAL;
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (const AdamState*)state,
                     beta1, beta2, eps, grad_scale, static_cast<uint16_t*>(p_bf16));
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_dice_metric(const float* logits, const void* target, int target_is_label, double* sums, float* out, int B, int K,
                                int64_t V, micf_stream_t stream) {
  if (!logits || !target || !sums || !out || B <= 0 || K <= 0 || V <= 0) return MICF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, sizeof(double) * 3 * B * K, s) != hipSuccess) return MICF_ELAUNCH;
  int chunks = (int)((V + 65535) / 65536);
  if (chunks > 256) chunks = 256;
  if (target_is_label) hipLaunchKernelGGL(micf::dice_metric_partial_kernel<true>, dim3(chunks, B * K), dim3(256), 0, s, logits, target, sums, K, V, chunks);
  else hipLaunchKernelGGL(micf::dice_metric_partial_kernel<false>, dim3(chunks, B * K), dim3(256), 0, s, logits, target, sums, K, V, chunks);
  hipLaunchKernelGGL(micf::dice_metric_final_kernel, dim3((B * K + 63) / 64), dim3(64), 0, s, sums, out, B * K);
  MICF_RETURN_LAUNCH();
}
