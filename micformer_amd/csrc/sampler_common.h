// sampler_common.h -- the per-token arithmetic of the offset head + deformable sampling (MS.py:263-273, 315-317, 326-337, 360-384;
// STN.py:9-32), shared by offset_sample.hip (stand-alone kernels, backward) and block_fwd.hip (the forward sampling fused into the
// cross block's launch).  One 16-lane group per token: lane k holds channel k of the 16-wide head.
#pragma once
#include "common.h"

namespace micf {

constexpr int kHid = 16;

__device__ __forceinline__ float sum16(float v) {   // all-reduce inside each 16-lane group
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct Taps {
  float cz, cy, cx;        // continuous source index per axis
  float z0, y0, x0;        // floor
  bool finite;
};

__device__ __forceinline__ float src_coord(int idx, float flow, int S) {
  const float nw = (float)idx + flow;                       // STN.py:20
  const float n = 2.f * (nw / (float)(S - 1) - 0.5f);       // STN.py:24   (S == 1: division by zero, kept)
  return ((n + 1.f) * (float)S - 1.f) / 2.f;                // grid_sample un-normalise, align_corners=False
}

__device__ __forceinline__ Taps make_taps(int d, int h, int w, const float* flow, int D, int H, int W) {
  Taps t;
  t.cz = src_coord(d, flow[0], D);
  t.cy = src_coord(h, flow[1], H);
  t.cx = src_coord(w, flow[2], W);
  t.finite = isfinite(t.cz) && isfinite(t.cy) && isfinite(t.cx);
  t.z0 = floorf(t.cz); t.y0 = floorf(t.cy); t.x0 = floorf(t.cx);
  return t;
}

// corner (dz,dy,dx): validity + linear token offset inside the sample + weight parts
__device__ __forceinline__ bool corner(const Taps& t, int dz, int dy, int dx, int D, int H, int W, int& lin) {
  const float z = t.z0 + dz, y = t.y0 + dy, x = t.x0 + dx;
  if (!(z >= 0.f && z <= (float)(D - 1) && y >= 0.f && y <= (float)(H - 1) && x >= 0.f && x <= (float)(W - 1))) return false;
  lin = ((int)z * H + (int)y) * W + (int)x;
  return true;
}

// 16-wide head: every lane k = lane & 15 holds channel k; returns off[3] (same in all lanes), fills per-lane pieces
__device__ __forceinline__ void head_fwd(const float* hrow, const float* ln_g, const float* ln_b, const float* w1, float eps,
                                         int k, float& xh, float& rs, float& ln, float& gl, float off[3]) {
  const float hv = hrow[k];
  const float mu = sum16(hv) * (1.f / kHid);
  const float dv = hv - mu;
  rs = 1.0f / sqrtf(sum16(dv * dv) * (1.f / kHid) + eps);
  xh = dv * rs;
  ln = xh * ln_g[k] + ln_b[k];
  gl = gelu_f(ln);
#pragma unroll
  for (int a = 0; a < 3; ++a) off[a] = sum16(w1[a * kHid + k] * gl);
}


}  // namespace micf
