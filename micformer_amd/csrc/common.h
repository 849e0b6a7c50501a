// common.h -- shared helpers for the C-ABI translation units of libmicformer_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/micformer_hip.h"
#include "gemm_core.h"

#define MICF_RETURN_LAUNCH()                          \
  do {                                                \
    return hipGetLastError() == hipSuccess ? MICF_OK : MICF_ELAUNCH; \
  } while (0)

namespace micf {

// Process-global TEST HOOKS and MEASUREMENT PROBES (micf_set_option / micf_get_option, misc.hip).  The product path never sets
// one: the defaults below ARE the product.  A hook selects a slower but equivalent kernel (the parity cross-checks of
// tests/test_gpu_block_wave.py / test_gpu_ops.py) or shrinks a capacity so that an overflow path runs; a probe computes WRONG
// results on purpose (timing only).  Plain ints read at launch time: set them from one thread, between launches.
struct Options {
  int block_wave = 1;          // hook: 0 = the tile-per-workgroup block kernels also at C = 48 in bf16 mode (cross-check of the wave-private kernels)
  int block_recompute_h = 0;   // MEMORY switch: 1 = the fused backward rebuilds the fc1 pre-activation instead of reading a saved copy
  int block_debug = 0;         // PROBE: skip flags of the fused forward's phases; bit 0 = the launch stores nothing (wrong results)
  int sample_tile = 1;         // hook: 0 = the sampler adjoint's d(xa) through the global cell lists / atomics of rounds 1-4
  int sample_e = -1;           // hook: radius of the NEAR neighbourhood of the box gather (-1 = 3; 0 = every token takes the far path)
  int cell_cap = -1;           // hook: capacity of a global cell list (-1 = kCellCap; smaller forces the overflow pass)
  int tile_cap_hits = -1, tile_cap_cell = -1, tile_cap_voxel = -1;   // hook: capacities of the box gather's LDS lists (-1 = 512 / 12 / 6)
};
Options& options();

// token <-> (b, d, h, w) on a channels-last grid
struct Geo {
  int B, D, H, W;
  FastDiv fW, fH, fD;
  Geo() {}
  Geo(int B_, int D_, int H_, int W_) : B(B_), D(D_), H(H_), W(W_), fW((uint32_t)W_), fH((uint32_t)H_), fD((uint32_t)D_) {}
  __host__ __device__ int64_t tokens() const { return (int64_t)B * D * H * W; }
  __device__ __forceinline__ void decode(int t, int& b, int& d, int& h, int& w) const {
    uint32_t q, r;
    fW.divmod((uint32_t)t, q, r); w = (int)r;
    fH.divmod(q, q, r); h = (int)r;
    fD.divmod(q, q, r); d = (int)r; b = (int)q;
  }
  __device__ __forceinline__ int token(int b, int d, int h, int w) const { return ((b * D + d) * H + h) * W + w; }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// column sums of (scale * dy) [M, N] accumulated into out[N] (bias gradients)
int colsum_atomic(const float* dy, const float* scale, int64_t rps, float* out, int64_t M, int N, hipStream_t s);

// ---- offset head (offset_sample.hip / offset_head.hip)
constexpr int kOffsetHidden = 16;   // channels of conv_offset[0]'s output (MS.py:314)
struct CellLists {               // workspace carved by the launcher (all int32)
  int* count;                    // [B*(D+1)*(H+1)*(W+1)] tokens registered per cell (may exceed kCellCap: the rest overflowed)
  int* ovf_count;                // [1]
  int* list;                     // [cells][kCellCap]
  int* ovf;                      // [T] tokens that did not fit their cell's list
  int cap;                       // list entries actually used (kCellCap; smaller only under the test hook "cell_cap")
  float* w8;                     // [T][8] trilinear weight of every corner of a listed token (written with the list entry: the
                                 // gather then costs one load per (voxel, token) instead of re-deriving the taps from the flow)
};
struct SampleFwdSet { const float *h, *ln_g, *ln_b, *w1, *xa; float *flow, *xs; };
struct SampleBwdSet {
  const float *dxs, *h, *ln_g, *ln_b, *w1, *xa, *flow;
  float *dxa, *dh, *dln_g, *dln_b, *dw1;
  CellLists cl;                    // filled in by the launcher
  float* partials;                 // "
};
int offset_sample_fwd_groups(const SampleFwdSet* sets, int n, int B, int D, int H, int W, int C, float eps, hipStream_t stream);
int offset_sample_bwd_groups(SampleBwdSet* sets, int n, int B, int D, int H, int W, int C, float eps, float* workspace,
                             int64_t workspace_floats, hipStream_t s, int phase = 0);
struct SampleFinishCall { const SampleBwdSet* sets; int n, B, D, H, W, C; float* workspace; int64_t workspace_floats; };
int offset_sample_finish_many(const SampleFinishCall* calls, int ncalls, hipStream_t s);

// pointer sets of the grouped (two modalities per launch) forms
struct Conv3FwdSet { const float* x1; const float* x2; const float* w; const float* bias; float* y; float* wt; };
struct Conv3BwdSet { const float* dy; const float* w; float* wt; float* dx1; float* dx2; };
int conv3_fwd_x_groups(const Conv3FwdSet* sets, int n, int c1, int c2, int B, int D, int H, int W, int N, hipStream_t stream, int dtype,
                       int prepared, int y_zeroed);
bool conv3_fwd_x_splits(int B, int D, int H, int W, int c1, int c2);
int conv3_bwd_data_x_groups(const Conv3BwdSet* sets, int ng, int c1, int acc1, int c2, int acc2, int B, int D, int H, int W, int N,
                            hipStream_t stream, int dtype, int prepared);

// direct data gradient for N <= 16 channels-last dy (conv3_bwdx.hip); wt = 27*(c1+c2)*16 floats of scratch
int conv3_bwd_data_x(const float* dy, const float* w, float* wt, float* dx1, int c1, int acc1, float* dx2, int c2, int acc2, int B,
                     int D, int H, int W, int N, hipStream_t stream, int dtype = 0, int prepared = 0);

// direct forward for N <= 16 channels-last outputs (conv3_fwdx.hip); wt = conv3_fwdx_workspace floats of scratch
int64_t conv3_fwdx_workspace(int N, int c1, int c2);
int conv3_fwd_x(const float* x1, int c1, const float* x2, int c2, const float* w, const float* bias, float* y, float* wt, int B, int D,
                int H, int W, int N, hipStream_t stream, int dtype = 0, int prepared = 0);
// MFMA weight gradient for 16 channels-last dy channels (conv3_wgradx.hip)
int64_t conv3_wgradx_workspace(int B, int D, int H, int W, int N, int c1, int c2, int items = 1);
int conv3_wgradx_items(const float* const* dy, const float* const* x1, const float* const* x2, float* const* dw, float* const* dbias,
                       int n, int c1, int c2, int B, int D, int H, int W, int N, float* ws, int64_t ws_floats, hipStream_t stream,
                       int dtype);
int conv3_wgradx(const float* dy, const float* x1, int c1, const float* x2, int c2, float* dw, float* dbias, int B, int D, int H,
                 int W, int N, float* ws, int64_t ws_floats, hipStream_t stream, int dtype = 0);

}  // namespace micf
