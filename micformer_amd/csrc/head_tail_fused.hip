// head_tail_fused.hip -- the composed head tail (head_tail.hip: reverse_patch_embedding, MS.py:1037, followed by Head.out_conv,
// MS.py:1053, as one linear map on the coarse grid) WITHOUT its patch matrices: bf16 mode, 8 classes, patch 4.
//
// head_tail.hip runs the map as  T = x Wb^T  (GEMM, [tokens, 6^3 * 8] = 453 MB at 128^3 / batch 2) + col2im gather, and backward as
// im2col (U, another 453 MB) + two GEMMs: 1.8 GB of HBM traffic for 67 MB of logits, 0.75 ms of the step's critical chain.  Here
// the overlap-add is folded into the GEMM's index arithmetic instead.  With fine voxel u = 4 q + r (r in [0,4)^3):
//   y[u, o] = sum over d in {-1,0,1}^3 of  Wc[d][(r, o), :] . x[q + d, :]          Wc[d][(r, o)] = Wb[(r - 4 d + 1, o)] where that
//   index lies in [0,6)^3 (d = 0: every r; d = -1 along an axis: r = 0 there; d = +1: r = 3), else no term.
// Forward: a workgroup owns 1 x 4 x 16 coarse voxels; their 3 x 6 x 18 halo of x rows sits in LDS as bf16 (out-of-volume rows zero);
// MFMA operand A = a prepacked 16-row slab of Wc[d] (rows = 4 rw x 4 classes, read from L2), operand B = 16 x rows shifted by d;
// the accumulator quad of a lane is then the 4 consecutive fine voxels (rw = 0..3) of ONE class at one coarse voxel and the 16 lanes
// of a row group hold 16 consecutive coarse voxels: every store instruction writes 256-byte runs of the NCDHW logits.  The
// composite bias Bf (which depends on which neighbours are inside the volume) rides in the GEMM: x rows carry a 1.0 column that
// is 0 for out-of-volume rows, Wc a matching column (Bf as a hi + lo bf16 pair: exact to 2^-17), b_out folded into d = 0.
// Backward data: dx[q, :] = sum_{f, o} dy[o, 4 q - 1 + f] Wb[(f, o), :].  The fine region of dy a workgroup's 1 x 2 x 16 coarse
// voxels touch (6 x 10 x 66 voxels x 8 classes) sits in LDS as bf16, class-fastest, so the MFMA operand of a coarse voxel and a
// tap f is ONE 16-byte LDS read (4 taps x 8 classes per k-step); U is never written.
// fp32 (parity) mode keeps head_tail.hip's path.
#include "common.h"
#include "gemm_dma.h"
#include "loss_terms.h"

namespace micf {

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

constexpr int kFSlots = 27 * 32;        // (d + 1 per axis: 27) x rd (4) x rh (4) x class group (2)
constexpr int kXPad = 24;               // bf16 columns after the Ci channels of an LDS x row: [1.0, 1.0, 0 x 14 | 8 unused]
                                        // (row stride (Ci + 24) * 2 bytes = 60 dwords mod 64 for Ci = 96: 16-byte reads of 16 rows hit 16 disjoint bank quads)
constexpr int kBK = 54;                 // backward k-steps: fd (6) x fh pair (3) x fw pair (3); lane group lr = (fh & 1) * 2 + (fw & 1)
constexpr int kDW = 83;                 // LDS slots per fine (d, h) row of dy: w + (w >> 2), w < 66 (a pad slot after every 4 voxels:
                                        // the 16 coarse voxels of an operand read are then 80 bytes apart = disjoint bank quads)

__device__ __forceinline__ uint16_t bf16_bits(float v) { return (uint16_t)(pack_bf16(v, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float bf16_val(uint16_t b) { return __uint_as_float((unsigned)b << 16); }

// ---- weight packs (once per step, off the critical path)
// forward:  wpf[slot][ks][lane][8]   slot = ((((dd*3 + dh)*3 + dw)*4 + rd)*4 + rh)*2 + og,  lane = (li, lr): row li <-> rw = li & 3,
//           class o = 4 og + (li >> 2); k = 32 ks + 8 lr + j;   then the bias slabs wpb[slot][lane][4] = {hi, lo, 0, 0} for lr == 0
// backward: wq[ks][jt][lane][8]      lane (li, lr): channel 16 jt + li, tap f(ks, lr), j = class
__global__ void __launch_bounds__(256) tail_pack_kernel(const float* __restrict__ wb, const float* __restrict__ bf,
                                                        const float* __restrict__ b_out, uint16_t* __restrict__ wpf,
                                                        uint16_t* __restrict__ wq, int Ci) {
  const int KS = Ci / 32, NJ = Ci / 16;
  const int64_t n_f = (int64_t)kFSlots * KS * 64, n_b = (int64_t)kFSlots * 64, n_q = (int64_t)kBK * NJ * 64;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id < n_f + n_b) {
    const bool bias = id >= n_f;
    const int64_t e = bias ? id - n_f : id;
    const int lane = (int)(e & 63);
    const int ks = bias ? 0 : (int)((e >> 6) % KS);
    int slot = bias ? (int)(e >> 6) : (int)((e >> 6) / KS);
    const int li = lane & 15, lr = lane >> 4;
    const int og = slot & 1; slot >>= 1;
    const int rh = slot & 3; slot >>= 2;
    const int rd = slot & 3; slot >>= 2;
    const int dw = slot % 3, dh = (slot / 3) % 3, dd = slot / 9;
    const int rw = li & 3, o = 4 * og + (li >> 2);
    const int fd = rd - 4 * (dd - 1) + 1, fh = rh - 4 * (dh - 1) + 1, fw = rw - 4 * (dw - 1) + 1;
    const bool ok = fd >= 0 && fd < 6 && fh >= 0 && fh < 6 && fw >= 0 && fw < 6;
    const int64_t row = (((int64_t)fd * 6 + fh) * 6 + fw) * 8 + o;
    if (bias) {
      uint16_t* dst = wpf + n_f * 8 + e * 4;
      float v = (ok && lr == 0) ? bf[row] + ((dd == 1 && dh == 1 && dw == 1) ? b_out[o] : 0.f) : 0.f;
      const uint16_t hi = bf16_bits(v);
      const uint16_t lo = bf16_bits(v - bf16_val(hi));
      dst[0] = hi; dst[1] = lo; dst[2] = 0; dst[3] = 0;
    } else {
      uint16_t* dst = wpf + e * 8;
      const float* src = wb + row * Ci + 32 * ks + 8 * lr;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = ok ? bf16_bits(src[j]) : (uint16_t)0;
    }
    return;
  }
  const int64_t e = id - n_f - n_b;
  if (e == n_q) {                                                      // the zero block behind the forward pack
#pragma unroll
    for (int j = 0; j < 8; ++j) wpf[n_f * 8 + n_b * 4 + j] = 0;
    return;
  }
  if (e > n_q) return;
  const int lane = (int)(e & 63), li = lane & 15, lr = lane >> 4;
  const int jt = (int)((e >> 6) % NJ), ks = (int)((e >> 6) / NJ);
  const int fd = ks / 9, fh = 2 * ((ks / 3) % 3) + (lr >> 1), fw = 2 * (ks % 3) + (lr & 1);
  const int64_t row0 = (((int64_t)fd * 6 + fh) * 6 + fw) * 8;
  uint16_t* dst = wq + e * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = bf16_bits(wb[(row0 + j) * Ci + 16 * jt + li]);
}

// ---- forward
// Per class group (2 passes) the MFMA stream of a wave is 27 weight slabs x (Ci / 32 + 1) k-steps, each slab fragment (16 bytes per lane, from L2) feeding the
// 4 MFMAs of the wave's 4 coarse h rows.  The fragments go through a ring of 16 registers quads issued 16 units ahead (L2
// latency ~ 12 units of MFMA time); the x fragments (LDS) one k-step group ahead.  Everything is indexed at compile time.
constexpr int kRing = 16;
struct FwdUnit { int sd, sh, dw, ks, rdi, rhi, group, first; };
constexpr FwdUnit fwd_unit(int u, int KS1) {
  int g = 0;
  for (int sd = 0; sd < 2; ++sd)
    for (int sh = 0; sh < 2; ++sh) {
      const int nrh = sh ? 1 : 2, nt = (sd ? 1 : 2) * nrh;
      for (int dw = 0; dw < 3; ++dw)
        for (int ks = 0; ks < KS1; ++ks) {
          if (u < nt) return FwdUnit{sd, sh, dw, ks, u / nrh, u % nrh, g, u == 0};
          u -= nt;
          ++g;
        }
    }
  return FwdUnit{0, 0, 0, 0, 0, 0, 0, 0};
}
template <int CI>
struct FwdState {
  static constexpr int KS = CI / 32, KS1 = KS + 1, NU = 27 * KS1, NG = 12 * KS1, RS = CI + kXPad;
  f32x4 acc[2][2][4];                   // [rd idx][rh idx][coarse h row] of the class group in flight
  int og;
  bf16x8 ring[kRing];
  bf16x8 bx[2][4];
  int rd_of[2], rh_of[2], dd_out, dh_out, lane, li, lr;
  const uint16_t* Xs;
  const uint16_t* wl;                   // forward pack + 8 * lane
  const uint16_t* wbl;                  // bias slabs + 4 * lane
  const uint16_t* zero;                 // 16 zero bytes (lanes whose slab row is structurally zero all read these)
};
template <int CI, int U>
__device__ __forceinline__ void fwd_issue(FwdState<CI>& st) {          // weight fragment of unit U -> ring
  if constexpr (U < FwdState<CI>::NU) {
    constexpr FwdUnit un = fwd_unit(U, FwdState<CI>::KS1);
    constexpr int KS = FwdState<CI>::KS;
    const int dd = un.sd ? st.dd_out : 1, dh = un.sh ? st.dh_out : 1;
    const int slot = ((((dd * 3 + dh) * 3 + un.dw) * 4 + st.rd_of[un.rdi]) * 4 + st.rh_of[un.rhi]) * 2 + st.og;
    const bool a_on = un.dw == 1 || (st.li & 3) == (un.dw == 0 ? 0 : 3);
    if constexpr (un.ks < KS) {
      const uint16_t* p = st.wl + (int64_t)(slot * KS + un.ks) * 512;
      st.ring[U % kRing] = *reinterpret_cast<const bf16x8*>(a_on ? p : st.zero);
    } else {
      const uint16_t* p = st.wbl + (int64_t)slot * 256;
      const bf16x4 v = *reinterpret_cast<const bf16x4*>((a_on && st.lr == 0) ? p : st.zero);
      st.ring[U % kRing] = bf16x8{v[0], v[1], v[2], v[3], 0, 0, 0, 0};
    }
  }
}
template <int CI, int G>
__device__ __forceinline__ void fwd_issue_b(FwdState<CI>& st) {        // x fragments of k-step group G -> bx[G & 1]
  if constexpr (G < FwdState<CI>::NG) {
    constexpr int KS1 = FwdState<CI>::KS1, KS = FwdState<CI>::KS, RS = FwdState<CI>::RS;
    constexpr int ks = G % KS1, dw = (G / KS1) % 3, sh = (G / (3 * KS1)) % 2, sd = G / (6 * KS1);
    const int dd = sd ? st.dd_out : 1, dh = sh ? st.dh_out : 1;
    // x rows of coarse h row m shifted by d: LDS row ((dd * 6 + m + dh) * 18 + li + dw)
    const uint16_t* xrow = st.Xs + ((dd * 6 + dh) * 18 + st.li + dw) * RS;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if constexpr (ks < KS) st.bx[G & 1][m] = *reinterpret_cast<const bf16x8*>(xrow + m * 18 * RS + 32 * ks + 8 * st.lr);
      else {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(xrow + m * 18 * RS + CI + 4 * st.lr);
        st.bx[G & 1][m] = bf16x8{v[0], v[1], v[2], v[3], 0, 0, 0, 0};
      }
    }
  }
}
template <int CI, int U>
__device__ __forceinline__ void fwd_prologue(FwdState<CI>& st) {
  if constexpr (U < kRing) {
    fwd_issue<CI, U>(st);
    fwd_prologue<CI, U + 1>(st);
  }
}
template <int CI, int U>
__device__ __forceinline__ void fwd_step(FwdState<CI>& st) {
  if constexpr (U < FwdState<CI>::NU) {
    constexpr FwdUnit un = fwd_unit(U, FwdState<CI>::KS1);
    if constexpr (un.first) fwd_issue_b<CI, un.group + 1>(st);
    const bf16x8 a = st.ring[U % kRing];
    fwd_issue<CI, U + kRing>(st);
    __builtin_amdgcn_sched_barrier(0);                                  // (the scheduler otherwise hoists every load of the pass: 500+ spills)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x4& c = st.acc[un.rdi][un.rhi][m];
      const bf16x8 bv = st.bx[un.group & 1][m];
      if constexpr (un.ks < FwdState<CI>::KS) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bv, c, 0, 0, 0);
      else c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bf16x4{a[0], a[1], a[2], a[3]}, bf16x4{bv[0], bv[1], bv[2], bv[3]}, c, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    fwd_step<CI, U + 1>(st);
  }
}

// LOSS: 0 = logits only; 1 = + partial loss sums against one-hot float planes [B, 8, Df, Hf, Wf]; 2 = against the uint8 class map
// [B, Df, Hf, Wf].  part: [workgroups][8 classes][4] floats {sum p t, sum p^2, sum t^2, sum bce} (summed by dice_bce_finish_kernel).
// LOSS == 3: SLIDING-WINDOW form (utils.py:226-234) -- batch entry b is WINDOW b of a volume: its logits are ADDED into the fp32
// volume accumulator y [VB, 8, VD, VH, VW] at the window's origin target[4 b ..] = int32 {volume sample, z0, y0, x0} (device
// memory: a captured predictor graph is replayed with new coordinates) and part = the visit counts [VB, VD, VH, VW] are bumped
// (windows of one batch overlap: fp32 atomics, as micf_head_tail_col2im_sw).
template <int CI, int LOSS>
__global__ void __launch_bounds__(256, (CI <= 96 ? 2 : 1)) tail_fwd_fused_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wpf,
                                                             float* __restrict__ y, int B, int Dc, int Hc, int Wc,
                                                             const void* __restrict__ target, float* __restrict__ part,
                                                             int VD, int VH, int VW) {
  constexpr int KS = CI / 32, RS = CI + kXPad, V4 = CI / 4, ROWS = 3 * 6 * 18;
  extern __shared__ __attribute__((aligned(16))) uint16_t Xs[];       // [3][6][18][RS]
  __shared__ float loss_red[(LOSS == 1 || LOSS == 2) ? 4 : 1][8][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lr = lane >> 4;
  // XCD-aware order: workgroups go to the 8 XCDs round-robin by id; each XCD gets a CONTIGUOUS eighth of the tiles, so the halo
  // rows neighbouring tiles share are fetched into that XCD's L2 once (round-robin order: every XCD fetched its own copy, 3-5x the bytes)
  int r = blockIdx.x;
  if (gridDim.x % 8 == 0) r = (r % 8) * (gridDim.x / 8) + r / 8;
  const int qw0 = (r % (Wc / 16)) * 16; r /= (Wc / 16);
  const int qh0 = (r % (Hc / 4)) * 4; r /= (Hc / 4);
  const int qd = r % Dc;
  const int b = r / Dc;

  // ---- wave = quadrant of (rd, rh): index 0 = the border residue (a neighbour along that axis contributes), 1 = the inner one
  FwdState<CI> st;
  st.lane = lane; st.li = li; st.lr = lr;
  const int qa = wave >> 1, qc = wave & 1;
  st.rd_of[0] = qa ? 3 : 0; st.rd_of[1] = qa ? 2 : 1; st.rh_of[0] = qc ? 3 : 0; st.rh_of[1] = qc ? 2 : 1;
  st.dd_out = qa ? 2 : 0; st.dh_out = qc ? 2 : 0;                       // (d + 1 of the neighbour)
  st.Xs = Xs; st.wl = wpf + lane * 8; st.wbl = wpf + (int64_t)kFSlots * KS * 512 + lane * 4;
  st.zero = wpf + (int64_t)kFSlots * KS * 512 + (int64_t)kFSlots * 256;   // 16 zero bytes behind the packs
  st.og = 0;
  fwd_prologue<CI, 0>(st);                                              // (weights only: in flight while the x halo is staged)
  __builtin_amdgcn_sched_barrier(0);
  // ---- the halo of x rows -> LDS (bf16), 16 float4 loads in flight per thread
  for (int base = 0; base < ROWS * V4; base += 256 * 16) {
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = base + u * 256 + tid;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < ROWS * V4) {
        const int hv = idx / V4, g = idx % V4;
        const int zd = qd + hv / 108 - 1, zh = qh0 + (hv / 18) % 6 - 1, zw = qw0 + hv % 18 - 1;
        if ((unsigned)zd < (unsigned)Dc && (unsigned)zh < (unsigned)Hc && (unsigned)zw < (unsigned)Wc)
          v[u] = *reinterpret_cast<const float4*>(x + ((((int64_t)b * Dc + zd) * Hc + zh) * Wc + zw) * CI + 4 * g);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = base + u * 256 + tid;
      if (idx < ROWS * V4)
        *reinterpret_cast<u32x2v*>(&Xs[(idx / V4) * RS + 4 * (idx % V4)]) = u32x2v{pack_bf16(v[u].x, v[u].y), pack_bf16(v[u].z, v[u].w)};
    }
  }
  for (int hv = tid; hv < ROWS; hv += 256) {                           // the bias columns
    const int zd = qd + hv / 108 - 1, zh = qh0 + (hv / 18) % 6 - 1, zw = qw0 + hv % 18 - 1;
    const bool in = (unsigned)zd < (unsigned)Dc && (unsigned)zh < (unsigned)Hc && (unsigned)zw < (unsigned)Wc;
    u32x4v* p = reinterpret_cast<u32x4v*>(&Xs[hv * RS + CI]);
    p[0] = u32x4v{in ? 0x3F803F80u : 0u, 0u, 0u, 0u};
    p[1] = u32x4v{0u, 0u, 0u, 0u};
    p[2] = u32x4v{0u, 0u, 0u, 0u};
  }
  __syncthreads();

  const int Df = 4 * Dc, Hf = 4 * Hc, Wf = 4 * Wc;
#pragma nounroll
  for (int og = 0; og < 2; ++og) {
    if (og) {
      st.og = og;
      fwd_prologue<CI, 0>(st);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) st.acc[i][j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    fwd_issue_b<CI, 0>(st);
    fwd_step<CI, 0>(st);
    // ---- logits: lane (li, lr) holds, per tile, the 4 fine voxels w = 4 (qw0 + li) .. + 3 of class 4 og + lr
    const int o = 4 * og + lr;
    LossAcc la{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rdi = 0; rdi < 2; ++rdi) {
      // (loss: the 8 target quads of this rd are requested together, then consumed store by store)
      float4 tq[LOSS == 1 ? 2 : 1][LOSS == 1 ? 4 : 1];
      unsigned lq[LOSS == 2 ? 2 : 1][LOSS == 2 ? 4 : 1];          // (class map: the 4 label bytes, expanded where they are used)
      if constexpr (LOSS == 1 || LOSS == 2) {
#pragma unroll
        for (int rhi = 0; rhi < 2; ++rhi)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int ud = 4 * qd + st.rd_of[rdi], uh = 4 * (qh0 + m) + st.rh_of[rhi];
            if constexpr (LOSS == 1) {
              tq[rhi][m] = *reinterpret_cast<const float4*>(static_cast<const float*>(target) +
                                                            ((((int64_t)b * 8 + o) * Df + ud) * Hf + uh) * Wf + 4 * (qw0 + li));
            } else {
              lq[rhi][m] = *reinterpret_cast<const unsigned*>(static_cast<const uint8_t*>(target) +
                                                              (((int64_t)b * Df + ud) * Hf + uh) * Wf + 4 * (qw0 + li));
            }
          }
      }
#pragma unroll
      for (int rhi = 0; rhi < 2; ++rhi)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int ud = 4 * qd + st.rd_of[rdi], uh = 4 * (qh0 + m) + st.rh_of[rhi];
          const f32x4 v = st.acc[rdi][rhi][m];
          if constexpr (LOSS == 3) {
            // The 16 lanes of a class hold 4 consecutive voxels each = one 64-voxel row of the window.  Lane L takes voxel L of the
            // row, class after class, so that every atomic instruction covers 256 consecutive bytes (a lane adding its own four
            // values touched a quarter of every 16 bytes, four instructions per line: 1.67 ms per 7 windows instead of 0.5 + 0.5 of
            // the patch-matrix form).  x0 is any voxel: scalar atomics.
            const int32_t* sw = static_cast<const int32_t*>(target) + 4 * b;        // (workgroup-uniform: scalar loads)
            const int64_t row = ((int64_t)(sw[1] + ud) * VH + (sw[2] + uh)) * VW + sw[3] + 4 * qw0 + lane;
            const int64_t vplane = (int64_t)VD * VH * VW;
            const int kk = lane & 3;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int src = (lane >> 2) + 16 * c;
              const float a0 = __shfl(v[0], src, 64), a1 = __shfl(v[1], src, 64), a2 = __shfl(v[2], src, 64), a3 = __shfl(v[3], src, 64);
              atomicAdd(y + ((int64_t)sw[0] * 8 + 4 * og + c) * vplane + row, kk == 0 ? a0 : (kk == 1 ? a1 : (kk == 2 ? a2 : a3)));
            }
            if (og == 0) atomicAdd(part + (int64_t)sw[0] * vplane + row, 1.f);
          } else {
            float* dst = y + ((((int64_t)b * 8 + o) * Df + ud) * Hf + uh) * Wf + 4 * (qw0 + li);
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          }
          if constexpr (LOSS == 1) {
            const float4 t = tq[rhi][m];
            la.term(v[0], t.x); la.term(v[1], t.y); la.term(v[2], t.z); la.term(v[3], t.w);
          } else if constexpr (LOSS == 2) {
            const unsigned l4 = lq[rhi][m], oo = (unsigned)o;
            la.term(v[0], (l4 & 255u) == oo ? 1.f : 0.f); la.term(v[1], ((l4 >> 8) & 255u) == oo ? 1.f : 0.f);
            la.term(v[2], ((l4 >> 16) & 255u) == oo ? 1.f : 0.f); la.term(v[3], (l4 >> 24) == oo ? 1.f : 0.f);
          }
        }
    }
    if constexpr (LOSS == 1 || LOSS == 2) {
      // the 16 lanes of a row group hold the same class: fold them, then lane li == 0 of each group leaves the wave's share in LDS
#pragma unroll
      for (int dlt = 1; dlt < 16; dlt <<= 1) {
        la.a += __shfl_xor(la.a, dlt, 64); la.b += __shfl_xor(la.b, dlt, 64);
        la.c += __shfl_xor(la.c, dlt, 64); la.d += __shfl_xor(la.d, dlt, 64);
      }
      if (li == 0) {
        float* r = &loss_red[wave][o][0];
        r[0] = la.a; r[1] = la.b; r[2] = la.c; r[3] = la.d;
      }
    }
  }
  if constexpr (LOSS == 1 || LOSS == 2) {
    __syncthreads();
    if (tid < 32) part[(int64_t)blockIdx.x * 32 + tid] = (loss_red[0][tid >> 2][tid & 3] + loss_red[1][tid >> 2][tid & 3]) +
                                                        (loss_red[2][tid >> 2][tid & 3] + loss_red[3][tid >> 2][tid & 3]);
  }
}

// sums[class][4] (double) = sum over the workgroups' partial rows; loss = (0.7 sum_c dice_c + 0.3 sum_c bce_c / count) / K.
// One workgroup of 1024 threads: 32 threads per (class, quantity), four independent double accumulators each (2048 partial rows at
// base / 128^3: 16 loads per accumulator instead of a 256-deep dependent chain per thread -- 35 us on the critical chain before).
__global__ void __launch_bounds__(1024) dice_bce_finish_kernel(const float* __restrict__ part, int nparts, double* __restrict__ sums,
                                                               float* __restrict__ loss, double count) {
  __shared__ double acc[32];
  const int e = threadIdx.x >> 5, sub = threadIdx.x & 31;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = sub;
  for (; i + 96 < nparts; i += 128) {
    s0 += (double)part[(int64_t)i * 32 + e]; s1 += (double)part[(int64_t)(i + 32) * 32 + e];
    s2 += (double)part[(int64_t)(i + 64) * 32 + e]; s3 += (double)part[(int64_t)(i + 96) * 32 + e];
  }
  for (; i < nparts; i += 32) s0 += (double)part[(int64_t)i * 32 + e];
  double s = (s0 + s1) + (s2 + s3);
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64);
  if (sub == 0) { acc[e] = s; sums[e] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double dice = 0.0, ce = 0.0;
    for (int i = 0; i < 8; ++i) {
      dice += 1.0 - (2.0 * acc[i * 4 + 0] + 1.0) / (acc[i * 4 + 1] + acc[i * 4 + 2] + 1.0);
      ce += acc[i * 4 + 3] / count;
    }
    *loss = (float)((0.7 * dice + 0.3 * ce) / 8.0);
  }
}

// ---- backward data
// A wave's MFMA stream: 54 k-steps (4 taps x 8 classes each), per k-step its CI/96 weight fragments (L2) and 2 dy fragments (LDS).
// Same ring discipline as the forward: weight fragments issued kBRing k-steps ahead (the first ones before the dy region is staged).
template <int CI>
struct BwdState {
  static constexpr int NJ = CI / 16, JPW = NJ / 6, R = JPW == 1 ? 12 : 6;
  f32x4 acc[JPW][2];
  bf16x8 ring[R][JPW];
  bf16x8 bx[2][2];
  const u32x4v* dlane;
  const uint16_t* wlane;
};
template <int CI, int U>
__device__ __forceinline__ void bwd_issue(BwdState<CI>& st) {
  if constexpr (U < kBK) {
#pragma unroll
    for (int j = 0; j < BwdState<CI>::JPW; ++j)
      st.ring[U % BwdState<CI>::R][j] = *reinterpret_cast<const bf16x8*>(st.wlane + ((int64_t)U * BwdState<CI>::NJ + j) * 512);
  }
}
template <int CI, int U>
__device__ __forceinline__ void bwd_issue_b(BwdState<CI>& st) {
  if constexpr (U < kBK) {
    constexpr int HR = 10;
    constexpr int fd = U / 9, fh2 = (U / 3) % 3, fw2 = U % 3;
    constexpr int off = (fd * HR + 2 * fh2) * kDW + 2 * fw2 + (fw2 == 2 ? 1 : 0);
    st.bx[U & 1][0] = __builtin_bit_cast(bf16x8, st.dlane[off]);
    st.bx[U & 1][1] = __builtin_bit_cast(bf16x8, st.dlane[off + 4 * kDW]);
  }
}
template <int CI, int U>
__device__ __forceinline__ void bwd_prologue(BwdState<CI>& st) {
  if constexpr (U < BwdState<CI>::R) {
    bwd_issue<CI, U>(st);
    bwd_prologue<CI, U + 1>(st);
  }
}
template <int CI, int U>
__device__ __forceinline__ void bwd_step(BwdState<CI>& st) {
  if constexpr (U < kBK) {
    constexpr int JPW = BwdState<CI>::JPW;
    bwd_issue_b<CI, U + 1>(st);
    bf16x8 a[JPW];
#pragma unroll
    for (int j = 0; j < JPW; ++j) a[j] = st.ring[U % BwdState<CI>::R][j];
    bwd_issue<CI, U + BwdState<CI>::R>(st);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < JPW; ++j) {
      st.acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], st.bx[U & 1][0], st.acc[j][0], 0, 0, 0);
      st.acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], st.bx[U & 1][1], st.acc[j][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    bwd_step<CI, U + 1>(st);
  }
}

template <int CI>
__global__ void __launch_bounds__(384) tail_bwd_data_fused_kernel(const float* __restrict__ dy, const uint16_t* __restrict__ wq,
                                                                  float* __restrict__ dx, int B, int Dc, int Hc, int Wc) {
  constexpr int NJ = CI / 16, JPW = NJ / 6, HR = 10;
  extern __shared__ __attribute__((aligned(16))) u32x4v Ds[];          // [6][HR][kDW] fine voxels x 8 classes (bf16)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lr = lane >> 4;
  // XCD-aware order: workgroups go to the 8 XCDs round-robin by id; each XCD gets a CONTIGUOUS eighth of the tiles, so the halo
  // rows neighbouring tiles share are fetched into that XCD's L2 once (round-robin order: every XCD fetched its own copy, 3-5x the bytes)
  int r = blockIdx.x;
  if (gridDim.x % 8 == 0) r = (r % 8) * (gridDim.x / 8) + r / 8;
  const int qw0 = (r % (Wc / 16)) * 16; r /= (Wc / 16);
  const int qh0 = (r % (Hc / 2)) * 2; r /= (Hc / 2);
  const int qd = r % Dc;
  const int b = r / Dc;
  const int Df = 4 * Dc, Hf = 4 * Hc, Wf = 4 * Wc;
  const int64_t plane = (int64_t)Df * Hf * Wf;
  const float* src = dy + (int64_t)b * 8 * plane;

  BwdState<CI> st;
  st.wlane = wq + ((int64_t)wave * JPW * 64 + lane) * 8;
  bwd_prologue<CI, 0>(st);                                             // (weights only: in flight while the dy region is staged)
  __builtin_amdgcn_sched_barrier(0);

  // ---- the fine region of dy -> LDS.  Item = (class pair, fine (d, h) row, aligned 4-voxel chunk along w): two 16-byte loads, four
  // packed class pairs to LDS; 6 items (12 loads) in flight per thread.  Chunk c covers region voxels w = 4 c - 3 .. 4 c (region
  // w = fine w - (4 qw0 - 1)), so chunks 0 and 17 contribute one voxel each.
  constexpr int NI = 4 * 6 * HR * 18, NBI = 6;
  for (int base = 0; base < NI; base += 384 * NBI) {
    float4 v[NBI][2];
#pragma unroll
    for (int u = 0; u < NBI; ++u) {
      const int p = base + u * 384 + tid;
      const int c = p % 18, h = (p / 18) % HR, d = (p / (18 * HR)) % 6, op = p / (18 * HR * 6);
      const int ud = 4 * qd - 1 + d, uh = 4 * qh0 - 1 + h, uw = 4 * qw0 - 4 + 4 * c;
      const bool in = p < NI && (unsigned)ud < (unsigned)Df && (unsigned)uh < (unsigned)Hf && (unsigned)uw < (unsigned)Wf;
      const float* s = src + (int64_t)(2 * op) * plane + ((int64_t)ud * Hf + uh) * Wf + uw;
      v[u][0] = in ? *reinterpret_cast<const float4*>(s) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[u][1] = in ? *reinterpret_cast<const float4*>(s + plane) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NBI; ++u) {
      const int p = base + u * 384 + tid;
      if (p < NI) {
        const int c = p % 18, h = (p / 18) % HR, d = (p / (18 * HR)) % 6, op = p / (18 * HR * 6);
        unsigned* row = reinterpret_cast<unsigned*>(Ds + (d * HR + h) * kDW) + op;      // (4 dwords per voxel slot: class pair op)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int w = 4 * c - 3 + e;
          if (w >= 0 && w < 66) row[(w + (w >> 2)) * 4] = pack_bf16(f4e(v[u][0], e), f4e(v[u][1], e));
        }
      }
    }
  }
  __syncthreads();

#pragma unroll
  for (int j = 0; j < JPW; ++j) st.acc[j][0] = st.acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // coarse voxel (m, li), tap (fd, fh, fw): fine (fd, 4 m + fh, 4 li + fw) -> slot (fd * HR + 4 m + fh) * kDW + 5 li + fw + (fw >> 2)
  st.dlane = Ds + (lr >> 1) * kDW + (lr & 1) + 5 * li;
  bwd_issue_b<CI, 0>(st);
  bwd_step<CI, 0>(st);
  auto& acc = st.acc;
#pragma unroll
  for (int j = 0; j < JPW; ++j)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int64_t q = (((int64_t)b * Dc + qd) * Hc + qh0 + m) * Wc + qw0 + li;
      const f32x4 v = acc[j][m];
      *reinterpret_cast<float4*>(dx + q * CI + 16 * (wave * JPW + j) + 4 * lr) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- backward weights: dWb[(f, o), k] = sum_q dy[o, 4 q - 1 + f] x[q, k],  dBf[(f, o)] = sum_q dy[o, 4 q - 1 + f]
// The reduction runs over coarse voxels, 32 (one stretch of a coarse w row) per MFMA k-step.  A workgroup owns one fd (6 workgroup
// kinds) and a range of coarse (b, d, h, w-stretch) steps; its 6 waves own one fh each: 6 fw x 8 classes = 48 rows of Wb = 3 row
// tiles x Ci / 16 column tiles of accumulators per wave.  Per step a
// wave de-interleaves ITS fine row of dy (d = 4 qd - 1 + fd, h = 4 qh - 1 + fh, 130 voxels x 8 classes) into LDS as
// S[class][fw][q] = dy[4 q - 1 + fw], so an operand fragment (8 consecutive q of one (fw, class)) is one 16-byte read; the x rows of
// the step are shared by the 6 waves (bf16, row-major) and read transposed with ds_read_b64_tr_b16.  Loads of step s + 1 are in
// flight under the MFMAs of step s.  dBf is a plain fp32 sum of the values a lane de-interleaves (its fw residue is fixed: lane & 3),
// exact like the bias gradient of the patch-matrix path.  Partial slabs go to a workspace; tail_wgrad_reduce_kernel sums them
// into dWb / dBf (=).
constexpr int kWS = 40;                 // bf16 elements per S row (32 + 8: 80 bytes -- the 16 rows of a fragment read hit disjoint bank quads)
typedef short s16x4w __attribute__((ext_vector_type(4)));

template <int CI>
__global__ void __launch_bounds__(384) tail_bwd_weight_fused_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                    float* __restrict__ ws, int B, int Dc, int Hc, int Wc,
                                                                    int steps_per_group) {
  constexpr int NJ = CI / 16, XS = CI + 20, V4 = CI / 4, XL = (32 * V4) / 384;
  static_assert((32 * V4) % 384 == 0, "x stretch divides over the workgroup");
  extern __shared__ __attribute__((aligned(16))) uint16_t wsm[];          // Xs[2][32 * XS] | Ss[2][6][48 * kWS]
  uint16_t (*Xs)[32 * XS] = reinterpret_cast<uint16_t (*)[32 * XS]>(wsm);
  uint16_t (*Ss)[6][48 * kWS] = reinterpret_cast<uint16_t (*)[6][48 * kWS]>(wsm + 2 * 32 * XS);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lr = lane >> 4;
  const int fd = blockIdx.x, grp = blockIdx.y;
  const int Df = 4 * Dc, Hf = 4 * Hc, Wf = 4 * Wc, wst = Wc / 32;
  const int64_t plane = (int64_t)Df * Hf * Wf;
  const int total = B * Dc * Hc * wst;
  const int s_begin = grp * steps_per_group, s_end = min(total, s_begin + steps_per_group);

  f32x4 acc[3][NJ];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs_main[8], bs_alt[8];                                         // dBf partials: fw = lane & 3, and fw + 4 (lanes with fw < 2)
#pragma unroll
  for (int o = 0; o < 8; ++o) bs_main[o] = bs_alt[o] = 0.f;

  float dyr[8][3];
  float4 xr[XL];
  auto fetch = [&](int s) {
    int r = s;
    const int qw0 = (r % wst) * 32; r /= wst;
    const int qh = r % Hc; r /= Hc;
    const int qd = r % Dc;
    const int b = r / Dc;
    const int ud = 4 * qd - 1 + fd, uh = 4 * qh - 1 + wave;
    const bool row_in = (unsigned)ud < (unsigned)Df && (unsigned)uh < (unsigned)Hf;
    const float* src = dy + (int64_t)b * 8 * plane + ((int64_t)ud * Hf + uh) * Wf;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = lane + 64 * j, uw = 4 * qw0 - 1 + t;
      const bool in = row_in && t < 130 && (unsigned)uw < (unsigned)Wf;
#pragma unroll
      for (int o = 0; o < 8; ++o) dyr[o][j] = in ? src[o * plane + uw] : 0.f;
    }
    const float* xs = x + ((((int64_t)b * Dc + qd) * Hc + qh) * Wc + qw0) * CI;
#pragma unroll
    for (int u = 0; u < XL; ++u) xr[u] = *reinterpret_cast<const float4*>(xs + 4 * (tid + 384 * u));
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < XL; ++u) {
      const int idx = tid + 384 * u;
      *reinterpret_cast<u32x2v*>(&Xs[buf][(idx / V4) * XS + 4 * (idx % V4)]) = u32x2v{pack_bf16(xr[u].x, xr[u].y), pack_bf16(xr[u].z, xr[u].w)};
    }
    uint16_t* S = Ss[buf][wave];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = lane + 64 * j, q = t >> 2, fwl = t & 3;            // fine voxel t of the row = 4 q + fwl
      if (t < 130) {
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          const uint16_t v = bf16_bits(dyr[o][j]);
          if (q < 32) { S[(o * 6 + fwl) * kWS + q] = v; bs_main[o] += dyr[o][j]; }
          if (fwl < 2 && q >= 1) { S[(o * 6 + fwl + 4) * kWS + q - 1] = v; bs_alt[o] += dyr[o][j]; }
        }
      }
    }
  };

  if (s_begin < s_end) fetch(s_begin);
  int buf = 0;
  for (int s = s_begin; s < s_end; ++s, buf ^= 1) {
    commit(buf);
    __syncthreads();
    if (s + 1 < s_end) fetch(s + 1);
    const uint16_t* S = Ss[buf][wave];
    bf16x8 a[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
      a[t] = *reinterpret_cast<const bf16x8*>(S + ((li & 7) * 6 + 2 * t + (li >> 3)) * kWS + 8 * lr);
    // transposed x fragments: lanes 4 j .. 4 j + 3 of a 16-lane group address row j (8 bytes each), lane i receives column i
    const unsigned xb = (unsigned)(uintptr_t)(&Xs[buf][0]) + (unsigned)(((8 * lr + (li >> 2)) * XS + 4 * (li & 3)) * 2);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(uintptr_t)(xb + 32 * j));
      const s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(uintptr_t)(xb + 32 * j + 4 * XS * 2));
      const bf16x8 bb = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], bb, acc[t][j], 0, 0, 0);
    }
  }
  float* out = ws + ((((int64_t)grp * 6 + fd) * 6 + wave) * (3 * NJ) * 64 + lane) * 4;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      *reinterpret_cast<float4*>(out + (int64_t)(t * NJ + j) * 256) = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
  // dBf partials: sum over the lanes of one residue (lane & 3), then [group][slab][fw][class] behind the slabs
  float* bout = ws + (int64_t)gridDim.y * 36 * (3 * NJ) * 256 + (((int64_t)grp * 6 + fd) * 6 + wave) * 48;
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float m = bs_main[o], al = bs_alt[o];
#pragma unroll
    for (int d = 4; d < 64; d <<= 1) { m += __shfl_xor(m, d, 64); al += __shfl_xor(al, d, 64); }
    if (lane < 4) bout[lane * 8 + o] = m;
    if (lane < 2) bout[(lane + 4) * 8 + o] = al;
  }
}

// dWb[(fd, fh, fw, o), ch] = sum over the groups' partial slabs (one thread per (slab tile, lane)); dBf likewise from its partials.
__global__ void __launch_bounds__(256) tail_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dwb,
                                                                float* __restrict__ dbf, int Ci, int groups) {
  const int NJ = Ci / 16;
  const int per_slab = 3 * NJ * 64;
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= 36 * per_slab) {                                           // dBf: [group][slab][fw][class] behind the slabs
    const int e = id - 36 * per_slab;
    if (e >= 36 * 48) return;
    const float* b = ws + (int64_t)groups * 36 * per_slab * 4;
    float sum = 0.f;
    for (int g = 0; g < groups; ++g) sum += b[(int64_t)g * 36 * 48 + e];
    dbf[e] = sum;                                                      // (slab * 6 + fw) * 8 + o == e
    return;
  }
  const int slab = id / per_slab, e = id % per_slab;
  const int lane = e & 63, tile = e >> 6, t = tile / NJ, j = tile % NJ, li = lane & 15, lr = lane >> 4;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = 0; g < groups; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(ws + (((int64_t)g * 36 + slab) * per_slab + e) * 4);
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  const float vals[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = 4 * lr + i, fw = 2 * t + (ra >> 3), o = ra & 7;
    const int row = (slab * 6 + fw) * 8 + o;                             // slab = fd * 6 + fh
    dwb[(int64_t)row * Ci + 16 * j + li] = vals[i];
  }
}

static bool fused_ok(int Dc, int Hc, int Wc, int Ci, int Co, int P) {
  return Co == 8 && P == 4 && (Ci == 96 || Ci == 192) && Dc > 0 && Hc > 0 && Wc > 0 && Wc % 16 == 0 && Hc % 4 == 0;
}

template <typename K>
static void allow_lds(K kernel, int bytes) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }

}  // namespace micf

using namespace micf;

extern "C" int micf_head_tail_fused_supported(int Dc, int Hc, int Wc, int Ci, int Co, int P, int dtype) {
  return (dtype == MICF_DTYPE_BF16 && fused_ok(Dc, Hc, Wc, Ci, Co, P)) ? 1 : 0;
}

// bytes of the two weight packs (forward incl. its bias slabs; backward data)
extern "C" int64_t micf_head_tail_pack_bytes(int Ci, int which) {
  if (Ci <= 0 || Ci % 32) return 0;
  if (which == 0) return ((int64_t)kFSlots * (Ci / 32) * 512 + (int64_t)kFSlots * 256 + 8) * 2;
  return (int64_t)kBK * (Ci / 16) * 512 * 2;
}

extern "C" int micf_head_tail_pack(const float* wb, const float* bf, const float* b_out, void* pack_fwd, void* pack_bwd, int Ci,
                                   int Co, int P, micf_stream_t stream) {
  if (!wb || !bf || !b_out || !pack_fwd || !pack_bwd) return MICF_EINVAL;
  if (Co != 8 || P != 4 || Ci <= 0 || Ci % 32) return MICF_EUNSUPPORTED;
  const int64_t n = (int64_t)kFSlots * (Ci / 32) * 64 + (int64_t)kFSlots * 64 + (int64_t)kBK * (Ci / 16) * 64 + 1;
  hipLaunchKernelGGL(tail_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wb, bf, b_out,
                     reinterpret_cast<uint16_t*>(pack_fwd), reinterpret_cast<uint16_t*>(pack_bwd), Ci);
  MICF_RETURN_LAUNCH();
}

template <int CI, int LOSS>
static void launch_tail_fwd(dim3 grid, hipStream_t s, const float* x, const uint16_t* wp, float* y, int B, int Dc, int Hc, int Wc,
                            const void* target, float* part, int VD = 0, int VH = 0, int VW = 0) {
  static std::once_flag once;
  std::call_once(once, [] { allow_lds(&tail_fwd_fused_kernel<CI, LOSS>, 324 * (CI + kXPad) * 2); });
  hipLaunchKernelGGL((tail_fwd_fused_kernel<CI, LOSS>), grid, dim3(256), 324 * (CI + kXPad) * 2, s, x, wp, y, B, Dc, Hc, Wc, target, part,
                     VD, VH, VW);
}

static int tail_fwd_any(const float* x, const void* pack_fwd, float* y, const void* target, int target_is_label, float* part, int B,
                        int Dc, int Hc, int Wc, int Ci, int Co, int P, hipStream_t s) {
  if (!x || !pack_fwd || !y || B <= 0) return MICF_EINVAL;
  if (!fused_ok(Dc, Hc, Wc, Ci, Co, P) || !aligned16(x) || !aligned16(y) || !aligned16(pack_fwd)) return MICF_EUNSUPPORTED;
  const dim3 grid((unsigned)((int64_t)B * Dc * (Hc / 4) * (Wc / 16)));
  const uint16_t* wp = reinterpret_cast<const uint16_t*>(pack_fwd);
  const int loss = target ? (target_is_label ? 2 : 1) : 0;
#define MICF_TF(CI_) do { if (loss == 0) launch_tail_fwd<CI_, 0>(grid, s, x, wp, y, B, Dc, Hc, Wc, nullptr, nullptr); \
                         else if (loss == 1) launch_tail_fwd<CI_, 1>(grid, s, x, wp, y, B, Dc, Hc, Wc, target, part); \
                         else launch_tail_fwd<CI_, 2>(grid, s, x, wp, y, B, Dc, Hc, Wc, target, part); } while (0)
  if (Ci == 96) MICF_TF(96); else MICF_TF(192);
#undef MICF_TF
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_head_tail_fwd_fused(const float* x, const void* pack_fwd, float* y, int B, int Dc, int Hc, int Wc, int Ci,
                                        int Co, int P, micf_stream_t stream) {
  return tail_fwd_any(x, pack_fwd, y, nullptr, 0, nullptr, B, Dc, Hc, Wc, Ci, Co, P, (hipStream_t)stream);
}

extern "C" int micf_head_tail_fwd_fused_sw(const float* x, const void* pack_fwd, float* out, float* count, const int32_t* coords,
                                           int n, int Dc, int Hc, int Wc, int Ci, int Co, int P, int VB, int VD, int VH, int VW,
                                           micf_stream_t stream) {
  if (!x || !pack_fwd || !out || !count || !coords || n <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0 || VB <= 0) return MICF_EINVAL;
  if (Dc * P > VD || Hc * P > VH || Wc * P > VW) return MICF_EINVAL;      // (the window must fit the volume; origins are the caller's)
  if (!fused_ok(Dc, Hc, Wc, Ci, Co, P) || !aligned16(x) || !aligned16(pack_fwd)) return MICF_EUNSUPPORTED;
  const dim3 grid((unsigned)((int64_t)n * Dc * (Hc / 4) * (Wc / 16)));
  const uint16_t* wp = reinterpret_cast<const uint16_t*>(pack_fwd);
  hipStream_t s = (hipStream_t)stream;
  if (Ci == 96) launch_tail_fwd<96, 3>(grid, s, x, wp, out, n, Dc, Hc, Wc, coords, count, VD, VH, VW);
  else launch_tail_fwd<192, 3>(grid, s, x, wp, out, n, Dc, Hc, Wc, coords, count, VD, VH, VW);
  MICF_RETURN_LAUNCH();
}

extern "C" int64_t micf_head_tail_loss_parts(int B, int Dc, int Hc, int Wc) {
  if (B <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0 || Hc % 4 || Wc % 16) return 0;
  return (int64_t)B * Dc * (Hc / 4) * (Wc / 16);
}

extern "C" int micf_head_tail_fwd_loss_fused(const float* x, const void* pack_fwd, float* y, const void* target, int target_is_label,
                                             float* part, double* sums, float* loss, int B, int Dc, int Hc, int Wc, int Ci, int Co,
                                             int P, micf_stream_t stream) {
  if (!target || !part || !sums || !loss) return MICF_EINVAL;
  if (!aligned16(target) || !aligned16(part)) return MICF_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int rc = tail_fwd_any(x, pack_fwd, y, target, target_is_label, part, B, Dc, Hc, Wc, Ci, Co, P, s);
  if (rc != MICF_OK) return rc;
  const int nparts = (int)micf_head_tail_loss_parts(B, Dc, Hc, Wc);
  const double count = (double)B * 64.0 * Dc * Hc * Wc;                 // elements of one class channel over the batch (P = 4)
  hipLaunchKernelGGL(dice_bce_finish_kernel, dim3(1), dim3(1024), 0, s, part, nparts, sums, loss, count);
  MICF_RETURN_LAUNCH();
}

extern "C" int micf_head_tail_bwd_data_fused(const float* dy, const void* pack_bwd, float* dx, int B, int Dc, int Hc, int Wc,
                                             int Ci, int Co, int P, micf_stream_t stream) {
  if (!dy || !pack_bwd || !dx || B <= 0) return MICF_EINVAL;
  if (!fused_ok(Dc, Hc, Wc, Ci, Co, P) || !aligned16(dx) || !aligned16(dy) || !aligned16(pack_bwd)) return MICF_EUNSUPPORTED;
  constexpr int lds = 6 * 10 * kDW * 16;
  static std::once_flag once;
  std::call_once(once, [] {
    allow_lds(&tail_bwd_data_fused_kernel<96>, lds);
    allow_lds(&tail_bwd_data_fused_kernel<192>, lds);
  });
  const dim3 grid((unsigned)((int64_t)B * Dc * (Hc / 2) * (Wc / 16)));
  const uint16_t* wq = reinterpret_cast<const uint16_t*>(pack_bwd);
  hipStream_t s = (hipStream_t)stream;
  if (Ci == 96) hipLaunchKernelGGL(tail_bwd_data_fused_kernel<96>, grid, dim3(384), lds, s, dy, wq, dx, B, Dc, Hc, Wc);
  else hipLaunchKernelGGL(tail_bwd_data_fused_kernel<192>, grid, dim3(384), lds, s, dy, wq, dx, B, Dc, Hc, Wc);
  MICF_RETURN_LAUNCH();
}

static int wgrad_groups(int total) {
  int g = 256 / 6;                                   // ~one workgroup per CU over the 6 fd kinds
  return g < total ? g : total;
}

extern "C" int64_t micf_head_tail_bwd_weight_fused_workspace(int B, int Dc, int Hc, int Wc, int Ci) {
  if (B <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0 || Wc % 32 || (Ci != 96 && Ci != 192)) return 0;
  const int total = B * Dc * Hc * (Wc / 32);
  return (int64_t)wgrad_groups(total) * 36 * (3 * (Ci / 16) * 256 + 48);
}

extern "C" int micf_head_tail_bwd_weight_fused(const float* dy, const float* x, float* dwb, float* dbf, float* workspace,
                                               int64_t workspace_floats, int B, int Dc, int Hc, int Wc, int Ci, int Co, int P,
                                               micf_stream_t stream) {
  if (!dy || !x || !dwb || !dbf || !workspace || B <= 0) return MICF_EINVAL;
  const int64_t need = micf_head_tail_bwd_weight_fused_workspace(B, Dc, Hc, Wc, Ci);
  if (!fused_ok(Dc, Hc, Wc, Ci, Co, P) || need == 0 || workspace_floats < need || !aligned16(x) || !aligned16(workspace))
    return MICF_EUNSUPPORTED;
  const int total = B * Dc * Hc * (Wc / 32);
  const int groups = wgrad_groups(total);
  const int spg = (total + groups - 1) / groups;
  hipStream_t s = (hipStream_t)stream;
  const int lds = (2 * 32 * (Ci + 20) + 2 * 6 * 48 * kWS) * 2;
  static std::once_flag once;
  std::call_once(once, [] {
    allow_lds(&tail_bwd_weight_fused_kernel<96>, (2 * 32 * (96 + 20) + 2 * 6 * 48 * kWS) * 2);
    allow_lds(&tail_bwd_weight_fused_kernel<192>, (2 * 32 * (192 + 20) + 2 * 6 * 48 * kWS) * 2);
  });
  if (Ci == 96) hipLaunchKernelGGL(tail_bwd_weight_fused_kernel<96>, dim3(6, groups), dim3(384), lds, s, dy, x, workspace, B, Dc, Hc, Wc, spg);
  else hipLaunchKernelGGL(tail_bwd_weight_fused_kernel<192>, dim3(6, groups), dim3(384), lds, s, dy, x, workspace, B, Dc, Hc, Wc, spg);
  if (hipGetLastError() != hipSuccess) return MICF_ELAUNCH;
  const int n = 36 * 3 * (Ci / 16) * 64 + 36 * 48;
  hipLaunchKernelGGL(tail_wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, workspace, dwb, dbf, Ci, groups);
  MICF_RETURN_LAUNCH();
}
