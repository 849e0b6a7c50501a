// offset_head.hip -- the offset head of a cross block as ONE C-ABI call for both modalities of a cross pair
// (MS.py:354-384: conv_offset = 3^3 conv on cat[LN1(x), raw xa] -> LayerNorm(16) -> GELU -> 1^3 conv -> deformable sampling of
// raw xa).  The two heads of a pair are independent and of the same shape: every kernel of the chain takes both pointer sets
// (blockIdx.z / .y), so the pair costs 2 launches forward (conv, head + sampler) and 3-4 backward (sampler adjoint [+ gather],
// finish, conv data gradient) -- instead of the same chains issued per modality on two streams, whose fork / join inside a
// captured graph costs ~16 us per pair at the 8^3 / 4^3 stages, more than the kernels themselves.
#include "common.h"

using namespace micf;

// 1 when micf_offset_head_fwd accumulates the conv output atomically for this shape, i.e. `hid` must be zero when the conv
// starts: the call clears it itself unless told (hid_zeroed) that the caller already did, off the critical path
extern "C" int micf_offset_head_needs_zero(int B, int D, int H, int W, int C) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || W < 4) return 0;
  return conv3_fwd_x_splits(B, D, H, W, C, C) ? 1 : 0;
}

extern "C" int micf_offset_head_fwd(const micf_offset_head_group* groups, int ngroups, int B, int D, int H, int W, int C, float eps,
                                    int prepared, int hid_zeroed, int dtype, micf_stream_t stream) {
  if (!groups || ngroups < 1 || ngroups > 2 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MICF_EINVAL;
  if (dtype != MICF_DTYPE_F32 && dtype != MICF_DTYPE_BF16) return MICF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  Conv3FwdSet cs[2];
  SampleFwdSet ss[2];
  bool conv_only = true;               // no flow / xs outputs: the caller samples inside micf_block_fwd
  for (int i = 0; i < ngroups; ++i) {
    const micf_offset_head_group& g = groups[i];
    if (!g.xn || !g.xa || !g.conv_w || !g.hid) return MICF_EINVAL;
    conv_only = conv_only && !g.flow && !g.xs;
    if (!conv_only && (!g.ln_g || !g.ln_b || !g.w1 || !g.flow || !g.xs)) return MICF_EINVAL;
    cs[i] = Conv3FwdSet{g.xn, g.xa, g.conv_w, g.conv_b, g.hid, g.conv_ws};
    ss[i] = SampleFwdSet{g.hid, g.ln_g, g.ln_b, g.w1, g.xa, g.flow, g.xs};
  }
  int rc = MICF_EUNSUPPORTED;
  if (groups[0].conv_ws && (ngroups == 1 || groups[1].conv_ws))
    rc = conv3_fwd_x_groups(cs, ngroups, C, C, B, D, H, W, kOffsetHidden, s, dtype, prepared, hid_zeroed);
  if (rc == MICF_EUNSUPPORTED) {           // shapes outside the direct kernel (W < 4): one generic call per head
    for (int i = 0; i < ngroups; ++i) {
      rc = micf_conv3_fwd(cs[i].x1, C, cs[i].x2, C, cs[i].w, cs[i].bias, cs[i].y, 0, B, D, H, W, kOffsetHidden, nullptr, 0, 0, dtype, stream);
      if (rc != MICF_OK) return rc;
    }
  }
  if (rc != MICF_OK || conv_only) return rc;
  return offset_sample_fwd_groups(ss, ngroups, B, D, H, W, C, eps, s);
}

extern "C" int64_t micf_offset_head_bwd_workspace(int ngroups, int B, int D, int H, int W) {
  return ngroups > 0 ? ngroups * micf_offset_sample_bwd_workspace(B, D, H, W) : 0;
}

static int head_bwd_sets(const micf_offset_head_bwd_group* groups, int ngroups, SampleBwdSet* ss) {
  for (int i = 0; i < ngroups; ++i) {
    const micf_offset_head_bwd_group& g = groups[i];
    if (!g.dxs || !g.hid || !g.flow || !g.xa || !g.ln_g || !g.ln_b || !g.w1 || !g.conv_w || !g.dxa || !g.dxn || !g.dhid || !g.dln_g ||
        !g.dln_b || !g.dw1)
      return MICF_EINVAL;
    ss[i] = SampleBwdSet{g.dxs, g.hid, g.ln_g, g.ln_b, g.w1, g.xa, g.flow, g.dxa, g.dhid, g.dln_g, g.dln_b, g.dw1,
                         CellLists{nullptr, nullptr, nullptr, nullptr, 0, nullptr}, nullptr};
  }
  return MICF_OK;
}

extern "C" int micf_offset_head_bwd_finish(const micf_offset_head_bwd_group* groups, int ngroups, int B, int D, int H, int W, int C,
                                           float* workspace, int64_t workspace_floats, micf_stream_t stream) {
  if (!groups || ngroups < 1 || ngroups > 2 || !micf_offset_head_finish_deferrable(B, D, H, W)) return MICF_EINVAL;
  SampleBwdSet ss[2];
  const int rc = head_bwd_sets(groups, ngroups, ss);
  if (rc != MICF_OK) return rc;
  return offset_sample_bwd_groups(ss, ngroups, B, D, H, W, C, 0.f, workspace, workspace_floats, (hipStream_t)stream, 2);
}

// The finishing launches of several deferring micf_offset_head_bwd calls (any mix of grids) as ONE launch.
extern "C" int micf_offset_head_bwd_finish_grouped(const micf_offset_head_finish_call* calls, int ncalls, micf_stream_t stream) {
  constexpr int kMax = 16;
  if (!calls || ncalls < 1 || ncalls > kMax) return MICF_EINVAL;
  SampleBwdSet ss[kMax][2];
  SampleFinishCall fc[kMax];
  for (int c = 0; c < ncalls; ++c) {
    const micf_offset_head_finish_call& q = calls[c];
    if (!q.groups || q.ngroups < 1 || q.ngroups > 2 || !micf_offset_head_finish_deferrable(q.B, q.D, q.H, q.W)) return MICF_EINVAL;
    const int rc = head_bwd_sets(q.groups, q.ngroups, ss[c]);
    if (rc != MICF_OK) return rc;
    fc[c] = SampleFinishCall{ss[c], q.ngroups, q.B, q.D, q.H, q.W, q.C, q.workspace, q.workspace_floats};
  }
  return offset_sample_finish_many(fc, ncalls, (hipStream_t)stream);
}

extern "C" int micf_offset_head_bwd(const micf_offset_head_bwd_group* groups, int ngroups, int B, int D, int H, int W, int C,
                                    float eps, int prepared, float* workspace, int64_t workspace_floats, int dtype,
                                    int defer_finish, micf_stream_t stream) {
  if (!groups || ngroups < 1 || ngroups > 2 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MICF_EINVAL;
  if (dtype != MICF_DTYPE_F32 && dtype != MICF_DTYPE_BF16) return MICF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  SampleBwdSet ss[2];
  Conv3BwdSet cs[2];
  for (int i = 0; i < ngroups; ++i) {
    const micf_offset_head_bwd_group& g = groups[i];
    if (!g.dxs || !g.hid || !g.flow || !g.xa || !g.ln_g || !g.ln_b || !g.w1 || !g.conv_w || !g.dxa || !g.dxn || !g.dhid || !g.dln_g ||
        !g.dln_b || !g.dw1)
      return MICF_EINVAL;
    ss[i] = SampleBwdSet{g.dxs, g.hid, g.ln_g, g.ln_b, g.w1, g.xa, g.flow, g.dxa, g.dhid, g.dln_g, g.dln_b, g.dw1,
                         CellLists{nullptr, nullptr, nullptr, nullptr, 0, nullptr}, nullptr};
    cs[i] = Conv3BwdSet{g.dhid, g.conv_w, g.conv_ws, g.dxn, g.dxa};
  }
  int rc = offset_sample_bwd_groups(ss, ngroups, B, D, H, W, C, eps, workspace, workspace_floats, s, defer_finish ? 1 : 0);
  if (rc != MICF_OK) return rc;
  rc = MICF_EUNSUPPORTED;
  if (groups[0].conv_ws && (ngroups == 1 || groups[1].conv_ws))
    rc = conv3_bwd_data_x_groups(cs, ngroups, C, 1, C, 1, B, D, H, W, kOffsetHidden, s, dtype, prepared);
  if (rc == MICF_EUNSUPPORTED) {
    for (int i = 0; i < ngroups; ++i) {
      rc = micf_conv3_bwd_data(cs[i].dy, 0, cs[i].w, cs[i].dx1, C, 1, cs[i].dx2, C, 1, B, D, H, W, kOffsetHidden, nullptr, 0, 0, dtype, stream);
      if (rc != MICF_OK) return rc;
    }
  }
  return rc;
}
