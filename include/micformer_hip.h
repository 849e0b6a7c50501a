/*
 * micformer_hip.h -- C-ABI of libmicformer_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * MicFormer training hot path (fxxJuses/MICFormer, MicFormer/models/MICFormer_self.py = "MS.py",
 * MicFormer/models/STN.py, MicFormer/loss/dice.py, MicFormer/train_mmwhs_noPad.py = "train.py").
 *
 * The reference has NO native / FFI layer (it is pure PyTorch); the boundary
 * this library sits behind is the nn.Module surface of MS.py.  Each entry point below replaces the
 * ATen op sequence of the cited reference lines and is what a ctypes binding inside those modules
 * calls.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the caller
 *     (PyTorch caching allocator), 16-byte aligned; kernels never allocate, free or synchronise, so
 *     every call is hipGraph-capturable; the stream is explicit (last argument).
 *   - activations are fp32, channels-last: (B, D, H, W, C) contiguous, C innermost ("tokens x C").
 *   - weights/biases are in the reference's state_dict layout (nn.Linear [N,K], Conv3d [N,C,k,k,k],
 *     ConvTranspose3d [C,N,k,k,k]); weight/bias gradients are ACCUMULATED (atomicAdd) into the
 *     destination, which the caller zero-fills (or which already holds .grad).
 *   - return 0 (MICF_OK) or a negative MICF_E* code; nothing throws across the ABI; re-entrant.
 */
#ifndef MICFORMER_HIP_H
#define MICFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* micf_stream_t; /* hipStream_t */

#define MICF_OK 0
#define MICF_EINVAL (-1)       /* bad argument (null pointer, non-positive size, misaligned) */
#define MICF_EUNSUPPORTED (-2) /* shape outside what the kernels implement (e.g. window > 8 tokens) */
#define MICF_ELAUNCH (-3)      /* HIP reported a launch error */

#define MICF_ABI_VERSION 1

/* dtype argument of the matrix-core entry points (nn.Linear, the 3x3x3 convolutions, the fused blocks): the arithmetic of the
 * MFMA products.  Everything in HBM stays fp32 in both modes (master weights, activations, gradients, Adam moments). */
#define MICF_DTYPE_F32 0  /* v_mfma_f32_16x16x4_f32: exact fp32 (bitwise a k-ordered fmaf chain) -- the parity mode */
#define MICF_DTYPE_BF16 1 /* v_mfma_f32_16x16x32_bf16: operands rounded to bf16 (RNE) at the fragment read, fp32 accumulate;
                             residual stream, LayerNorm, softmax, GELU, loss and everything stored stay fp32 */
#define MICF_DTYPE_BF16_ATTN_FP8 2 /* BASELINE config 4's leg: MICF_DTYPE_BF16 everywhere, except that the two products of window
                             attention (q k^T and P v, forward) take e4m3 operands -- on the tile-per-workgroup and few-token block
                             kernels as v_mfma_f32_16x16x32_fp8_fp8 on two windows x one head per tile (csrc/attn_fp8.h), on the per-op
                             entry point as the same rounding on the VALU.  Accepted by micf_block_fwd and micf_window_attn_fwd_fp8
                             only; every other entry point of a step in this mode is called with MICF_DTYPE_BF16.  The backward
                             is the bf16 one (straight-through). */

int micf_abi_version(void);
const char* micf_strerror(int code);

/* TEST HOOKS / MEASUREMENT PROBES, process-global, never set by the product path (csrc/common.h struct Options lists them with
 * their defaults and meanings): "block_wave", "block_recompute_h", "block_debug", "sample_tile", "sample_e", "cell_cap",
 * "tile_cap_hits", "tile_cap_cell", "tile_cap_voxel".  A hook selects a slower equivalent kernel or shrinks a capacity so an
 * overflow path runs (the parity tests); a probe skips work (timing only).  Unknown name: MICF_EINVAL.  Not thread-safe: set
 * between launches. */
int micf_set_option(const char* name, int value);
int micf_get_option(const char* name, int* value);

/* ---- LayerNorm over the last dim, eps inside rsqrt (nn.LayerNorm: MS.py:308,321,461,468,540,569,987,988).
 * Row r of the input is [x1[r, 0:c1] | x2[r, 0:C-c1]] (x2 may be NULL with c1 == C); the two-source form
 * replaces torch.cat([moving, fixed], -1) + norm2 (MS.py:1033-1034).  mean/rstd [rows] are saved for backward. */
int micf_layernorm_fwd(const float* x1, const float* x2, int c1, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int64_t rows, int C, float eps, micf_stream_t stream);
/* dx = LN'(dy) + add, where add [rows,C] is optional (NULL = 0; it may alias dx1 when c1 == C) -- this fuses the
 * residual-branch gradient sum of a block (d(shortcut) + d(norm(x))); dgamma/dbeta are accumulated. */
/* partials (optional): when non-NULL, dgamma / dbeta are NOT touched; instead every workgroup stores its [2C] partial sums at
 * partials[block * 2C] (micf_layernorm_bwd_partial_rows blocks; 0 = this shape has no partial form, pass NULL) and the caller
 * adds them later with micf_layernorm_bwd_finish -- many LayerNorms per launch, off the backward chain, no atomics. */
int micf_layernorm_bwd(const float* dy, const float* x1, const float* x2, int c1, const float* mean,
                       const float* rstd, const float* gamma, float* dx1, float* dx2, float* dgamma, float* dbeta,
                       int64_t rows, int C, const float* add, float* partials, micf_stream_t stream);
int micf_layernorm_bwd_partial_rows(int64_t rows, int C, int c1);
typedef struct micf_ln_finish_item {
  const float* partials;   /* [blocks, 2C] as written by micf_layernorm_bwd */
  float* dgamma;           /* [C] accumulated (may be NULL) */
  float* dbeta;            /* [C] accumulated (may be NULL) */
  int32_t blocks;
  int32_t C;
} micf_ln_finish_item;
/* `items` is HOST memory, read during the call only. */
int micf_layernorm_bwd_finish(const micf_ln_finish_item* items, int n, micf_stream_t stream);

/* ---- nn.Linear + fused epilogue (q/kv/proj MS.py:188-201,246-259; Mlp MS.py:28-34; concat_back_dim MS.py:1027-1030).
 *   lin = [a1 | a2] @ W^T + bias        a1 [M,k1], a2 [M,K-k1] (NULL when k1 == K), W [N,K]
 *   if pre_act != NULL: pre_act = lin   (saved for GELU backward)
 *   act: 0 none, 1 exact-erf GELU
 *   y = resid ? resid + s[row / rows_per_sample] * act(lin) : act(lin)    (s = DropPath scale per sample, NULL = 1)
 * The residual form fuses `shortcut + drop_path(x)` (MS.py:419,424,517,522). */
int micf_linear_fwd(const float* a1, const float* a2, int k1, const float* w, const float* bias, const float* resid,
                    const float* dp_scale, int64_t rows_per_sample, float* y, float* pre_act, int64_t M, int N, int K,
                    int act, int dtype, micf_stream_t stream);
/* d[a1|a2] (=|+=) (s * dy) @ W, optionally multiplied element-wise by GELU'(pre_act[M,K]) (fc2 -> fc1 seam). */
int micf_linear_bwd_data(const float* dy, const float* dp_scale, int64_t rows_per_sample, const float* w,
                         const float* pre_act, float* da1, float* da2, int k1, int accumulate, int64_t M, int N, int K,
                         int dtype, micf_stream_t stream);
/* dW += (s*dy)^T @ A, dbias += colsum(s*dy);  A = [a1|a2], or GELU(a1) when a_gelu != 0 (a1 = saved pre-activation).
 * workspace (optional, caller-owned scratch of workspace_floats fp32, see micf_linear_bwd_weight_workspace): when large
 * enough the token-split partial products are written there with plain 16-byte stores and reduced by a second launch
 * instead of being atomically added (device-scope fp32 atomics cost one fabric transaction each). */
int micf_linear_bwd_weight(const float* dy, const float* dp_scale, int64_t rows_per_sample, const float* a1,
                           const float* a2, int k1, int a_gelu, float* dw, float* dbias, int64_t M, int N, int K,
                           float* workspace, int64_t workspace_floats, int dtype, micf_stream_t stream);
/* Upper bound of the scratch (in floats) micf_linear_bwd_weight can use for (M, N, K). */
int64_t micf_linear_bwd_weight_workspace(int64_t M, int N, int K);

/* Weight gradients of MANY linear layers in one call (what autograd computes layer by layer for F.linear, MS.py:28-34,
 * 188-201, 246-259).  Nothing on the backward chain consumes a weight gradient, so the host may queue
 * (input, output-gradient) pairs during backward and flush them here once: a handful of chip-filling launches instead of
 * two small ones per layer.  Per item: dw[N,K] += (s*dy)^T @ a, dbias[N] += colsum(s*dy) (both accumulated).
 * Requirements (else MICF_EUNSUPPORTED): M % 16 == 0, N % 4 == 0, K % 4 == 0, 16-byte aligned a / dy / dw, and with
 * dp_scale: rows_per_sample % 16 == 0 and M % rows_per_sample == 0.  `items` is HOST memory (read during the call only);
 * the buffers it points to are device memory and must stay valid until the stream has run the launches. */
typedef struct micf_wgrad_item {
  const float* a;          /* [M, K] layer input (bfloat16 elements when operand_dtype == MICF_DTYPE_BF16) */
  const float* dy;         /* [M, N] gradient of the layer output (likewise) */
  const float* dp_scale;   /* [M / rows_per_sample] DropPath scale per sample, or NULL */
  float* dw;               /* [N, K] */
  float* dbias;            /* [N] or NULL */
  int64_t M;
  int64_t rows_per_sample;
  int32_t N;
  int32_t K;
  int32_t operand_dtype;   /* MICF_DTYPE_F32: a / dy are fp32; MICF_DTYPE_BF16: both were STORED as bfloat16 (what the fused block
                              kernels leave in bf16 mode); then M % 32 == 0, N % 8 == 0, K % 8 == 0, rows_per_sample % 32 == 0 */
  int32_t ldw;             /* row stride of dw in floats when dw is a column block of a wider matrix (a layer applied to a
                              concatenation [a1 | a2]: one item per part); 0 = K.  ldw >= K, ldw % 4 == 0 */
} micf_wgrad_item;
int micf_linear_bwd_weight_grouped(const micf_wgrad_item* items, int n, float* workspace, int64_t workspace_floats,
                                   int dtype, micf_stream_t stream);
/* Scratch floats the grouped call needs for these items (layers longer than one token split), or -1 if an item is unsupported. */
int64_t micf_linear_bwd_weight_grouped_workspace(const micf_wgrad_item* items, int n);

/* ---- reverse_patch_embedding (ConvTranspose3d(Ci -> Cm, k = s = P), MS.py:1037) followed by Head.out_conv
 * (Conv3d(Cm -> Co, 3, padding=1), MS.py:1053) as one linear map on the coarse token grid (no norm / activation sits
 * between them): with F = P + 2,
 *   T[q, (f, o)] = Bf[f, o] + sum_k Wb[f, o, k] x[q, k]       -- micf_linear_fwd(x, w = Wb [F^3*Co, Ci], bias = Bf)
 *   y[b, o, u]   = b_out[o] + sum of T[q, (u - P*q + 1, o)] over the in-volume coarse voxels q whose patch covers u
 * compose: Wb / Bf from (w_up [Ci,Cm,P,P,P], b_up [Cm], w_out [Co,Cm,3,3,3]).  col2im: T [B*Dc*Hc*Wc, F^3*Co] -> NCDHW
 * logits y [B, Co, P*Dc, P*Hc, P*Wc].  im2col: the transpose gather, U[q, (f, o)] = dy[b, o, P*q - 1 + f] (0 outside).
 * decompose: the chain rule through the composition, given dWb = U^T x and dBf = colsum(U) (micf_linear_bwd_weight):
 * dw_up, db_up, dw_out, db_out are ACCUMULATED.  Co <= 32, 2 <= P <= 8.
 * w_up_t (optional, compose and decompose): w_up viewed as [Ci, Cm*P^3] and transposed to [Cm*P^3, Ci]
 * (micf_weight_prep_grouped): the kernels then read it coalesced (5-10x faster); NULL = read w_up in place. */
int micf_head_tail_compose(const float* w_up, const float* b_up, const float* w_out, float* wb, float* bf, int Ci, int Cm,
                           int Co, int P, const float* w_up_t, micf_stream_t stream);
int micf_head_tail_col2im(const float* t, const float* b_out, float* y, int B, int Dc, int Hc, int Wc, int Co, int P,
                          micf_stream_t stream);
/* The same gather-sum for SLIDING-WINDOW INFERENCE (utils.py:226-234): T holds n windows; window i's logits are ADDED into the fp32
 * volume accumulator out [VB, Co, VD, VH, VW] at coords[4 i ..] = {volume sample, z0, y0, x0} and count [VB, VD, VH, VW] += 1
 * there -- the inferer's accumulate / count epilogue fused into the logits store (no [n, Co, roi] prediction tensor, no separate
 * accumulate launch).  coords is DEVICE memory (int32): a captured predictor graph is replayed with new origins. */
int micf_head_tail_col2im_sw(const float* t, const float* b_out, float* out, float* count, const int32_t* coords, int n, int Dc,
                             int Hc, int Wc, int Co, int P, int VB, int VD, int VH, int VW, micf_stream_t stream);
int micf_head_tail_im2col(const float* dy, float* u, int B, int Dc, int Hc, int Wc, int Co, int P, micf_stream_t stream);
int micf_head_tail_decompose(const float* dwb, const float* dbf, const float* w_up, const float* b_up, const float* w_out,
                             float* dw_up, float* db_up, float* dw_out, float* db_out, int Ci, int Cm, int Co, int P,
                             const float* w_up_t, micf_stream_t stream);

/* ---- the same map WITHOUT the patch matrices T / U (MICF_DTYPE_BF16, Co == 8, P == 4, Ci in {96, 192}, Wc % 16 == 0, Hc % 4 == 0;
 * micf_head_tail_fused_supported says so): the overlap-add is index arithmetic inside the GEMMs (head_tail_fused.hip).
 * pack: wb / bf (micf_head_tail_compose) and b_out -> the bf16 operand packs of the two kernels (micf_head_tail_pack_bytes(Ci, 0)
 * and (Ci, 1) bytes, 16-byte aligned), once per optimizer step.  fwd: x [B*Dc*Hc*Wc, Ci] -> NCDHW logits y [B, 8, 4Dc, 4Hc, 4Wc]
 * (bias terms included).  bwd_data: dy (NCDHW) -> dx [B*Dc*Hc*Wc, Ci] (overwritten). */
int micf_head_tail_fused_supported(int Dc, int Hc, int Wc, int Ci, int Co, int P, int dtype);
int64_t micf_head_tail_pack_bytes(int Ci, int which);
int micf_head_tail_pack(const float* wb, const float* bf, const float* b_out, void* pack_fwd, void* pack_bwd, int Ci, int Co,
                        int P, micf_stream_t stream);
int micf_head_tail_fwd_fused(const float* x, const void* pack_fwd, float* y, int B, int Dc, int Hc, int Wc, int Ci, int Co,
                             int P, micf_stream_t stream);
/* ... as the predictor of sliding-window inference (utils.py:226-234; the patch-matrix form is micf_head_tail_col2im_sw): x holds n
 * windows' coarse features [n*Dc*Hc*Wc, Ci]; window i's logits are ADDED into the fp32 volume accumulator out [VB, Co, VD, VH, VW]
 * at coords[4 i ..] = int32 {volume sample, z0, y0, x0} (device memory) and count [VB, VD, VH, VW] += 1 there (fp32 atomics:
 * windows of a batch overlap).  The caller guarantees every window lies inside the volume. */
int micf_head_tail_fwd_fused_sw(const float* x, const void* pack_fwd, float* out, float* count, const int32_t* coords, int n, int Dc,
                                int Hc, int Wc, int Ci, int Co, int P, int VB, int VD, int VH, int VW, micf_stream_t stream);
/* ... with MDiceLoss's forward (dice.py:130-166) in the logits store: every lane folds the terms of the logits it writes (target:
 * one-hot float planes [B, 8, 4Dc, 4Hc, 4Wc], or target_is_label != 0 the uint8 class map [B, 4Dc, 4Hc, 4Wc]) into
 * part [micf_head_tail_loss_parts(...)][32] floats; a one-workgroup finishing launch of the same call writes sums [8][4] (double:
 * sum p t, sum p^2, sum t^2, sum bce per class -- what micf_dice_bce_bwd reads) and the scalar loss.  Replaces
 * micf_head_tail_fwd_fused + micf_dice_bce_fwd: the logits are not re-read. */
int64_t micf_head_tail_loss_parts(int B, int Dc, int Hc, int Wc);
int micf_head_tail_fwd_loss_fused(const float* x, const void* pack_fwd, float* y, const void* target, int target_is_label,
                                  float* part, double* sums, float* loss, int B, int Dc, int Hc, int Wc, int Ci, int Co, int P,
                                  micf_stream_t stream);
int micf_head_tail_bwd_data_fused(const float* dy, const void* pack_bwd, float* dx, int B, int Dc, int Hc, int Wc, int Ci,
                                  int Co, int P, micf_stream_t stream);
/* dwb [216 * 8, Ci] / dbf [216 * 8] = the composed map's parameter gradient (OVERWRITTEN; feed micf_head_tail_decompose), reduced
 * over the coarse voxels with dy gathered on the fly -- additionally Wc % 32 == 0; workspace: ..._workspace floats (0 = this grid is
 * not covered: use micf_head_tail_im2col + micf_linear_bwd_weight). */
int micf_head_tail_bwd_weight_fused(const float* dy, const float* x, float* dwb, float* dbf, float* workspace,
                                    int64_t workspace_floats, int B, int Dc, int Hc, int Wc, int Ci, int Co, int P,
                                    micf_stream_t stream);
int64_t micf_head_tail_bwd_weight_fused_workspace(int B, int Dc, int Hc, int Wc, int Ci);

/* ---- (Cross)WindowAttention3D core on channels-last token grids, windows by index math (never materialised):
 * softmax((q*scale) k^T) v per head and per non-overlapping (wd,wh,ww) window (MS.py:193-200, 251-258;
 * window_partition/reverse MS.py:37-50,117-132).  q [T,ldq], k/v [T,ldkv] (k = kv, v = kv + C), o [T,ldo].
 * D,H,W must be multiples of the window (the host pads, MS.py:345-350); window tokens <= 8. */
int micf_window_attn_fwd(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo, int B,
                         int D, int H, int W, int C, int heads, int wd, int wh, int ww, float scale,
                         micf_stream_t stream);
/* The same with e4m3-rounded operands of both products (MICF_DTYPE_BF16_ATTN_FP8 on the shapes micf_block_fwd does not take). */
int micf_window_attn_fwd_fp8(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo, int B,
                         int D, int H, int W, int C, int heads, int wd, int wh, int ww, float scale,
                         micf_stream_t stream);
int micf_window_attn_bwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* d_o, int ldo,
                         float* dq, int lddq, float* dk, float* dv, int lddkv, int B, int D, int H, int W, int C,
                         int heads, int wd, int wh, int ww, float scale, micf_stream_t stream);

/* ---- 3x3x3, stride 1, zero-pad 1 convolution as implicit GEMM over channels-last sources.
 * Input channels are [x1 (c1) | x2 (c2)] (x2 may be NULL): conv_offset.0 on cat[LN(x), xa] (MS.py:314,354-356)
 * and Head.out_conv (MS.py:1046,1053).  w [N, c1+c2, 3,3,3].  y_layout 0: channels-last [T,N]; 1: NCDHW. */
int micf_conv3_fwd(const float* x1, int c1, const float* x2, int c2, const float* w, const float* bias, float* y,
                   int y_layout, int B, int D, int H, int W, int N, float* workspace, int64_t workspace_floats,
                   int prepared, int dtype, micf_stream_t stream);
/* scratch (floats) that enables the direct forward kernel for channels-last outputs with N <= 16 (0 = not applicable) */
int64_t micf_conv3_fwd_workspace(int N, int c1, int c2);
/* workspace (optional scratch, micf_conv3_bwd_data_workspace floats): enables the direct data-gradient kernel for
 * channels-last dy with N <= 16 (the weights are re-laid out as [tap][c][16 n] there by the same call). */
int micf_conv3_bwd_data(const float* dy, int dy_layout, const float* w, float* dx1, int c1, int acc1, float* dx2,
                        int c2, int acc2, int B, int D, int H, int W, int N, float* workspace, int64_t workspace_floats,
                        int prepared, int dtype, micf_stream_t stream);
/* `prepared` != 0 in the two calls above: `workspace` is not scratch but already holds the re-laid-out copy of w that the
 * direct kernel streams (fwd / bwd of micf_conv3_weight_prep_grouped below), so the call skips its own re-layout launch.
 * The weights of a training step only change in the optimiser: one grouped launch per step prepares all of them.
 * fwd: micf_conv3_fwd_workspace(N, Cin, 0) floats; bwd: micf_conv3_bwd_data_workspace(N, Cin, 0) floats; either may be NULL.
 * `items` is HOST memory, read during the call only. */
typedef struct micf_conv3_prep_item {
  const float* w; /* [N, Cin, 3, 3, 3], N <= 16 */
  float* fwd;
  float* bwd;
  int32_t N, Cin;
} micf_conv3_prep_item;
int micf_conv3_weight_prep_grouped(const micf_conv3_prep_item* items, int n, micf_stream_t stream);
int64_t micf_conv3_bwd_data_workspace(int N, int c1, int c2);
/* workspace (optional scratch, micf_conv3_bwd_weight_workspace floats; 0 = not used for this shape): enables the
 * register-resident MFMA weight-gradient kernel for channels-last dy with N == 16. */
int micf_conv3_bwd_weight(const float* dy, int dy_layout, const float* x1, int c1, const float* x2, int c2, float* dw,
                          float* dbias, int B, int D, int H, int W, int N, float* workspace, int64_t workspace_floats,
                          int dtype, micf_stream_t stream);
int64_t micf_conv3_bwd_weight_workspace(int B, int D, int H, int W, int N, int c1, int c2);
/* n layers of ONE shape (channels-last dy; e.g. the offset convs of both modalities of every cross pair of a stage) as one MFMA
 * launch + one reduce (<= 12 layers per launch; more are chunked); dw / dbias are ACCUMULATED.  Shapes the MFMA kernel does not
 * take (N != 16, W < 4, channel counts not multiples of 4) run layer by layer as micf_conv3_bwd_weight does.  `items` is HOST
 * memory read during the call only.  workspace: micf_conv3_bwd_weight_grouped_workspace floats (0 = none needed). */
typedef struct micf_conv3_wgrad_item {
  const float* dy;   /* [T, N] */
  const float* x1;   /* [T, c1] */
  const float* x2;   /* [T, c2] or NULL when c2 == 0 */
  float* dw;         /* [N, c1 + c2, 3, 3, 3] */
  float* dbias;      /* [N] or NULL */
} micf_conv3_wgrad_item;
int micf_conv3_bwd_weight_grouped(const micf_conv3_wgrad_item* items, int n, int c1, int c2, int B, int D, int H, int W, int N,
                                  float* workspace, int64_t workspace_floats, int dtype, micf_stream_t stream);
int64_t micf_conv3_bwd_weight_grouped_workspace(int n, int B, int D, int H, int W, int N, int c1, int c2);

/* ---- deformable re-sampling of the key/value modality (MS.py:313-318 tail, 326-337, 360-384; STN.py:9-32):
 *   off = W1 @ GELU(LN16(h))           h [T,16] = conv_offset.0 output, W1 [3,16] (no bias)
 *   flow = off + ref,  ref = ((i+.5)/Hk*2-1, (j+.5)/Wk*2-1, (k+.5)/Dk*2-1)   (divisors permuted as in MS.py:333-335)
 *   coord_ax = ((2*((idx+flow)/(S-1) - .5) + 1) * S - 1) / 2                   (STN.py:24 + grid_sample align_corners=False)
 *   xs = trilinear gather of xa at coord, zero padding; non-finite coords (S == 1) give 0.
 * flow [T,3] is saved for backward. */
int micf_offset_sample_fwd(const float* h, const float* ln_g, const float* ln_b, const float* w1, const float* xa,
                           float* flow, float* xs, int B, int D, int H, int W, int C, float eps, micf_stream_t stream);
/* dxa += (atomic scatter), dh [T,16] written, dln_g/dln_b/dw1 accumulated. */
int micf_offset_sample_bwd(const float* dxs, const float* h, const float* ln_g, const float* ln_b, const float* w1,
                           const float* xa, const float* flow, float* dxa, float* dh, float* dln_g, float* dln_b,
                           float* dw1, int B, int D, int H, int W, int C, float eps, float* workspace,
                           int64_t workspace_floats, micf_stream_t stream);
/* Scratch (in floats) that lets micf_offset_sample_bwd turn the atomic d(xa) scatter into a gather over per-cell token
 * lists (big grids only; 0 = the grid is small enough that the single atomic kernel is used anyway). */
int64_t micf_offset_sample_bwd_workspace(int B, int D, int H, int W);

/* Standalone SpatialTransformer.forward (STN.py:9-32) on channels-last src [B,D,H,W,C] with a given flow [T,3]
 * (voxel displacement, z,y,x): out [T,C].  Backward: dsrc += (atomic scatter, caller zero-fills), dflow [T,3] written. */
int micf_stn_fwd(const float* src, const float* flow, float* out, int B, int D, int H, int W, int C,
                 micf_stream_t stream);
int micf_stn_bwd(const float* dout, const float* src, const float* flow, float* dsrc, float* dflow, int B, int D,
                 int H, int W, int C, micf_stream_t stream);

/* ---- stride == kernel convolutions = per-patch GEMMs with index math.
 * patch_embed: Conv3d(1->E, k=s=p) on modality `mod` of vol [B,nmod,D,H,W], right-padded with zeros to a multiple
 * of p (PatchEmbed3D.forward MS.py:860-878); y [B,D',H',W',E] channels-last. */
int micf_patch_embed_fwd(const float* vol, int nmod, int mod, const float* w, const float* bias, float* y, int B,
                         int D, int H, int W, int E, int p, micf_stream_t stream);
int micf_patch_embed_bwd_weight(const float* dy, const float* vol, int nmod, int mod, float* dw, float* dbias, int B,
                                int D, int H, int W, int E, int p, micf_stream_t stream);
/* conv_down: Conv3d(C->N, k=s=2) on channels-last x [B,D,H,W,C], odd dims zero-padded (PatchMerging MS.py:548-557). */
int micf_conv_down_fwd(const float* x, const float* w, const float* bias, float* y, int B, int D, int H, int W, int C,
                       int N, micf_stream_t stream);
int micf_conv_down_bwd_data(const float* dy, const float* w, float* dx, int B, int D, int H, int W, int C, int N,
                            micf_stream_t stream);
int micf_conv_down_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int B, int D, int H, int W,
                              int C, int N, micf_stream_t stream);
/* The same k = s convolutions as plain GEMMs (what the autograd Functions launch): micf_space_to_depth gathers the k^3 voxels
 * of every patch of a channels-last x [B,D,H,W,C] (voxel stride C, sample stride batch_stride elements, 0 = D*H*W*C; a
 * single-channel NCDHW volume is C = 1 with batch_stride = nmod*D*H*W and x advanced to the modality) into one row of
 * a [B*ceil(D/k)*ceil(H/k)*ceil(W/k), C*k^3]; column = c*k^3 + (tz*k + ty)*k + tx, i.e. the flattening of the Conv3d weight
 * [N, C, k,k,k] (zeros beyond odd dims).  Then conv_down = micf_linear_fwd(a, w as [N, C k^3], bias); its weight gradient
 * is the linear one on the same a; its data gradient micf_linear_bwd_data followed by micf_depth_to_space (the inverse
 * scatter, + bias[c] when given, cropping what lies beyond D, H, W).  ConvTranspose3d [C, N, k,k,k]: y = depth_to_space(
 * micf_linear_bwd_data(x, w as [C, N k^3]), bias[N]); dx = micf_linear_fwd(space_to_depth(dy), w as [C, N k^3]);
 * dw = linear weight gradient of (dy := x, a := space_to_depth(dy)); dbias = micf_colsum(dy as [voxels, N]) (accumulated). */
int micf_space_to_depth(const float* x, float* a, int B, int D, int H, int W, int C, int k, int64_t batch_stride,
                        micf_stream_t stream);
int micf_depth_to_space(const float* a, const float* bias, float* y, int B, int D, int H, int W, int C, int k,
                        micf_stream_t stream);
/* ... the same with y = scatter(a) + bias + add, add [B, D, H, W, C] (may alias y): where a tensor has a second gradient -- the
 * skip connection of an encoder stage -- it is summed here instead of by a separate elementwise launch. */
int micf_depth_to_space_add(const float* a, const float* bias, const float* add, float* y, int B, int D, int H, int W, int C,
                            int k, micf_stream_t stream);
int micf_colsum(const float* x, float* out, int64_t M, int N, micf_stream_t stream);
/* conv_up: ConvTranspose3d(C->N, k=s in {2,4}) on channels-last x [B,D,H,W,C] -> y [B,kD,kH,kW,N] channels-last
 * (PatchExpand MS.py:575-577; reverse_patch_embedding MS.py:990,1037). */
int micf_conv_up_fwd(const float* x, const float* w, const float* bias, float* y, int B, int D, int H, int W, int C,
                     int N, int k, micf_stream_t stream);
int micf_conv_up_bwd_data(const float* dy, const float* w, float* dx, int B, int D, int H, int W, int C, int N, int k,
                          micf_stream_t stream);
int micf_conv_up_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int B, int D, int H, int W,
                            int C, int N, int k, micf_stream_t stream);

/* ---- sliding-window inference plumbing (utils.py:226-234: monai.inferers.sliding_window_inference(roi, sw_batch_size,
 * predictor, overlap), mode="constant"), one batch element, NCDHW fp32:
 *   window:     win[c, z, y, x] = vol[c, z0+z, y0+y, x0+x]                  (the predictor's input crop)
 *   accumulate: out[k, z0+z, y0+y, x0+x] += pred[k, z, y, x];  count[z0+z, y0+y, x0+x] += 1
 *   normalize:  out[k, v] /= count[v]                                       (every voxel is covered at least once) */
int micf_sw_window(const float* vol, float* win, int C, int D, int H, int W, int rd, int rh, int rw, int z0, int y0, int x0,
                   micf_stream_t stream);
int micf_sw_accumulate(const float* pred, float* out, float* count, int K, int D, int H, int W, int rd, int rh, int rw, int z0,
                       int y0, int x0, micf_stream_t stream);
int micf_sw_normalize(float* out, const float* count, int K, int64_t V, micf_stream_t stream);
/* Batched forms: crop `n` <= 64 windows of vol [B,C,D,H,W] into win [n,C,rd,rh,rw] / accumulate `n`
 * predictions [n,K,rd,rh,rw] into out [B,K,D,H,W] and count [B,D,H,W] in ONE launch each.  coords: HOST int32 [n,4] = (b, z0, y0,
 * x0), read during the call.  Windows of a batch may overlap (fp32 atomic adds). */
int micf_sw_window_batch(const float* vol, float* win, const int32_t* coords, int n, int B, int C, int D, int H, int W, int rd,
                         int rh, int rw, micf_stream_t stream);
int micf_sw_accumulate_batch(const float* pred, float* out, float* count, const int32_t* coords, int n, int B, int K, int D, int H,
                             int W, int rd, int rh, int rw, micf_stream_t stream);

/* ---- Input-pipeline tail on the device (train.py:116-125 MONAI dict transforms, in their order: RandFlipd x3 on image + label,
 * NormalizeIntensityd(nonzero=True, channel_wise=True), RandScaleIntensityd(0.1), RandShiftIntensityd(0.1)) and the loader's
 * float16 -> float32 cast (MMWHS.py:386, train.py:177).  vol [B,Cm,D,H,W] fp32 or fp16 (is_half); label map uint8 [B,D,H,W].
 * micf_intensity_stats: sums [B*Cm*3] doubles = {sum, sum of squares, count} over the NON-ZERO voxels of each channel.
 * micf_input_prepare: out = ((raw[flip] - mean) / std where raw != 0, else 0) * (1 + f) + o, label_out = label_in[flip];
 * params [B,5] = {flip D, flip H, flip W (0/1), f, o} on the device, or NULL (validation: normalise only).
 * (Normalising before or after the flips is the same: the statistics are permutation-invariant.) */
int micf_intensity_stats(const void* vol, int is_half, double* sums, int B, int Cm, int64_t V, micf_stream_t stream);
/* The same tail fused into PATCH EMBEDDING's gather (MS.py:860-878): rows0 / rows1 [B * ceil(D/k) * ceil(H/k) * ceil(W/k), k^3] =
 * the patch-row matrices of modality 0 / 1 (the operand of the k = s convolution as a GEMM: micf_space_to_depth's layout) computed
 * from the RAW volume with the flips as index arithmetic and normalise / scale / shift as one affine map per (sample, channel);
 * the prepared float32 volume is never written.  Cm <= 2; rows1 NULL for one modality.  micf_input_prepare with out == NULL then
 * only flips the label map. */
int micf_patch_rows_prepared(const void* vol, int is_half, const double* sums, const float* params, float* rows0, float* rows1, int B,
                             int Cm, int D, int H, int W, int k, micf_stream_t stream);
int micf_input_prepare(const void* vol, int is_half, const double* sums, const float* params, float* out, const uint8_t* label_in,
                       uint8_t* label_out, int B, int Cm, int D, int H, int W, micf_stream_t stream);

/* ---- zero-pad / crop of channels-last volumes (F.pad to window multiples MS.py:349-350,483; crop MS.py:399-400,497-498) */
int micf_pad3d(const float* src, float* dst, int B, int D, int H, int W, int Dp, int Hp, int Wp, int C,
               micf_stream_t stream);
int micf_crop3d(const float* src, float* dst, int B, int D, int H, int W, int Dp, int Hp, int Wp, int C,
                int accumulate, micf_stream_t stream);
/* F.interpolate(mode='trilinear', align_corners=True) on channels-last volumes (MS.py:1018-1025) and its adjoint. */
int micf_resize_trilinear_fwd(const float* src, float* dst, int B, int D, int H, int W, int Do, int Ho, int Wo, int C,
                              micf_stream_t stream);
int micf_resize_trilinear_bwd(const float* ddst, float* dsrc, int B, int D, int H, int W, int Do, int Ho, int Wo,
                              int C, micf_stream_t stream);

/* ---- MDiceLoss (dice.py:130-166): sigmoid-Dice (squared denominator, smooth 1, batch-joint sums) + per-channel
 * BCE on the sigmoid output, (0.7*sum dice + 0.3*sum bce)/K.  logits/target [B,K,V] (NCDHW).
 * sums [K*4] doubles {sum p t, sum p^2, sum t^2, sum bce} is scratch that the call zero-fills; loss [1]. */
int micf_dice_bce_fwd(const float* logits, const float* target, double* sums, float* loss, int B, int K, int64_t V,
                      micf_stream_t stream);
int micf_dice_bce_bwd(const float* logits, const float* target, const double* sums, const float* grad_out,
                      float* dlogits, int B, int K, int64_t V, micf_stream_t stream);
/* Same loss with the target given as the uint8 class map [B, V] the one-hot planes are expanded from (train.py:177):
 * t[b, k, v] = (label[b, v] == k).  8x fewer target bytes in HBM and over PCIe. */
int micf_dice_bce_label_fwd(const float* logits, const uint8_t* label, double* sums, float* loss, int B, int K, int64_t V,
                            micf_stream_t stream);
int micf_dice_bce_label_bwd(const float* logits, const uint8_t* label, const double* sums, const float* grad_out,
                            float* dlogits, int B, int K, int64_t V, micf_stream_t stream);
/* meandice of argmax masks (train.py:392-407, :305): classes 1..K-1, smooth 1e-6, batch-joint.
 * counts [3*K] int64 scratch (zero-filled by the call); label is the integer class map [B,V] (uint8); out [1] double. */
int micf_argmax_meandice(const float* logits, const uint8_t* label, uint8_t* mask_out, int64_t* counts, double* out,
                         int B, int K, int64_t V, micf_stream_t stream);

/* MDiceLoss(_Val).metric (dice.py:168-175, 223-230): per (sample, class) Dice of the thresholded prediction sigmoid(z) > 0.5
 * against the target plane -- one-hot float planes [B,K,V] (target_is_label = 0) or the uint8 class map [B,V] (1); an empty
 * target plane scores 1 if the prediction is empty too, else 0.  sums [B*K*3] doubles is scratch; out [B*K] floats. */
int micf_dice_metric(const float* logits, const void* target, int target_is_label, double* sums, float* out, int B, int K,
                     int64_t V, micf_stream_t stream);

/* ---- torch.optim.Adam(lr, betas, eps, weight_decay=0) over a flat fp32 buffer + CosineAnnealingLR stepped per
 * iteration (train.py:114,148,206-207).  state = {int64 step; double lr} on the device so a captured graph advances:
 * micf_adam_tick increments step and recomputes lr = eta_min + (base-eta_min)*(1+cos(pi*(step-1)/t_max))/2,
 * micf_adam_step applies the update with bias corrections for `step`; the gradient is read as grad_scale * g (1/world after a
 * sum all-reduce, 1 on one GPU).  p_bf16 (optional, n uint16_t): a bf16 (round-to-nearest-even) mirror of the updated
 * parameters, written in the same pass -- what the fused block kernels stream in bf16 mode, so no per-step conversion launch. */
int micf_adam_tick(void* state, double base_lr, double eta_min, int64_t t_max, micf_stream_t stream);
int micf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const void* state, float beta1,
                   float beta2, float eps, float grad_scale, void* p_bf16, micf_stream_t stream);


/* ---- two LayerNorms of the same shape per launch (the two modalities of a pair); fields as in micf_layernorm_fwd / _bwd
 * (single-source rows).  `items` is HOST memory, read during the call only. */
typedef struct micf_ln_pair_item {
  const float *x, *gamma, *beta;
  float *y, *mean, *rstd;
} micf_ln_pair_item;
typedef struct micf_ln_bwd_pair_item {
  const float *dy, *x, *mean, *rstd, *gamma;
  float *dx, *dgamma, *dbeta; /* dgamma / dbeta: accumulated (NULL with partials) */
  const float* add;           /* optional: dx = add + LN'(dy) */
  float* partials;            /* optional [micf_layernorm_bwd_partial_rows][2C] for micf_layernorm_bwd_finish */
} micf_ln_bwd_pair_item;
/* zero / zero_floats: optional buffer the same launch clears (the offset conv's atomically accumulated output) */
int micf_layernorm_fwd_pair(const micf_ln_pair_item* items, int n, int64_t rows, int C, float eps, float* zero,
                            int64_t zero_floats, micf_stream_t stream);
int micf_layernorm_bwd_pair(const micf_ln_bwd_pair_item* items, int n, int64_t rows, int C, micf_stream_t stream);

/* ---- The offset head of a cross block for both modalities of a cross pair in one call (csrc/offset_head.hip; MS.py:354-384):
 * hid = conv3(cat[xn, xa]) ; flow = conv1(GELU(LN16(hid))) ; xs = trilinear sample of raw xa at the reference points + flow.
 * Replaces micf_conv3_fwd + micf_offset_sample_fwd (and, backward, micf_offset_sample_bwd + micf_conv3_bwd_data) per
 * modality; the groups' kernels run as one launch each.  The conv weight gradient stays micf_conv3_bwd_weight (deferred).
 * conv_ws: the re-laid-out conv weight of micf_conv3_weight_prep_grouped (fwd / bwd layout; `prepared` = 1) or that many
 * floats of scratch (`prepared` = 0), NULL = generic conv path.  `groups` is HOST memory, read during the call only. */
typedef struct micf_offset_head_group {
  const float* xn;      /* [T, C] LN1(x): conv input channels 0 .. C-1 */
  const float* xa;      /* [T, C] raw other modality: conv input channels C .. 2C-1 and the sampled source */
  const float *conv_w, *conv_b; /* conv_offset.0: [16, 2C, 3,3,3], [16] */
  float* conv_ws;
  const float *ln_g, *ln_b, *w1; /* conv_offset.1.norm [16], conv_offset.3.weight [3, 16] */
  float *hid, *flow, *xs;        /* out: [T, 16] conv output, [T, 3] offsets, [T, C] sampled source */
} micf_offset_head_group;
typedef struct micf_offset_head_bwd_group {
  const float* dxs;     /* [T, C] gradient of the sampled source */
  const float *hid, *flow, *xa, *ln_g, *ln_b, *w1, *conv_w;
  float* conv_ws;
  float* dxa;           /* [T, C] ACCUMULATED: raw-source gradient of the sampler + the conv's input channels C .. 2C-1 */
  float* dxn;           /* [T, C] ACCUMULATED: the conv's input channels 0 .. C-1 (pre-LayerNorm-1 gradient) */
  float* dhid;          /* [T, 16] out: gradient of the conv output (operand of the conv weight gradient) */
  float *dln_g, *dln_b, *dw1; /* accumulated */
} micf_offset_head_bwd_group;
/* 1 = the conv accumulates atomically at this shape: hid must be zero when it starts (the call clears it unless hid_zeroed) */
int micf_offset_head_needs_zero(int B, int D, int H, int W, int C);
/* flow == NULL and xs == NULL in every group: the 3^3 conv only (the sampling then runs inside micf_block_fwd, see
 * micf_block_fwd_group.hid). */
int micf_offset_head_fwd(const micf_offset_head_group* groups, int ngroups, int B, int D, int H, int W, int C, float eps,
                         int prepared, int hid_zeroed, int dtype, micf_stream_t stream);
int64_t micf_offset_head_bwd_workspace(int ngroups, int B, int D, int H, int W);
/* defer_finish != 0 on a grid where micf_offset_head_finish_deferrable(...) != 0 (every grid since the backward kernel scatters
 * cell-list overflow itself; the predicate stays in the ABI): the launch that
 * sums the head-parameter partials into dw1 / dln_g / dln_b is left out -- nothing on the data path waits for it -- and the
 * caller issues it later with micf_offset_head_bwd_finish on the SAME workspace (which must stay untouched until then). */
int micf_offset_head_bwd(const micf_offset_head_bwd_group* groups, int ngroups, int B, int D, int H, int W, int C, float eps,
                         int prepared, float* workspace, int64_t workspace_floats, int dtype, int defer_finish, micf_stream_t stream);
int micf_offset_head_finish_deferrable(int B, int D, int H, int W);
int micf_offset_head_bwd_finish(const micf_offset_head_bwd_group* groups, int ngroups, int B, int D, int H, int W, int C,
                                float* workspace, int64_t workspace_floats, micf_stream_t stream);
/* ... of up to 16 deferring calls (any mix of grids and layers) as ONE launch; `calls` and the group arrays it points to are HOST
 * memory read during the call only. */
typedef struct micf_offset_head_finish_call {
  const micf_offset_head_bwd_group* groups;   /* as passed to the deferring micf_offset_head_bwd */
  int32_t ngroups, B, D, H, W, C;
  float* workspace;                            /* the workspace of that call */
  int64_t workspace_floats;
} micf_offset_head_finish_call;
int micf_offset_head_bwd_finish_grouped(const micf_offset_head_finish_call* calls, int ncalls, micf_stream_t stream);

/* ---- Fused window-local transformer block (csrc/block_fwd.hip, block_bwd.hip): everything of a TransformerBlock3D
 * (MS.py:430-524), and everything of a CrossTransformerBlock3D (MS.py:277-426) downstream of the deformable sampling, in ONE
 * launch per direction for up to two independent blocks of the same shape (the CT and MR branches of a depth slot,
 * MS.py:699-701).  Token grid (B, D, H, W) with even D, H, W and 2x2x2 windows, C % 16 == 0, head_dim 16 or 32,
 * hidden % 16 == 0; micf_block_tile_tokens returns 0 for shapes the fused kernels do not take (callers then use the
 * per-op entry points above).  dtype: arithmetic of the matrix-core products. */
typedef struct micf_block_fwd_group {
  const float* x;      /* [T, C] block input (residual stream), T = B*D*H*W tokens in natural order */
  const float* kvsrc;  /* cross: [T, C] deformably sampled raw other modality (K/V source, never normed); NULL = self attention */
  const float *ln1_g, *ln1_b, *bq, *bkv, *bp, *ln2_g, *ln2_b, *b1, *b2; /* state_dict layout */
  const void *wq, *wkv, *wp, *w1, *w2; /* the five weight matrices, state_dict layout [out, in]: K16-blocked shadow copies made by
                                          micf_weight_prep_grouped: float (bf16 = 3) for MICF_DTYPE_F32, bf16 (uint16_t,
                                          bf16 = 2) for MICF_DTYPE_BF16 */
  const float *s1, *s2; /* DropPath scales [B] of the two residual branches (NULL = 1) */
  float* y;            /* [T, C] block output */
  /* saved for backward / the deferred weight gradients, natural token order.  INFERENCE FORM (round 5): ALL of xn, q, kv, o, x1,
   * xn2, h, g, stats, kvs16, flow, xs32 NULL in every group -> the launch writes y only (not for the few-token decomposition at
   * C = 384 with head_dim 16: MICF_EUNSUPPORTED there); any other mix of NULL and non-NULL among q .. stats is MICF_EINVAL: */
  float *xn, *q, *kv, *o, *x1, *xn2; /* LN1(x) [T,C] (may be NULL: not written); q [T,C]; k|v [T,2C]; attention out [T,C]; x + s1*attn [T,C]; LN2(x1) [T,C] */
  void* h;             /* fc1 pre-activation [T, hidden]: float for MICF_DTYPE_F32, bf16 (uint16_t, round-to-nearest-even)
                          for MICF_DTYPE_BF16 -- only micf_block_bwd reads it (GELU'), so the bf16 mode stores it at half width.
                          May be NULL on the tile-per-workgroup kernels (micf_block_fuses_sampler != 0): not written (8 of the 36 bytes a bf16 block
                          writes per element of T*C); micf_block_bwd then rebuilds it from xn2 with one more GEMM phase */
  float* g;            /* GELU(h) [T, hidden] (operand of the fc2 weight gradient) */
  float* stats;        /* [4, T]: mean1, rstd1, mean2, rstd2 */
  void* kvs16;         /* cross, bf16 storage only (micf_block_saves_bf16): [T, C] bf16 copy of the K/V source, the operand of the kv
                          weight gradient; NULL otherwise */
  /* Cross block with the deformable sampling FUSED IN (micf_block_fuses_sampler(C, heads) != 0; then kvsrc must be NULL): the
   * launch itself runs LayerNorm(16) -> GELU -> 1^3 conv on the offset conv's output rows, adds the reference points and
   * gathers the trilinear taps of the raw other modality (what micf_offset_sample_fwd does as its own launch). */
  const float* hid;        /* [T, 16] output of conv_offset.0 (micf_offset_head_fwd with flow = xs = NULL), or NULL */
  const float* samp_src;   /* [T, C] raw other modality */
  const float *ln16_g, *ln16_b, *w1c; /* conv_offset.1.norm.{weight,bias} [16], conv_offset.3.weight [3, 16] */
  float* flow;             /* out [T, 3]: offsets + reference points (the sampler's backward reads them) */
  float* xs32;             /* out, optional, fp32 storage only: [T, C] the sampled rows (operand of the kv weight gradient) */
  /* Optional epilogue: the LayerNorm the NEXT block applies to this block's output -- a cross block's norm1 on the self block's y
   * (MS.py:343), whose result feeds conv_offset[0] -- written by the same launch from the y rows it holds, so that no
   * micf_layernorm_fwd(_pair) launch has to re-read y (same arithmetic: mean, biased variance, 1 / sqrt(var + eps), eps of this
   * call).  Available exactly where micf_block_fuses_sampler(C, heads) != 0 (MICF_EUNSUPPORTED on the few-token decomposition);
   * also in the inference form.  nln_g == NULL: no epilogue; otherwise nln_b, nln_y, nln_mean, nln_rstd must be non-NULL. */
  const float *nln_g, *nln_b;             /* [C] gain / bias of that LayerNorm */
  float *nln_y, *nln_mean, *nln_rstd;     /* out [T, C], [T], [T] */
  float* zero16;           /* optional, with nln_g: [T, 16] floats cleared by the launch (the atomically accumulated output of
                              micf_offset_head_fwd on small grids, hid_zeroed = 1) */
} micf_block_fwd_group;
int micf_block_fuses_sampler(int C, int heads);
/* STORAGE of the saved tensors.  micf_block_saves_bf16(C, heads, dtype) != 0 (MICF_DTYPE_BF16 on the tile-per-workgroup kernels,
 * C <= 192, and C = 384 with head_dim 32 since round 6): xn, q, kv, o, xn2, g (forward) and dq, dkv, dh, dx1 (backward) are bfloat16 arrays of the documented shapes (the
 * struct fields keep their float* type for the fp32 case), and kvs16 / dy16 receive bf16 copies of a cross block's K/V source and
 * of dy, so that every operand pair of the five nn.Linear weight gradients of a block is stored as bf16
 * (micf_wgrad_item.operand_dtype = MICF_DTYPE_BF16).  x1, y, stats, dx, dxs, dx1_copy and the LayerNorm partials are always
 * fp32.  Otherwise (MICF_DTYPE_F32, or the few-token decomposition at C = 384 / head_dim 16) everything is fp32 except h. */
int micf_block_saves_bf16(int C, int heads, int dtype);
typedef struct micf_block_bwd_group {
  const float* dy;     /* [T, C] gradient w.r.t. the block output (also fc2's output gradient for the weight-gradient GEMM) */
  const float *x, *x1, *stats, *q, *kv; /* as saved by micf_block_fwd (x and ln1_g may be NULL for a cross block) */
  const void* h;       /* ... float or bf16 by dtype, as micf_block_fwd wrote it; NULL (micf_block_recomputes_h only) = rebuild it
                          in the kernel as xn2 W1^T + b1 from the three fields at the end of this struct */
  const float *ln1_g, *ln2_g;
  const void *wqt, *wkvt, *wpt, *w1t, *w2t;  /* TRANSPOSED weights: q^T [C,C], kv^T [C,2C], proj^T [C,C], fc1^T [C,hidden],
                                                fc2^T [hidden,C] from micf_weight_prep_grouped (dst_t), K16-blocked:
                                                float (bf16 = 3) for MICF_DTYPE_F32, bf16 (bf16 = 2) for MICF_DTYPE_BF16 */
  const float *s1, *s2;
  float* dx;           /* self: [T, C] gradient w.r.t. the block input.  cross: the q path's PRE-LayerNorm gradient dq Wq (the
                          caller adds the offset-conv path and applies LN1 backward with add = dx1) */
  float* dxs;          /* cross: [T, C] gradient w.r.t. kvsrc; NULL = self attention */
  float *dx1, *dh, *dq, *dkv; /* [T,C], [T,hidden], [T,C], [T,2C]: output gradients of proj (before s1), fc1, q, kv */
  float *ln1_part, *ln2_part; /* [tiles, 2C] per-tile partial dgamma | dbeta for micf_layernorm_bwd_finish (NULL = skip;
                                 tiles = ceil(T / micf_block_tile_tokens)) */
  float* dx1_copy;     /* optional second copy of dx1 [T, C]: the buffer a cross PAIR then accumulates the other block's
                          K/V-source gradient and its own LN1 backward into (no zero fill, no separate add); always fp32 */
  void* dy16;          /* bf16 storage only: [T, C] bf16 copy of dy (operand of the fc2 weight gradient); NULL otherwise */
  /* h == NULL: the fc1 pre-activation is recomputed (same MFMA order as the forward: bit-identical in fp32 mode) from */
  const float* xn2;    /* LN2(x1) [T, C] as micf_block_fwd saved it (bf16 where micf_block_saves_bf16) */
  const void* w1;      /* mlp.fc1.weight [hidden, C], the forward's K16-blocked shadow copy (micf_block_fwd_group.w1) */
  const float* b1;     /* mlp.fc1.bias [hidden] */
  /* pre_d != NULL (self blocks on the tile-per-workgroup kernels with bf16 storage only): the LayerNorm backward that PRODUCES this
   * block's output gradient runs as the kernel's prologue -- the adjoint of the LayerNorm-1 of the cross block that consumed this
   * block's output (MS.py:343): dy_effective = dy + LN'(pre_d) with `dy` the partial sum the caller already holds (residual path +
   * the other branches).  One launch (micf_layernorm_bwd_pair) and one [T, C] round trip less per depth slot; dy_effective is never
   * written in fp32 (its bf16 copy dy16 is, as always). */
  const float* pre_d;    /* [T, C] gradient w.r.t. that LayerNorm's output */
  const float* pre_x;    /* [T, C] its input (= this block's forward output y) */
  const float *pre_mean, *pre_rstd; /* [T] its saved statistics */
  const float* pre_g;    /* [C] its gain */
  float* pre_part;       /* out [tiles, 2C]: per-tile partial dgamma | dbeta of that LayerNorm (micf_layernorm_bwd_finish) */
} micf_block_bwd_group;
/* != 0: the caller should pass h == NULL in both groups structs (see there).  Only with the option "block_recompute_h" set
 * (micf_set_option) and only for the tile-per-workgroup kernels (everything but the few-token decomposition at C = 384 / head_dim 16): a memory
 * switch (8 of the 34 saved bytes per element), not a speed one -- the extra GEMM phase of the backward costs more time than the
 * bytes save (LABNOTES.md, round 4). */
int micf_block_recomputes_h(int C, int heads);
int micf_block_tile_tokens(int B, int D, int H, int W, int C, int heads, int hidden, int backward);
/* MEASUREMENT PROBE, not a product entry point (tools/bench_persist.py; LABNOTES.md, round 5): `repeats` (1 .. 64) passes of
 * micf_block_fwd's tile kernel inside ONE launch with a device-wide barrier between the passes -- the cost of a persistent kernel
 * walking the depth slots of the 8^3 stage, measured against `repeats` launches.  Base 8^3 shape only (C = 192, 12 heads, bf16
 * mode; at most 256 workgroups, all resident): MICF_EUNSUPPORTED otherwise.  sync_ws: 2 device ints (barrier counter, error flag;
 * cleared by the call): the flag reads 1 afterwards if a barrier timed out (bounded spin: a mis-sized launch does not hang). */
int micf_block_fwd_persistent_probe(const micf_block_fwd_group* groups, int ngroups, int B, int D, int H, int W, int C, int heads,
                                    int hidden, float eps, float scale, int dtype, int repeats, int* sync_ws, micf_stream_t stream);
/* HAZARD PROBE, not a product entry point (LABNOTES.md, round 5): one 16 x 16 x 48 bf16 product on the matrix cores as
 * form 0 = v_mfma_f32_16x16x16_bf16 accumulating onto the v_mfma_f32_16x16x32_bf16 issued right before it, form 1 = two independent
 * products + a vector add (what csrc/block_wave.h::mfma48 does).  a / b: 64 lanes x 8 bf16 (A / B fragments of the 32-deep product),
 * c / d: 64 lanes x 4 bf16 (the 16-deep one), out: 64 lanes x 4 floats in accumulator order (row 4 (lane / 16) + i, column lane % 16). */
int micf_probe_mfma_chain(const void* a, const void* b, const void* c, const void* d, float* out, int form, micf_stream_t stream);
/* Per-step weight preparation for the fused block kernels: for each row-major fp32 matrix of a list (one launch per 64 items,
 * one read of the source) write dst = src and / or dst_t = src^T, as float (bf16 = 0), as row-major bf16 bit patterns in
 * uint16_t, round-to-nearest-even (bf16 = 1), or K16-BLOCKED as bf16 (bf16 = 2) or as float (bf16 = 3; rows and cols multiples
 * of 16): element (r, c) of an [R, Cn] output at ((r / 16) * (Cn / 16) + c / 16) * 256 + (r % 16) * 16 + c % 16, the order in
 * which the block kernels' matrix-core fragments read them (one load instruction = 1 KB of consecutive addresses).
 * Both block kernels stream BLOCKED copies: micf_block_fwd W (bf16 = 3 for MICF_DTYPE_F32, 2 for MICF_DTYPE_BF16),
 * micf_block_bwd W^T likewise.  `items` is HOST memory, read during the call only. */
typedef struct micf_weight_prep_item {
  const float* src; /* [rows, cols] */
  void* dst;        /* [rows, cols] or NULL */
  void* dst_t;      /* [cols, rows] or NULL */
  int32_t rows, cols, bf16 /* 0 | 1 | 2 | 3, see above */, reserved;
} micf_weight_prep_item;
int micf_weight_prep_grouped(const micf_weight_prep_item* items, int n, micf_stream_t stream);
/* `groups` is HOST memory, read during the call only. */
int micf_block_fwd(const micf_block_fwd_group* groups, int ngroups, int B, int D, int H, int W, int C, int heads,
                   int hidden, float eps, float scale, int dtype, micf_stream_t stream);
int micf_block_bwd(const micf_block_bwd_group* groups, int ngroups, int B, int D, int H, int W, int C, int heads,
                   int hidden, float scale, int dtype, micf_stream_t stream);

/* ---- step plumbing without ATen kernels.
 * micf_zero: optimizer.zero_grad() over the flat gradient buffer (train.py:183) as ONE memset node. */
int micf_zero(void* p, int64_t bytes, micf_stream_t stream);
/* DropPath draws of a whole forward (timm drop_path, scale_by_keep=True; MS.py:419,424,517,522): out[i*B + b] =
 * (u < keep[i]) / keep[i] with u ~ U[0,1) from a counter-based hash of (seed, call counter, i, b).
 * rng = {uint64 seed; uint64 counter} on the device: the kernel advances the counter itself, so a captured
 * graph draws fresh masks at every replay.  keep [n] (0 < keep <= 1), out [n*B]. */
int micf_drop_path_draw(void* rng, const float* keep, float* out, int n, int B, micf_stream_t stream);


/* ---- bf16 wire format of the data-parallel gradient exchange (no reference counterpart: train_mmwhs_noPad.py:410-414 is one
 * process): bf16 on the xGMI links, fp32 in every sum.  A slice of the flat fp32 gradient is rounded into
 * `ranks` equal zero-padded bf16 shards (pack: dst[i] = bf16_rne(src[i]), i < n; 0 up to `padded`, a multiple of 8), shard j of
 * every rank is sent to rank j (all-to-all), which adds its `ranks` received shards IN FP32 and rounds the sum once
 * (sum: out[i] = bf16_rne(sum_r recv[r * shard + i]), shard a multiple of 8), the reduced shards are all-gathered and widened back
 * (unpack: dst[i] = float(src[i]), i < n).  All buffers 16-byte aligned; the collectives themselves are the caller's (RCCL). */
int micf_grad_wire_pack(const float* src, int64_t n, void* dst, int64_t padded, micf_stream_t stream);
int micf_grad_wire_sum(const void* recv, int ranks, int64_t shard, void* out, micf_stream_t stream);
int micf_grad_wire_unpack(const void* src, float* dst, int64_t n, micf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MICFORMER_HIP_H */
