"""Tensor-level wrappers over the C-ABI (allocate outputs with torch, pass raw pointers + sizes + the current stream).

PyTorch is used here only for device memory and streams.  Every function takes contiguous fp32 CUDA tensors;
argument order follows include/micformer_hip.h.
"""
import ctypes

import torch

from . import _lib
from ._lib import call, f32, ptr


DETAIL = False   # profiler: key entries by shape as well as by entry point

# Arithmetic of the matrix-core products (include/micformer_hip.h micf_dtype): "fp32" = v_mfma_f32_16x16x4_f32, exact (the
# parity mode); "bf16" = v_mfma_f32_16x16x32_bf16 operands rounded to bf16 at the fragment read, fp32 accumulation.
_COMPUTE_DTYPE = "fp32"


def compute_dtype():
    return _COMPUTE_DTYPE


def _dt():
    return 1 if _COMPUTE_DTYPE == "bf16" else 0


def set_compute_dtype(name):
    """'fp32' (parity mode), 'bf16' (BASELINE config 2) or 'bf16+fp8attn' (BASELINE config 4's leg: bf16 mode with the two products
    of window attention, q k^T and P v, on e4m3 operands -- MICF_DTYPE_BF16_ATTN_FP8; compute_dtype() still reads 'bf16', every
    cache / shadow-weight key of the bf16 mode applies unchanged)."""
    global _COMPUTE_DTYPE, _ATTN_FP8
    if name not in ("fp32", "bf16", "bf16+fp8attn"):
        raise ValueError("compute dtype must be 'fp32', 'bf16' or 'bf16+fp8attn'")
    _ATTN_FP8 = name == "bf16+fp8attn"
    _COMPUTE_DTYPE = "bf16" if _ATTN_FP8 else name


_ATTN_FP8 = False


def attention_fp8():
    return _ATTN_FP8


def arith_mode():
    """The name set_compute_dtype() was last called with: what a captured graph bakes in (compute_dtype() + the attention flag)."""
    return "bf16+fp8attn" if _ATTN_FP8 else _COMPUTE_DTYPE


def _dt_attn():
    """dtype argument of the entry points that contain the attention products."""
    return 2 if _ATTN_FP8 else _dt()


def _cost(flops, *tensors, tag=None):
    """(algorithmic bytes, flops[, shape tag]) of one launch for bench.py's profiler: every listed tensor is moved once."""
    if _lib.PROFILE is None:
        return None
    c = (sum(t.numel() * t.element_size() for t in tensors if t is not None), int(flops))
    if DETAIL:
        c = c + (tag if tag is not None else "x".join(str(t.shape[0]) + "." + str(t.shape[-1]) for t in tensors[:2] if t is not None),)
    return c

_empty = torch.empty


def _new(like, *shape, dtype=torch.float32):
    return _empty(shape, dtype=dtype, device=like.device)


# ----------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x1, gamma, beta, eps, x2=None):
    """x1 [rows, c1] (+ x2 [rows, c2]) -> y [rows, C], mean [rows], rstd [rows]."""
    rows, c1 = x1.shape
    C = c1 + (x2.shape[1] if x2 is not None else 0)
    y = _new(x1, rows, C)
    mean = _new(x1, rows)
    rstd = _new(x1, rows)
    call("micf_layernorm_fwd", f32(x1), f32(x2), c1, f32(gamma), f32(beta), f32(y), f32(mean), f32(rstd), rows, C,
         float(eps),
         cost=_cost(8 * rows * C, x1, x2, y))
    return y, mean, rstd


def layernorm_bwd(dy, x1, mean, rstd, gamma, dgamma, dbeta, x2=None, add=None, defer=None, out=None):
    """Returns dx1 (and dx2); dgamma/dbeta are accumulated in place.  dx = LN'(dy) + add (out: destination, may alias add).
    defer: a list -> the parameter gradients are left as per-workgroup partials and (partials, blocks, C, dgamma, dbeta) is
    appended to it for a later `layernorm_bwd_finish` (shapes without a partial form accumulate immediately as usual)."""
    rows, c1 = x1.shape
    C = dy.shape[1]
    dx1 = out if out is not None else _new(x1, rows, c1)
    dx2 = _new(x1, rows, C - c1) if x2 is not None else None
    partials = None
    if defer is not None and x2 is None:
        nb = _lib.lib.micf_layernorm_bwd_partial_rows(rows, C, c1)
        if nb > 0 and not ((dy.data_ptr() | x1.data_ptr() | gamma.data_ptr() | (add.data_ptr() if add is not None else 0)) & 15):
            partials = _new(x1, nb, 2 * C)
            defer.append((partials, nb, C, dgamma, dbeta))
    call("micf_layernorm_bwd", f32(dy), f32(x1), f32(x2), c1, f32(mean), f32(rstd), f32(gamma), f32(dx1), f32(dx2),
         f32(dgamma), f32(dbeta), rows, C, f32(add), f32(partials),
         cost=_cost(12 * rows * C, dy, x1, x2, dx1, dx2, add))
    return (dx1, dx2) if x2 is not None else dx1


def layernorm_fwd_pair(xs, gammas, betas, eps, zero=None):
    """Two LayerNorms of the same shape in ONE launch: xs / gammas / betas are 2-lists -> [(y, mean, rstd)] * 2.
    zero: optional fp32 tensor (numel % 4 == 0) cleared by the same launch."""
    rows, C = xs[0].shape
    arr = (_lib.LnPairItem * 2)()
    outs = []
    for it, x, g, b in zip(arr, xs, gammas, betas):
        y, mean, rstd = _new(x, rows, C), _new(x, rows), _new(x, rows)
        it.x, it.gamma, it.beta, it.y, it.mean, it.rstd = f32(x), f32(g), f32(b), f32(y), f32(mean), f32(rstd)
        outs.append((y, mean, rstd))
    call("micf_layernorm_fwd_pair", ctypes.cast(arr, ctypes.c_void_p), len(xs), rows, C, float(eps), f32(zero),
         zero.numel() if zero is not None else 0,
         cost=_cost(8 * rows * C * len(xs), *xs, *[o[0] for o in outs]))
    return outs


def layernorm_bwd_pair(items, defers):
    """items: 2-list of dicts {dy, x, mean, rstd, gamma, dgamma, dbeta, add, out}; defers: per item a list (queue the parameter
    gradient partials for layernorm_bwd_finish) or None.  ONE launch; returns the two dx (= out when given)."""
    rows, C = items[0]["x"].shape
    arr = (_lib.LnBwdPairItem * 2)()
    outs = []
    nb = _lib.lib.micf_layernorm_bwd_partial_rows(rows, C, C)
    for it, d, defer in zip(arr, items, defers):
        dx = d.get("out") if d.get("out") is not None else _new(d["x"], rows, C)
        partials = None
        if defer is not None and nb > 0:
            partials = _new(d["x"], nb, 2 * C)
            defer.append((partials, nb, C, d["dgamma"], d["dbeta"]))
        it.dy, it.x, it.mean, it.rstd, it.gamma = f32(d["dy"]), f32(d["x"]), f32(d["mean"]), f32(d["rstd"]), f32(d["gamma"])
        it.dx, it.dgamma, it.dbeta, it.add, it.partials = f32(dx), f32(d["dgamma"]), f32(d["dbeta"]), f32(d.get("add")), f32(partials)
        outs.append(dx)
    call("micf_layernorm_bwd_pair", ctypes.cast(arr, ctypes.c_void_p), len(items), rows, C,
         cost=_cost(12 * rows * C * len(items), *[d["dy"] for d in items], *[d["x"] for d in items], *outs, *[d.get("add") for d in items]))
    return outs


class LnFinishPlan:
    """ctypes item array of queued LayerNorm parameter-gradient partials (reusable while the tensors keep their addresses)."""

    def __init__(self, items):
        self.items = list(items)
        self.n = len(self.items)
        self.arr = (_lib.LnFinishItem * max(self.n, 1))()
        for it, (partials, nb, C, dg, db) in zip(self.arr, self.items):
            it.partials, it.dgamma, it.dbeta, it.blocks, it.C = f32(partials), f32(dg), f32(db), nb, C

    def launch(self):
        if self.n:
            call("micf_layernorm_bwd_finish", ctypes.cast(self.arr, ctypes.c_void_p), self.n,
                 cost=_cost(0, *[i[0] for i in self.items[:1]]) if _lib.PROFILE is not None else None)


def layernorm_bwd_finish(items):
    LnFinishPlan(items).launch()


# ----------------------------------------------------------------------------- Linear
def linear_fwd(a1, w, bias, a2=None, resid=None, dp_scale=None, rows_per_sample=0, act=0, want_pre=False):
    M, k1 = a1.shape
    N, K = w.shape
    assert k1 + (a2.shape[1] if a2 is not None else 0) == K
    y = _new(a1, M, N)
    pre = _new(a1, M, N) if want_pre else None
    call("micf_linear_fwd", f32(a1), f32(a2), k1, f32(w), f32(bias), f32(resid), f32(dp_scale), rows_per_sample,
         f32(y), f32(pre), M, N, K, act, _dt(),
         cost=_cost(2 * M * N * K, a1, a2, w, resid, y, pre, tag=f'{M}x{N}x{K}'))
    return (y, pre) if want_pre else y


def linear_bwd_data(dy, w, dp_scale=None, rows_per_sample=0, pre_act=None, k1=None, out=None, accumulate=False):
    """d[a1|a2] = (s*dy) @ W (* GELU'(pre_act)).  Returns da1 or (da1, da2) when k1 < K."""
    M, N = dy.shape
    K = w.shape[1]
    k1 = K if k1 is None else k1
    if out is not None:
        da1, da2 = out, None
    else:
        da1 = _new(dy, M, k1)
        da2 = _new(dy, M, K - k1) if k1 < K else None
    call("micf_linear_bwd_data", f32(dy), f32(dp_scale), rows_per_sample, f32(w), f32(pre_act), f32(da1), f32(da2), k1,
         1 if accumulate else 0, M, N, K, _dt(),
         cost=_cost(2 * M * N * K, dy, w, pre_act, da1, da2, tag=f'{M}x{N}x{K}'))
    return (da1, da2) if da2 is not None else da1


_SCRATCH = {}
_SCRATCH_RETIRED = []


def scratch(device, nfloats):
    """Per-device fp32 scratch reused by every launch that wants a workspace (stream-ordered, so sharing is safe on one stream).
    It only grows; under HIP-graph capture it is allocated during the eager warm-up steps."""
    key = (device, _lib.stream())                                      # one scratch per stream: launches on different
    buf = _SCRATCH.get(key)                                            # streams may run concurrently
    if buf is None or buf.numel() < nfloats:
        if buf is not None:
            # a captured HIP graph may have baked the old buffer's address into its kernels (also a graph captured EARLIER in a
            # sequence of graphs of the same step): it is retired, never freed
            _SCRATCH_RETIRED.append(buf)
        buf = torch.empty(max(int(nfloats), 1 << 22), dtype=torch.float32, device=device)
        _SCRATCH[key] = buf
    return buf


def linear_bwd_weight(dy, a1, dw, dbias, a2=None, dp_scale=None, rows_per_sample=0, a_gelu=False):
    """dw += (s*dy)^T [a1|a2] (or GELU(a1)); dbias += colsum(s*dy).  dw/dbias are accumulated in place."""
    M, N = dy.shape
    k1 = a1.shape[1]
    K = dw.shape[1]
    need = _lib.lib.micf_linear_bwd_weight_workspace(M, N, K)
    ws = scratch(dy.device, need) if need > 0 else None
    call("micf_linear_bwd_weight", f32(dy), f32(dp_scale), rows_per_sample, f32(a1), f32(a2), k1, 1 if a_gelu else 0,
         f32(dw), f32(dbias), M, N, K, f32(ws), ws.numel() if ws is not None else 0, _dt(),
         cost=_cost(2 * M * N * K, dy, a1, a2, dw, tag=f'{M}x{N}x{K}'))


def wgrad_groupable(dy, a1, dp_scale=None, rows_per_sample=0):
    """Shape / alignment gate of micf_linear_bwd_weight_grouped for one layer."""
    M, N = dy.shape
    K = a1.shape[1]
    if dy.dtype != a1.dtype:
        return False
    if dy.dtype == torch.bfloat16:          # operands stored as bf16 (the fused block kernels' outputs in bf16 mode)
        if M % 32 or N % 8 or K % 8 or (dy.data_ptr() | a1.data_ptr()) & 15 or (dp_scale is not None and rows_per_sample % 32):
            return False
    if N % 4 or K % 4 or (dy.data_ptr() | a1.data_ptr()) & 15:         # (fp32 operands: any token count -- the kernel adds ragged tails)
        return False
    if dp_scale is not None and (rows_per_sample <= 0 or M % rows_per_sample):
        return False
    return True


GROUP_ITEMS = 32          # layers per kernel launch inside micf_linear_bwd_weight_grouped (linear_grouped.hip::kGroupMax)


class GroupedWgradPlan:
    """The ctypes item array of a list of queued weight gradients, reusable while the tensors keep their addresses (HIP-graph
    memory): `launch(first, count)` issues the layers [first, first+count) on the current stream."""

    def __init__(self, items):
        self.items = list(items)                      # holds the tensors (and so their addresses) alive
        n = self.n = len(self.items)
        self.arr = (_lib.WgradItem * max(n, 1))()
        self.flops = [0] * n
        self.nbytes = [0] * n
        for k, (it, (dy, a, dw, db, sc, rps)) in enumerate(zip(self.arr, self.items)):
            M, N = dy.shape
            K = a.shape[1]
            if dy.dtype != a.dtype or dy.dtype not in (torch.float32, torch.bfloat16):
                raise _lib.MicfError("grouped weight gradient: both operands must be float32 or both bfloat16")
            if dw.dtype != torch.float32 or not dw.is_cuda:
                raise _lib.MicfError("grouped weight gradient: dw must be a float32 device tensor")
            it.a, it.dy, it.dp_scale, it.dw, it.dbias = ptr(a), ptr(dy), f32(sc), dw.data_ptr(), f32(db)
            it.M, it.rows_per_sample, it.N, it.K = M, int(rps) if sc is not None else 0, N, K
            it.operand_dtype = 1 if dy.dtype == torch.bfloat16 else 0
            if dw.dim() != 2 or dw.stride(1) != 1 or dw.shape != (N, K):
                raise _lib.MicfError("grouped weight gradient: dw must be an [N, K] (column block of a) row-major matrix")
            it.ldw = 0 if dw.stride(0) == K else dw.stride(0)
            self.flops[k] = 2 * M * N * K
            self.nbytes[k] = dy.element_size() * (dy.numel() + a.numel()) + 8 * dw.numel()
        self.device = self.items[0][0].device if n else None
        self.item_bytes = ctypes.sizeof(_lib.WgradItem)

    def launch(self, first=0, count=None):
        count = self.n - first if count is None else count
        if count <= 0:
            return
        ptr_ = ctypes.c_void_p(ctypes.addressof(self.arr) + first * self.item_bytes)
        need = _lib.lib.micf_linear_bwd_weight_grouped_workspace(ptr_, count)
        if need < 0:
            raise _lib.MicfError("micf_linear_bwd_weight_grouped: unsupported item")
        ws = scratch(self.device, need) if need > 0 else None
        cost = None
        if _lib.PROFILE is not None:
            cost = (sum(self.nbytes[first:first + count]), sum(self.flops[first:first + count])) + ((f"{count}",) if DETAIL else ())
        call("micf_linear_bwd_weight_grouped", ptr_, count, f32(ws), ws.numel() if ws is not None else 0, _dt(), cost=cost)


def linear_bwd_weight_grouped(items):
    """items: list of (dy [M,N], a [M,K], dw [N,K], dbias [N] | None, dp_scale | None, rows_per_sample).  One call."""
    if items:
        GroupedWgradPlan(items).launch()


# ----------------------------------------------------------------------------- window attention
def window_attn_fwd(q, kv, dims, heads, ws, scale):
    B, D, H, W = dims
    T, C = q.shape
    o = _new(q, T, C)
    call("micf_window_attn_fwd_fp8" if _ATTN_FP8 else "micf_window_attn_fwd", f32(q), C, f32(kv), kv.data_ptr() + 4 * C, 2 * C, f32(o), C, B, D, H, W, C, heads,
         ws[0], ws[1], ws[2], float(scale),
         cost=_cost(4 * T * C * ws[0] * ws[1] * ws[2], q, kv, o))
    return o


def window_attn_bwd(q, kv, d_o, dims, heads, ws, scale):
    B, D, H, W = dims
    T, C = q.shape
    dq = _new(q, T, C)
    dkv = _new(q, T, 2 * C)
    call("micf_window_attn_bwd", f32(q), C, f32(kv), kv.data_ptr() + 4 * C, 2 * C, f32(d_o), C, f32(dq), C, f32(dkv),
         dkv.data_ptr() + 4 * C, 2 * C, B, D, H, W, C, heads, ws[0], ws[1], ws[2], float(scale),
         cost=_cost(8 * T * C * ws[0] * ws[1] * ws[2], q, kv, d_o, dq, dkv))
    return dq, dkv


def window_attn_fwd_qkv(qkv, dims, heads, ws, scale):
    """Self attention on a packed [T, 3C] = [q | k | v] token matrix (one fused q/kv projection)."""
    B, D, H, W = dims
    T, C3 = qkv.shape
    C = C3 // 3
    o = _new(qkv, T, C)
    base = qkv.data_ptr()
    call("micf_window_attn_fwd_fp8" if _ATTN_FP8 else "micf_window_attn_fwd", base, C3, base + 4 * C, base + 8 * C, C3, f32(o), C, B, D, H, W, C, heads,
         ws[0], ws[1], ws[2], float(scale),
         cost=_cost(4 * T * C * ws[0] * ws[1] * ws[2], qkv, o))
    return o


def window_attn_bwd_qkv(qkv, d_o, dims, heads, ws, scale):
    """-> dqkv [T, 3C] = [dq | dk | dv] for the packed layout."""
    B, D, H, W = dims
    T, C3 = qkv.shape
    C = C3 // 3
    dqkv = _new(qkv, T, C3)
    base, dbase = qkv.data_ptr(), dqkv.data_ptr()
    call("micf_window_attn_bwd", base, C3, base + 4 * C, base + 8 * C, C3, f32(d_o), C, dbase, C3, dbase + 4 * C,
         dbase + 8 * C, C3, B, D, H, W, C, heads, ws[0], ws[1], ws[2], float(scale),
         cost=_cost(8 * T * C * ws[0] * ws[1] * ws[2], qkv, d_o, dqkv))
    return dqkv


# ----------------------------------------------------------------------------- conv 3x3x3
def conv3_fwd(x1, w, bias, dims, x2=None, ncdhw_out=False):
    B, D, H, W = dims
    c1 = x1.shape[-1]
    c2 = x2.shape[-1] if x2 is not None else 0
    N = w.shape[0]
    y = _new(x1, B, N, D, H, W) if ncdhw_out else _new(x1, B * D * H * W, N)
    need = 0 if ncdhw_out else _lib.lib.micf_conv3_fwd_workspace(N, c1, c2)
    ws = _conv_layouts(w)[0] if need > 0 else None               # engine mode: re-laid-out copy, refreshed once per step
    prepared = 1 if ws is not None else 0
    if ws is None and need > 0:
        ws = scratch(x1.device, need)
    call("micf_conv3_fwd", f32(x1), c1, f32(x2), c2, f32(w), f32(bias), f32(y), 1 if ncdhw_out else 0, B, D, H, W, N,
         f32(ws), ws.numel() if ws is not None else 0, prepared, _dt(),
         cost=_cost(2 * B * D * H * W * 27 * (c1 + c2) * N, x1, x2, w, y))
    return y


def conv3_bwd_data(dy, w, dims, c1, c2=0, ncdhw=False, dx1=None, dx2=None, acc1=False, acc2=False, want1=True, want2=True):
    B, D, H, W = dims
    N = w.shape[0]
    T = B * D * H * W
    if dx1 is None and want1:
        dx1 = _new(dy, T, c1)
    if dx2 is None and want2 and c2 > 0:
        dx2 = _new(dy, T, c2)
    need = 0 if ncdhw else _lib.lib.micf_conv3_bwd_data_workspace(N, c1, c2)
    ws = _conv_layouts(w)[1] if need > 0 else None
    prepared = 1 if ws is not None else 0
    if ws is None and need > 0:
        ws = scratch(dy.device, need)
    call("micf_conv3_bwd_data", f32(dy), 1 if ncdhw else 0, f32(w), f32(dx1), c1, 1 if acc1 else 0, f32(dx2), c2,
         1 if acc2 else 0, B, D, H, W, N, f32(ws), ws.numel() if ws is not None else 0, prepared, _dt(),
         cost=_cost(2 * T * 27 * (c1 + c2) * N, dy, w, dx1, dx2))
    return dx1, dx2


def conv3_bwd_weight(dy, x1, dw, dbias, dims, x2=None, ncdhw=False):
    B, D, H, W = dims
    c1 = x1.shape[-1]
    c2 = x2.shape[-1] if x2 is not None else 0
    N = dw.shape[0]
    need = 0 if ncdhw else _lib.lib.micf_conv3_bwd_weight_workspace(B, D, H, W, N, c1, c2)
    ws = scratch(dy.device, need) if need > 0 else None
    call("micf_conv3_bwd_weight", f32(dy), 1 if ncdhw else 0, f32(x1), c1, f32(x2), c2, f32(dw), f32(dbias), B, D, H, W, N,
         f32(ws), ws.numel() if ws is not None else 0, _dt(),
         cost=_cost(2 * B * D * H * W * 27 * (c1 + c2) * N, dy, x1, x2, dw))


def conv3_bwd_weight_grouped(items, dims):
    """items: [(dy [T,N], x1 [T,c1], x2 [T,c2] | None, dw, dbias | None)] of ONE shape (channels-last): one launch + one reduce."""
    if len(items) == 1:
        dy, x1, x2, dw, db = items[0]
        return conv3_bwd_weight(dy, x1, dw, db, dims, x2=x2)
    B, D, H, W = dims
    dy0, x10, x20, dw0, _ = items[0]
    c1 = x10.shape[-1]
    c2 = x20.shape[-1] if x20 is not None else 0
    N = dw0.shape[0]
    n = len(items)
    arr = (_lib.Conv3WgradItem * n)()
    for k, (dy, x1, x2, dw, db) in enumerate(items):
        assert dy.shape == dy0.shape and x1.shape == x10.shape and dw.shape == dw0.shape
        arr[k].dy, arr[k].x1, arr[k].x2 = f32(dy), f32(x1), f32(x2)
        arr[k].dw, arr[k].dbias = f32(dw), f32(db)
    need = _lib.lib.micf_conv3_bwd_weight_grouped_workspace(n, B, D, H, W, N, c1, c2)
    ws = scratch(dy0.device, need) if need > 0 else None
    flat = [t for it in items for t in it if t is not None]
    call("micf_conv3_bwd_weight_grouped", ctypes.addressof(arr), n, c1, c2, B, D, H, W, N, f32(ws),
         ws.numel() if ws is not None else 0, _dt(),
         cost=_cost(2 * n * B * D * H * W * 27 * (c1 + c2) * N, *flat))


# ----------------------------------------------------------------------------- offset head + deformable sampling
def offset_sample_fwd(h, ln_g, ln_b, w1, xa, dims, eps):
    B, D, H, W = dims
    T, C = xa.shape
    flow = _new(xa, T, 3)
    xs = _new(xa, T, C)
    call("micf_offset_sample_fwd", f32(h), f32(ln_g), f32(ln_b), f32(w1), f32(xa), f32(flow), f32(xs), B, D, H, W, C,
         float(eps),
         cost=_cost(T * (16 * C + 400), h, xa, flow, xs))
    return flow, xs


def offset_sample_bwd(dxs, h, ln_g, ln_b, w1, xa, flow, dxa, dln_g, dln_b, dw1, dims, eps):
    """dxa (pre-zeroed or holding a partial sum) is accumulated; returns dh [T,16]."""
    B, D, H, W = dims
    T, C = xa.shape
    dh = _new(xa, T, h.shape[1])
    need = _lib.lib.micf_offset_sample_bwd_workspace(B, D, H, W)
    ws = scratch(xa.device, need) if need > 0 else None
    call("micf_offset_sample_bwd", f32(dxs), f32(h), f32(ln_g), f32(ln_b), f32(w1), f32(xa), f32(flow), f32(dxa), f32(dh),
         f32(dln_g), f32(dln_b), f32(dw1), B, D, H, W, C, float(eps), f32(ws), ws.numel() if ws is not None else 0,
         cost=_cost(T * (40 * C + 800), dxs, h, xa, flow, dxa, dxa, dh))
    return dh


def offset_head_bwd_workspace(n, dims):
    return int(_lib.lib.micf_offset_head_bwd_workspace(n, *dims))


def offset_head_needs_zero(dims, C):
    B, D, H, W = dims
    return bool(_lib.lib.micf_offset_head_needs_zero(B, D, H, W, C))


def offset_head_fwd(groups, dims, eps, hid=None, sample=True):
    """groups: 1 or 2 dicts {xn [T,C], xa [T,C], P {conv_offset.* parameters}}.  The whole head(s) in one call (2-3 launches).
    hid: optional pre-zeroed [n, T, 16] buffer (see offset_head_needs_zero).  Returns per group (hid, flow, xs).
    sample=False: the 3^3 conv only -- (hid, None, None); the sampling then runs inside block_fwd (its `hid` group field)."""
    B, D, H, W = dims
    T, C = groups[0]["xn"].shape
    n = len(groups)
    zeroed = hid is not None
    if hid is None:
        hid = _new(groups[0]["xn"], n, T, 16)
    arr = (_lib.OffsetHeadGroup * 2)()
    outs, keep, prepared = [], [], None
    for i, (it, gd) in enumerate(zip(arr, groups)):
        P = gd["P"]
        w = P["conv_offset.0.weight"]
        ws = _conv_layouts(w)[0]
        prepared = (ws is not None) if prepared is None else (prepared and ws is not None)
        if ws is None:
            need = _lib.lib.micf_conv3_fwd_workspace(16, C, C)
            ws = _new(w, need) if need > 0 else None
        keep.append(ws)
        flow, xs = (_new(gd["xn"], T, 3), _new(gd["xn"], T, C)) if sample else (None, None)
        it.xn, it.xa, it.conv_w, it.conv_b, it.conv_ws = f32(gd["xn"]), f32(gd["xa"]), f32(w), f32(P["conv_offset.0.bias"]), f32(ws)
        it.ln_g, it.ln_b, it.w1 = f32(P["conv_offset.1.norm.weight"]), f32(P["conv_offset.1.norm.bias"]), f32(P["conv_offset.3.weight"])
        it.hid, it.flow, it.xs = f32(hid[i]), f32(flow), f32(xs)
        outs.append((hid[i], flow, xs))
    if not prepared:                      # mixed: let the call re-lay-out into private scratch
        for it, gd in zip(arr, groups):
            need = _lib.lib.micf_conv3_fwd_workspace(16, C, C)
            ws = _new(gd["xn"], need) if need > 0 else None
            keep.append(ws)
            it.conv_ws = f32(ws)
    call("micf_offset_head_fwd", ctypes.cast(arr, ctypes.c_void_p), n, B, D, H, W, C, float(eps), 1 if prepared else 0,
         1 if zeroed else 0, _dt(), cost=_cost(n * (2 * T * 27 * 2 * C * 16 + (T * (20 * C + 400) if sample else 0)), *[g["xn"] for g in groups],
                                                *[g["xa"] for g in groups], *[o[2] for o in outs], *[o[0] for o in outs]))
    del keep
    return outs


def offset_head_finish_deferrable(dims):
    return bool(_lib.lib.micf_offset_head_finish_deferrable(*dims))


def _head_bwd_array(groups, dhids):
    arr = (_lib.OffsetHeadBwdGroup * 2)()
    for it, gd, dhid in zip(arr, groups, dhids):
        P, G = gd["P"], gd["G"]
        it.dxs, it.hid, it.flow, it.xa = f32(gd["dxs"]), f32(gd["hid"]), f32(gd["flow"]), f32(gd["xa"])
        it.ln_g, it.ln_b, it.w1 = f32(P["conv_offset.1.norm.weight"]), f32(P["conv_offset.1.norm.bias"]), f32(P["conv_offset.3.weight"])
        it.conv_w = f32(P["conv_offset.0.weight"])
        it.dxa, it.dxn, it.dhid = f32(gd["dxa"]), f32(gd["dxn"]), f32(dhid)
        it.dln_g, it.dln_b, it.dw1 = f32(G["conv_offset.1.norm.weight"]), f32(G["conv_offset.1.norm.bias"]), f32(G["conv_offset.3.weight"])
    return arr


def offset_head_bwd_finish(groups, dhids, dims, ws):
    """The finishing launch a deferring offset_head_bwd left out (head-parameter partial sums -> dw1 / dln_g / dln_b)."""
    B, D, H, W = dims
    C = groups[0]["xa"].shape[1]
    arr = _head_bwd_array(groups, dhids)
    call("micf_offset_head_bwd_finish", ctypes.cast(arr, ctypes.c_void_p), len(groups), B, D, H, W, C, f32(ws), ws.numel())


def offset_head_bwd_finish_grouped(calls):
    """calls: [(groups, dhids, dims, ws)] of deferring offset_head_bwd calls (any mix of grids): their finishing launches as one."""
    for first in range(0, len(calls), 16):
        part = calls[first:first + 16]
        arr = (_lib.OffsetHeadFinishCall * len(part))()
        keep = []
        for it, (groups, dhids, dims, ws) in zip(arr, part):
            ga = _head_bwd_array(groups, dhids)
            keep.append(ga)
            it.groups = ctypes.cast(ga, ctypes.c_void_p)
            it.ngroups = len(groups)
            it.B, it.D, it.H, it.W = dims
            it.C = groups[0]["xa"].shape[1]
            it.workspace, it.workspace_floats = f32(ws), ws.numel()
        call("micf_offset_head_bwd_finish_grouped", ctypes.addressof(arr), len(part))


def offset_head_bwd(groups, dims, eps, defer_ws=None):
    """groups: 1 or 2 dicts {dxs, hid, flow, xa, P, G, dxa (accumulated), dxn (accumulated)}.  Sampler adjoint(s) + conv data
    gradient(s) in one call; returns the dhid [T,16] of every group (operand of the conv weight gradient).
    defer_ws: a dedicated workspace tensor (offset_head_finish_deferrable grids only) -- the finishing launch is then left to a later
    offset_head_bwd_finish(groups, dhids, dims, defer_ws)."""
    B, D, H, W = dims
    T, C = groups[0]["xa"].shape
    n = len(groups)
    arr = (_lib.OffsetHeadBwdGroup * 2)()
    outs, keep, prepared = [], [], None
    for it, gd in zip(arr, groups):
        P, G = gd["P"], gd["G"]
        w = P["conv_offset.0.weight"]
        ws = _conv_layouts(w)[1]
        prepared = (ws is not None) if prepared is None else (prepared and ws is not None)
        keep.append(ws)
        dhid = _new(gd["xa"], T, 16)
        it.dxs, it.hid, it.flow, it.xa = f32(gd["dxs"]), f32(gd["hid"]), f32(gd["flow"]), f32(gd["xa"])
        it.ln_g, it.ln_b, it.w1 = f32(P["conv_offset.1.norm.weight"]), f32(P["conv_offset.1.norm.bias"]), f32(P["conv_offset.3.weight"])
        it.conv_w, it.conv_ws = f32(w), f32(ws)
        it.dxa, it.dxn, it.dhid = f32(gd["dxa"]), f32(gd["dxn"]), f32(dhid)
        it.dln_g, it.dln_b, it.dw1 = f32(G["conv_offset.1.norm.weight"]), f32(G["conv_offset.1.norm.bias"]), f32(G["conv_offset.3.weight"])
        outs.append(dhid)
    if not prepared:
        for it, gd in zip(arr, groups):
            need = _lib.lib.micf_conv3_bwd_data_workspace(16, C, C)
            ws = _new(gd["xa"], need) if need > 0 else None
            keep.append(ws)
            it.conv_ws = f32(ws)
    need = _lib.lib.micf_offset_head_bwd_workspace(n, B, D, H, W)
    ws = defer_ws if defer_ws is not None else (scratch(groups[0]["xa"].device, need) if need > 0 else None)
    call("micf_offset_head_bwd", ctypes.cast(arr, ctypes.c_void_p), n, B, D, H, W, C, float(eps), 1 if prepared else 0, f32(ws),
         ws.numel() if ws is not None else 0, _dt(), 1 if defer_ws is not None else 0,
         cost=_cost(n * (2 * T * 27 * 2 * C * 16 + T * (40 * C + 800)), *[g["dxs"] for g in groups], *[g["xa"] for g in groups],
                    *[g["dxa"] for g in groups], *[g["dxa"] for g in groups], *[g["dxn"] for g in groups]))
    del keep
    return outs


def stn_fwd(src, flow, dims):
    """src [T,C] channels-last, flow [T,3] -> out [T,C]."""
    B, D, H, W = dims
    out = torch.empty_like(src)
    call("micf_stn_fwd", f32(src), f32(flow), f32(out), B, D, H, W, src.shape[1])
    return out


def stn_bwd(dout, src, flow, dims):
    B, D, H, W = dims
    dsrc = torch.zeros_like(src)
    dflow = torch.empty_like(flow)
    call("micf_stn_bwd", f32(dout), f32(src), f32(flow), f32(dsrc), f32(dflow), B, D, H, W, src.shape[1])
    return dsrc, dflow


# ----------------------------------------------------------------------------- stride == kernel convs
def patch_embed_fwd(vol, mod, w, bias, p):
    B, nmod, D, H, W = vol.shape
    E = w.shape[0]
    Dc, Hc, Wc = -(-D // p), -(-H // p), -(-W // p)
    y = _new(vol, B, Dc, Hc, Wc, E)
    call("micf_patch_embed_fwd", f32(vol), nmod, mod, f32(w), f32(bias), f32(y), B, D, H, W, E, p,
         cost=_cost(2 * y.numel() * p ** 3, y, w) if _lib.PROFILE is not None else None)
    return y


def patch_embed_bwd_weight(dy, vol, mod, dw, dbias, p):
    B, nmod, D, H, W = vol.shape
    E = dw.shape[0]
    call("micf_patch_embed_bwd_weight", f32(dy), f32(vol), nmod, mod, f32(dw), f32(dbias), B, D, H, W, E, p,
         cost=_cost(2 * dy.numel() * p ** 3, dy, dw))


def conv_down_fwd(x, w, bias):
    B, D, H, W, C = x.shape
    N = w.shape[0]
    y = _new(x, B, -(-D // 2), -(-H // 2), -(-W // 2), N)
    call("micf_conv_down_fwd", f32(x), f32(w), f32(bias), f32(y), B, D, H, W, C, N,
         cost=_cost(2 * y.numel() * 8 * C, x, w, y))
    return y


def conv_down_bwd_data(dy, w, xshape):
    B, D, H, W, C = xshape
    N = w.shape[0]
    dx = _new(dy, B, D, H, W, C)
    call("micf_conv_down_bwd_data", f32(dy), f32(w), f32(dx), B, D, H, W, C, N,
         cost=_cost(2 * dy.numel() * 8 * C, dy, w, dx))
    return dx


def conv_down_bwd_weight(dy, x, dw, dbias):
    B, D, H, W, C = x.shape
    N = dw.shape[0]
    call("micf_conv_down_bwd_weight", f32(dy), f32(x), f32(dw), f32(dbias), B, D, H, W, C, N,
         cost=_cost(2 * dy.numel() * 8 * C, dy, x, dw))


def conv_up_fwd(x, w, bias, k):
    B, D, H, W, C = x.shape
    N = w.shape[1]
    y = _new(x, B, D * k, H * k, W * k, N)
    call("micf_conv_up_fwd", f32(x), f32(w), f32(bias), f32(y), B, D, H, W, C, N, k,
         cost=_cost(2 * y.numel() * C, x, w, y))
    return y


# ----------------------------------------------------------------------------- reverse_patch_embedding + out_conv, composed
def head_tail_transposed_up(w_up):
    """w_up [Ci, Cm, P, P, P] as [Cm * P^3, Ci]: what the composition kernels read coalesced (one small launch)."""
    Ci = w_up.shape[0]
    src = w_up.reshape(Ci, -1)
    dst = _new(w_up, src.shape[1], Ci)
    WeightPrepPlan([(src, None, dst)]).launch()
    return dst


def head_tail_compose(w_up, b_up, w_out, w_up_t=None):
    Ci, Cm, P = w_up.shape[0], w_up.shape[1], w_up.shape[2]
    Co = w_out.shape[0]
    rows = (P + 2) ** 3 * Co
    wb, bf = _new(w_up, rows, Ci), _new(w_up, rows)
    call("micf_head_tail_compose", f32(w_up), f32(b_up), f32(w_out), f32(wb), f32(bf), Ci, Cm, Co, P, f32(w_up_t),
         cost=_cost(2 * rows * (Ci + 1) * Cm * 4, w_up, w_out, wb))
    return wb, bf


def head_tail_fused_supported(dims, Ci, Co, P):
    """The patch-matrix-free head tail (head_tail_fused.hip) covers this grid / mode?"""
    _, Dc, Hc, Wc = dims
    return bool(_lib.lib.micf_head_tail_fused_supported(Dc, Hc, Wc, Ci, Co, P, _dt()))


def head_tail_pack(wb, bf, b_out, P):
    """The bf16 operand packs of micf_head_tail_fwd_fused / _bwd_data_fused from the composed map (once per step)."""
    Ci = wb.shape[1]
    Co = b_out.shape[0]
    pf = torch.empty(_lib.lib.micf_head_tail_pack_bytes(Ci, 0) // 2, dtype=torch.bfloat16, device=wb.device)
    pq = torch.empty(_lib.lib.micf_head_tail_pack_bytes(Ci, 1) // 2, dtype=torch.bfloat16, device=wb.device)
    call("micf_head_tail_pack", f32(wb), f32(bf), f32(b_out), ptr(pf), ptr(pq), Ci, Co, P, cost=_cost(0, wb, pf, pq))
    return pf, pq


def head_tail_fwd_fused(x, pack_fwd, dims, Co, P):
    B, Dc, Hc, Wc = dims
    Ci = x.shape[-1]
    y = _new(x, B, Co, Dc * P, Hc * P, Wc * P)
    call("micf_head_tail_fwd_fused", f32(x), ptr(pack_fwd), f32(y), B, Dc, Hc, Wc, Ci, Co, P,
         cost=_cost(2 * x.shape[0] * (P + 2) ** 3 * Co * Ci, x, y, tag=f"{x.shape[0]}x{Ci}"))
    return y


def head_tail_fwd_fused_sw(x, pack_fwd, out, count, coords, dims, Co, P):
    """Sliding-window form of head_tail_fwd_fused: x holds len(coords) windows' coarse features; their logits are ADDED into the volume
    accumulator `out` (VB, Co, VD, VH, VW) at coords[i] = (sample, z0, y0, x0) (int32 device tensor [n, 4]) and `count` (VB, VD, VH, VW)
    += 1 there -- no prediction tensor, no patch matrix."""
    n, Dc, Hc, Wc = dims
    Ci = x.shape[-1]
    VB, _, VD, VH, VW = out.shape
    assert coords.dtype == torch.int32 and coords.is_contiguous() and tuple(coords.shape) == (n, 4)
    call("micf_head_tail_fwd_fused_sw", f32(x), ptr(pack_fwd), f32(out), f32(count), ptr(coords), n, Dc, Hc, Wc, Ci, Co, P, VB, VD, VH, VW,
         cost=_cost(2 * x.shape[0] * (P + 2) ** 3 * Co * Ci, x, tag=f"{x.shape[0]}x{Ci}"))


def head_tail_fwd_loss_fused(x, pack_fwd, dims, Co, P, target):
    """head_tail_fwd_fused + MDiceLoss's forward sums in the logits store (micf_head_tail_fwd_loss_fused).  target: float one-hot
    planes (B, Co, 4Dc, 4Hc, 4Wc) or the uint8 class map (B, 4Dc, 4Hc, 4Wc).  -> (logits, loss [1], sums [Co*4] float64)."""
    B, Dc, Hc, Wc = dims
    Ci = x.shape[-1]
    y = _new(x, B, Co, Dc * P, Hc * P, Wc * P)
    nparts = int(_lib.lib.micf_head_tail_loss_parts(B, Dc, Hc, Wc))
    part = _new(x, nparts, 32)
    sums = _new(x, Co * 4, dtype=torch.float64)
    loss = _new(x, 1)
    call("micf_head_tail_fwd_loss_fused", f32(x), ptr(pack_fwd), f32(y), ptr(target), 1 if target.dtype == torch.uint8 else 0, f32(part),
         ptr(sums), f32(loss), B, Dc, Hc, Wc, Ci, Co, P,
         cost=_cost(2 * x.shape[0] * (P + 2) ** 3 * Co * Ci + 40 * y.numel(), x, y, target, tag=f"{x.shape[0]}x{Ci}"))
    return y, loss, sums


def head_tail_bwd_data_fused(dy, pack_bwd, dims, Ci, P):
    B, Dc, Hc, Wc = dims
    Co = dy.shape[1]
    dx = _new(dy, B * Dc * Hc * Wc, Ci)
    call("micf_head_tail_bwd_data_fused", f32(dy), ptr(pack_bwd), f32(dx), B, Dc, Hc, Wc, Ci, Co, P,
         cost=_cost(2 * dx.shape[0] * (P + 2) ** 3 * Co * Ci, dy, dx, tag=f"{dx.shape[0]}x{Ci}"))
    return dx


def head_tail_bwd_weight_fused(dy, x, dims, P):
    """(dwb, dbf) of the composed map with dy gathered on the fly (no U), or None when this grid is not covered."""
    B, Dc, Hc, Wc = dims
    Ci, Co = x.shape[-1], dy.shape[1]
    need = _lib.lib.micf_head_tail_bwd_weight_fused_workspace(B, Dc, Hc, Wc, Ci)
    if need == 0 or not head_tail_fused_supported(dims, Ci, Co, P):
        return None
    rows = (P + 2) ** 3 * Co
    dwb, dbf = _new(x, rows, Ci), _new(x, rows)
    ws = scratch(x.device, need)
    call("micf_head_tail_bwd_weight_fused", f32(dy), f32(x), f32(dwb), f32(dbf), f32(ws), ws.numel(), B, Dc, Hc, Wc, Ci, Co, P,
         cost=_cost(2 * x.shape[0] * rows * Ci, dy, x, dwb, tag=f"{x.shape[0]}x{Ci}"))
    return dwb, dbf


def head_tail_col2im(t, b_out, dims, P):
    B, Dc, Hc, Wc = dims
    Co = b_out.shape[0]
    y = _new(t, B, Co, Dc * P, Hc * P, Wc * P)
    call("micf_head_tail_col2im", f32(t), f32(b_out), f32(y), B, Dc, Hc, Wc, Co, P, cost=_cost(t.numel(), t, y))
    return y


def head_tail_col2im_sw(t, b_out, out, count, coords, dims, P):
    """Sliding-window form of head_tail_col2im: t holds len(coords) windows; their logits are ADDED into the volume accumulator `out`
    [VB, Co, VD, VH, VW] at coords [n, 4] int32 (device: {sample, z0, y0, x0}) and `count` [VB, VD, VH, VW] += 1 there.  ONE launch."""
    n, Dc, Hc, Wc = dims
    VB, Co, VD, VH, VW = out.shape
    if coords.dtype != torch.int32 or coords.shape != (n, 4):
        raise TypeError("coords must be an int32 [n, 4] device tensor")
    call("micf_head_tail_col2im_sw", f32(t), f32(b_out), f32(out), f32(count), ptr(coords), n, Dc, Hc, Wc, Co, P, VB, VD, VH, VW,
         cost=_cost(t.numel(), t))


def head_tail_im2col(dy, dims, P):
    B, Dc, Hc, Wc = dims
    Co = dy.shape[1]
    u = _new(dy, B * Dc * Hc * Wc, (P + 2) ** 3 * Co)
    call("micf_head_tail_im2col", f32(dy), f32(u), B, Dc, Hc, Wc, Co, P, cost=_cost(0, dy, u))
    return u


def head_tail_decompose(dwb, dbf, w_up, b_up, w_out, dw_up, db_up, dw_out, db_out, w_up_t=None):
    Ci, Cm, P = w_up.shape[0], w_up.shape[1], w_up.shape[2]
    Co = w_out.shape[0]
    call("micf_head_tail_decompose", f32(dwb), f32(dbf), f32(w_up), f32(b_up), f32(w_out), f32(dw_up), f32(db_up),
         f32(dw_out), f32(db_out), Ci, Cm, Co, P, f32(w_up_t),
         cost=_cost(4 * dwb.numel() * Cm * 27 // (P + 2) ** 3 * P ** 3, dwb, w_up, w_out))


def conv_up_bwd_data(dy, w, xshape, k):
    B, D, H, W, C = xshape
    N = w.shape[1]
    dx = _new(dy, B, D, H, W, C)
    call("micf_conv_up_bwd_data", f32(dy), f32(w), f32(dx), B, D, H, W, C, N, k,
         cost=_cost(2 * dy.numel() * C, dy, w, dx))
    return dx


def conv_up_bwd_weight(dy, x, dw, dbias, k):
    B, D, H, W, C = x.shape
    N = dw.shape[1]
    call("micf_conv_up_bwd_weight", f32(dy), f32(x), f32(dw), f32(dbias), B, D, H, W, C, N, k,
         cost=_cost(2 * dy.numel() * C, dy, x, dw))


def space_to_depth(x, dims, C, k, batch_stride=0, offset=0, out=None):
    """Fine channels-last tensor (voxel stride C; `x` may be any contiguous fp32 buffer: `offset` elements to its first voxel,
    `batch_stride` elements between samples, 0 = dense) -> [B * ceil(D/k) * ceil(H/k) * ceil(W/k), C * k^3] patch rows."""
    B, D, H, W = dims
    rows = B * (-(-D // k)) * (-(-H // k)) * (-(-W // k))
    a = out if out is not None else _new(x, rows, C * k ** 3)
    call("micf_space_to_depth", ctypes.c_void_p(x.data_ptr() + 4 * offset), f32(a), B, D, H, W, C, k, batch_stride,
         cost=_cost(0, a, a))
    return a


def depth_to_space(a, dims, C, k, bias=None, add=None):
    """Inverse scatter of space_to_depth: [rows, C * k^3] -> channels-last [B, D, H, W, C] (+ bias[C]); every voxel is written.
    add: a [B, D, H, W, C] tensor summed in (a second gradient of the same tensor: the skip connection's)."""
    B, D, H, W = dims
    y = _new(a, B, D, H, W, C)
    if add is not None:
        assert add.numel() == y.numel() and add.is_contiguous()
        call("micf_depth_to_space_add", f32(a), f32(bias), f32(add), f32(y), B, D, H, W, C, k, cost=_cost(0, a, add, y))
        return y
    call("micf_depth_to_space", f32(a), f32(bias), f32(y), B, D, H, W, C, k, cost=_cost(0, a, y))
    return y


def colsum_(x2d, out):
    """out[N] += column sums of x2d [M, N]."""
    call("micf_colsum", f32(x2d), f32(out), x2d.shape[0], x2d.shape[1], cost=_cost(0, x2d))


# ----------------------------------------------------------------------------- pad / crop / resize
def pad3d(x, dims, pdims):
    B, D, H, W = dims
    Dp, Hp, Wp = pdims
    C = x.shape[-1]
    y = _new(x, B * Dp * Hp * Wp, C)
    call("micf_pad3d", f32(x), f32(y), B, D, H, W, Dp, Hp, Wp, C)
    return y


def crop3d(xp, dims, pdims, out=None, accumulate=False):
    B, D, H, W = dims
    Dp, Hp, Wp = pdims
    C = xp.shape[-1]
    y = out if out is not None else _new(xp, B * D * H * W, C)
    call("micf_crop3d", f32(xp), f32(y), B, D, H, W, Dp, Hp, Wp, C, 1 if accumulate else 0)
    return y


def resize_trilinear_fwd(x, size):
    B, D, H, W, C = x.shape
    y = _new(x, B, size[0], size[1], size[2], C)
    call("micf_resize_trilinear_fwd", f32(x), f32(y), B, D, H, W, size[0], size[1], size[2], C)
    return y


def resize_trilinear_bwd(dy, xshape):
    B, D, H, W, C = xshape
    dx = _new(dy, B, D, H, W, C)
    call("micf_resize_trilinear_bwd", f32(dy), f32(dx), B, D, H, W, dy.shape[1], dy.shape[2], dy.shape[3], C)
    return dx


# ----------------------------------------------------------------------------- loss / metrics / optimiser
def dice_bce_fwd(logits, target):
    B, K = logits.shape[:2]
    V = logits[0, 0].numel()
    sums = _new(logits, K * 4, dtype=torch.float64)
    loss = _new(logits, 1)
    if target.dtype == torch.uint8:        # class map [B, ...] instead of one-hot planes [B, K, ...]
        call("micf_dice_bce_label_fwd", f32(logits), ptr(target), ptr(sums), f32(loss), B, K, V,
             cost=_cost(30 * logits.numel(), logits, target))
    else:
        call("micf_dice_bce_fwd", f32(logits), f32(target), ptr(sums), f32(loss), B, K, V,
             cost=_cost(30 * logits.numel(), logits, target))
    return loss, sums


def dice_bce_bwd(logits, target, sums, grad_out):
    B, K = logits.shape[:2]
    V = logits[0, 0].numel()
    dz = torch.empty_like(logits)
    name = "micf_dice_bce_label_bwd" if target.dtype == torch.uint8 else "micf_dice_bce_bwd"
    call(name, f32(logits), ptr(target), ptr(sums), f32(grad_out), f32(dz), B, K, V,
         cost=_cost(30 * logits.numel(), logits, target, dz))
    return dz


def argmax_meandice(logits, label=None, want_mask=True):
    """logits [B,K,...] fp32; label uint8 class map [B,...] or None.  Returns (mask uint8, meandice 0-d double or None)."""
    B, K = logits.shape[:2]
    V = logits[0, 0].numel()
    mask = torch.empty((B,) + tuple(logits.shape[2:]), dtype=torch.uint8, device=logits.device) if want_mask else None
    counts = torch.empty(3 * K, dtype=torch.int64, device=logits.device) if label is not None else None
    out = torch.empty(1, dtype=torch.float64, device=logits.device) if label is not None else None
    if label is not None and label.dtype != torch.uint8:
        raise TypeError("label map must be uint8")
    call("micf_argmax_meandice", f32(logits), ptr(label), ptr(mask), ptr(counts), ptr(out), B, K, V)
    return mask, out


def dice_metric(logits, target):
    """MDiceLoss(_Val).metric on the device: [B, K] thresholded-sigmoid Dice per (sample, class); target = one-hot float planes
    [B, K, ...] or a uint8 class map [B, ...]."""
    B, K = logits.shape[:2]
    V = logits[0, 0].numel()
    sums = _new(logits, B * K * 3, dtype=torch.float64)
    out = _new(logits, B, K)
    is_label = target.dtype == torch.uint8
    if not is_label and target.dtype != torch.float32:
        raise TypeError("target must be float32 one-hot planes or a uint8 class map")
    call("micf_dice_metric", f32(logits), ptr(target), 1 if is_label else 0, ptr(sums), f32(out), B, K, V)
    return out


def adam_state(device):
    """{int64 step; double lr} on the device, zero-initialised (16 bytes)."""
    return torch.zeros(2, dtype=torch.int64, device=device)


def adam_tick(state, base_lr, eta_min, t_max):
    call("micf_adam_tick", ptr(state), float(base_lr), float(eta_min), int(t_max))


def adam_step(p, g, m, v, state, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, mirror=None):
    """mirror: optional bfloat16 tensor of p's size that receives bf16(p) after the update (same pass)."""
    if mirror is not None and (mirror.dtype != torch.bfloat16 or mirror.numel() != p.numel()):
        raise _lib.MicfError("the Adam mirror must be a bfloat16 tensor of the parameter buffer's size")
    call("micf_adam_step", f32(p), f32(g), f32(m), f32(v), p.numel(), ptr(state), float(beta1), float(beta2), float(eps),
         float(grad_scale), ptr(mirror),
         cost=_cost(12 * p.numel(), p, p, g, m, m, v, v))


class HipWireOps:
    """The three streaming kernels of the bf16 gradient wire (dist.WireExchange: bf16 on the links, fp32 in the sums) as C-ABI
    launches on the current stream -- the product counterpart of dist.TorchWireOps."""

    @staticmethod
    def pack(src, dst):
        if dst.dtype != torch.bfloat16 or dst.numel() % 8 or dst.numel() < src.numel():
            raise _lib.MicfError("gradient wire: the bf16 buffer must hold the slice padded to a multiple of 8")
        call("micf_grad_wire_pack", f32(src), src.numel(), ptr(dst), dst.numel(), cost=_cost(0, src, dst))

    @staticmethod
    def sum_shards(recv, ranks, out):
        if recv.numel() != ranks * out.numel() or out.numel() % 8:
            raise _lib.MicfError("gradient wire: recv must be [ranks, shard] with shard a multiple of 8")
        call("micf_grad_wire_sum", ptr(recv), int(ranks), out.numel(), ptr(out), cost=_cost(ranks * out.numel(), recv, out))

    @staticmethod
    def unpack(src, dst):
        if src.numel() < dst.numel():
            raise _lib.MicfError("gradient wire: source shorter than the slice")
        call("micf_grad_wire_unpack", ptr(src), f32(dst), dst.numel(), cost=_cost(0, dst, dst))


# ----------------------------------------------------------------------------- fused window-local transformer block
_DT = {"fp32": 0, "bf16": 1}

FWD_W = (("ln1_g", "norm1.weight"), ("ln1_b", "norm1.bias"), ("bq", "{a}.q.bias"), ("bkv", "{a}.kv.bias"), ("bp", "{a}.proj.bias"),
         ("ln2_g", "norm2.weight"), ("ln2_b", "norm2.bias"), ("b1", "mlp.fc1.bias"), ("b2", "mlp.fc2.bias"))
FWD_WM = (("wq", "{a}.q.weight"), ("wkv", "{a}.kv.weight"), ("wp", "{a}.proj.weight"), ("w1", "mlp.fc1.weight"), ("w2", "mlp.fc2.weight"))
BWD_W = (("ln1_g", "norm1.weight"), ("ln2_g", "norm2.weight"))
BWD_WT = (("wqt", "{a}.q.weight"), ("wkvt", "{a}.kv.weight"), ("wpt", "{a}.proj.weight"), ("w1t", "mlp.fc1.weight"),
          ("w2t", "mlp.fc2.weight"))


def block_tile_tokens(dims, C, heads, hidden, backward=False):
    """Tokens per workgroup tile of the fused block kernels for this shape; 0 = not handled (use the per-op path)."""
    B, D, H, W = dims
    return int(_lib.lib.micf_block_tile_tokens(B, D, H, W, C, heads, hidden, 1 if backward else 0))


def blocked16(w):
    """The K16-blocked order of a [R, Cn] matrix (R, Cn multiples of 16) in which the block kernels stream their bf16 shadow
    weights: [R/16][Cn/16][16][16], returned with w's shape (torch restatement of micf_weight_prep_grouped's bf16 = 2 layout)."""
    R, Cn = w.shape
    return w.reshape(R // 16, 16, Cn // 16, 16).permute(0, 2, 1, 3).contiguous().reshape(R, Cn)


class WeightPrepPlan:
    """ctypes item array of (src [r, c] fp32, dst [r, c] | None, dst_t [c, r] | None) triples for micf_weight_prep_grouped (the
    outputs of one item are both fp32 or both bfloat16); reusable while the tensors keep their addresses.  blocked: the
    outputs are written K16-blocked (the block kernels' shadow weights; rows and cols multiples of 16) instead of row-major."""

    def __init__(self, triples, blocked=False):
        self.triples = list(triples)
        self.n = len(self.triples)
        self.arr = (_lib.WeightPrepItem * max(self.n, 1))()
        self.nbytes = 0
        for it, (src, dst, dst_t) in zip(self.arr, self.triples):
            outs = [t for t in (dst, dst_t) if t is not None]
            assert outs and len({t.dtype for t in outs}) == 1 and outs[0].dtype in (torch.float32, torch.bfloat16)
            it.src, it.dst, it.dst_t, it.rows, it.cols = f32(src), ptr(dst), ptr(dst_t), src.shape[0], src.shape[1]
            it.bf16 = (2 if blocked else 1) if outs[0].dtype == torch.bfloat16 else (3 if blocked else 0)
            self.nbytes += src.numel() * (4 + sum(t.element_size() for t in outs))

    def launch(self):
        if self.n:
            call("micf_weight_prep_grouped", ctypes.cast(self.arr, ctypes.c_void_p), self.n,
                 cost=(self.nbytes, 0) if _lib.PROFILE is not None else None)


class Conv3PrepPlan:
    """ctypes item array for micf_conv3_weight_prep_grouped: (w [N, Cin, 3,3,3], fwd layout | None, bwd layout | None)."""

    def __init__(self, triples):
        self.triples = list(triples)
        self.n = len(self.triples)
        self.arr = (_lib.Conv3PrepItem * max(self.n, 1))()
        for it, (w, fwd, bwd) in zip(self.arr, self.triples):
            it.w, it.fwd, it.bwd, it.N, it.Cin = f32(w), f32(fwd), f32(bwd), w.shape[0], w.shape[1]

    def launch(self):
        if self.n:
            call("micf_conv3_weight_prep_grouped", ctypes.cast(self.arr, ctypes.c_void_p), self.n)


def conv3_prepared_like(w):
    """Empty (fwd, bwd) re-layout buffers for a few-output-channel conv weight, or (None, None) when the direct kernels do not apply."""
    N, Cin = w.shape[0], w.shape[1]
    nf, nb = _lib.lib.micf_conv3_fwd_workspace(N, Cin, 0), _lib.lib.micf_conv3_bwd_data_workspace(N, Cin, 0)
    mk = lambda n: torch.empty(n, dtype=torch.float32, device=w.device) if n > 0 else None
    return mk(nf), mk(nb)


# parameter attribute holding an engine-maintained shadow copy, per (arithmetic mode, direction): (name, transposed, dtype)
_SHADOW = {("fp32", False): ("_micf_w32", False, torch.float32), ("fp32", True): ("_micf_wt", True, torch.float32),
           ("bf16", False): ("_micf_w16", False, torch.bfloat16),
           ("bf16", True): ("_micf_wt16", True, torch.bfloat16)}


ENGINE_SHADOWS = False      # set by the engine around its step: the shadow copies parked on the parameters are current


def shadow_spec(backward):
    """(attribute name, transposed, dtype) of the K16-blocked weight copy the fused kernels stream in the current arithmetic mode."""
    return _SHADOW.get((compute_dtype(), bool(backward)))


def shadow_like(w, transposed, dtype):
    return torch.empty((w.shape[1], w.shape[0]) if transposed else tuple(w.shape), dtype=dtype, device=w.device)


PARAM_EPOCH = [0]           # bumped by TrainEngine after every optimiser step: its Adam kernel writes the parameters without
                            # touching torch's version counters, so the caches below would otherwise serve pre-step copies


def _inference_cache(w, key, build):
    """Shadow copies for forward passes WITHOUT an engine (validation, sliding-window inference under no_grad): built once per
    weight and kept on the tensor until torch code (version counter) or an engine step (PARAM_EPOCH) writes it.  A captured
    predictor graph (inference.GraphedPredictor) then replays without any per-window weight preparation.  None when autograd
    is recording."""
    if torch.is_grad_enabled():
        return None
    cache = w.__dict__.setdefault("_micf_cache", {})
    ent = cache.get(key)
    stamp = (w._version, PARAM_EPOCH[0])
    if ent is None or ent[0] != stamp:
        ent = cache[key] = (stamp, build())
    return ent[1]


def _conv_layouts(w):
    """(fwd, bwd) prepared layouts of an offset-conv weight: the engine's (refreshed per step), the inference cache's, or None."""
    f, b = (getattr(w, "_micf_c3f", None), getattr(w, "_micf_c3b", None)) if ENGINE_SHADOWS else (None, None)
    if f is None and b is None and w.dim() == 5 and w.shape[0] <= 16:
        def build():
            fb = conv3_prepared_like(w)
            if fb[0] is not None or fb[1] is not None:
                Conv3PrepPlan([(w, fb[0], fb[1])]).launch()
            return fb
        fb = _inference_cache(w, "conv3", build)
        if fb is not None:
            f, b = fb
    return f, b


def block_weights(P, attn, backward, only=None):
    """The five weight matrices a fused block kernel streams (or the fields named in `only`), in the layout of the current
    arithmetic mode.  Engine mode: the parameter carries the shadow copy (refreshed once per step).  Otherwise (module-level drop-in,
    validation, sliding-window inference) the copies live on the weight tensor, keyed by torch's version counter and PARAM_EPOCH:
    a weight that was written since (optimizer.step(), load_state_dict) is re-derived -- BOTH orientations (the forward's copy and
    the backward's transposed copy) by ONE grouped launch for all stale weights of the call, so a training step prepares every
    weight once, at its forward.  (A write through `.data` does not move the version counter: bump ops.PARAM_EPOCH[0] after one.)"""
    fields = BWD_WT if backward else FWD_WM
    if only is not None:
        fields = tuple(f for f in fields if f[0] in only)
    spec = shadow_spec(backward)
    if spec is None:
        return {field: P[key.format(a=attn)] for field, key in fields}
    fspec, bspec = shadow_spec(False), shadow_spec(True)
    out, todo = {}, []
    for field, key in fields:
        w = P[key.format(a=attn)]
        sh = getattr(w, spec[0], None) if ENGINE_SHADOWS else None
        if sh is None:
            cache = w.__dict__.setdefault("_micf_cache", {})
            stamp = (w._version, PARAM_EPOCH[0], compute_dtype())
            ent = cache.get("shadows")
            if ent is None or ent[0] != stamp:
                fwd = shadow_like(w, fspec[1], fspec[2]) if fspec is not None else None
                bwd = shadow_like(w, bspec[1], bspec[2])
                ent = cache["shadows"] = (stamp, fwd, bwd)
                todo.append((w, fwd, bwd))
            sh = ent[2] if backward else ent[1]
        out[field] = sh
    if todo:
        WeightPrepPlan(todo, blocked=True).launch()
    return out


def _block_cost(nb, fl, groups, T, C, hidden, self_passes, cross_passes):
    """Profiler cost of one fused block launch.  The 4th field is SURVEY.md 8(d)'s figure for it: with ideal whole-block fusion a
    self block moves its activation 2 times forward (R x, W x') and 3 times backward (R x, R dy, W dx), a cross block 3 / 5 times
    (+ R xa; + R xa, W dxa), each pass T * C * e bytes with e the element size of the arithmetic mode (bf16 = 2), plus the block's
    weights (3 C^2 + C^2 + 2 C hidden elements) once."""
    if _lib.PROFILE is None:
        return None
    e = 2 if _dt() else 4
    s8d = 0
    for gd in groups:
        cross = gd.get("cross", gd.get("kvsrc") is not None or gd.get("hid") is not None)
        s8d += (cross_passes if cross else self_passes) * T * C * e + (4 * C * C + 2 * C * hidden) * e
    return (nb, fl, f"{len(groups)}x{T}x{C}" if DETAIL else None, s8d)


def block_fuses_sampler(C, heads):
    """True when block_fwd can run a cross block's deformable sampling itself for this shape (group field `hid`)."""
    return bool(_lib.lib.micf_block_fuses_sampler(C, heads))


def block_saves_bf16(C, heads):
    """True when the fused block kernels store their saved tensors / weight-gradient operands as bf16 for this shape in the current
    arithmetic mode (include/micformer_hip.h micf_block_saves_bf16)."""
    return bool(_lib.lib.micf_block_saves_bf16(C, heads, _dt()))


def block_recomputes_h(C, heads):
    """True when block_bwd rebuilds the fc1 pre-activation from xn2 for this shape: block_fwd then does not store it
    (include/micformer_hip.h micf_block_recomputes_h: opt-in through the option "block_recompute_h", a memory switch)."""
    return bool(_lib.lib.micf_block_recomputes_h(C, heads))


def block_fwd(groups, dims, C, heads, eps, scale, persist_probe=None, save=True):
    """groups: 1 or 2 dicts {x [T,C], kvsrc [T,C] | None, P {state_dict-style name: tensor}, attn 'self_attn' | 'cross_attn',
    s1, s2 [B] | None, want_xn bool}.  ONE launch.  Returns per group a dict of the tensors saved for backward (+ 'y').
    A group may carry "next_ln": (gamma, beta, zero16 | None) -- the LayerNorm the NEXT block applies to this block's output (a cross
    block's norm1), written by the same launch (block_fuses_sampler shapes only): its dict then has "nln" = (y_normed, mean, rstd);
    zero16: a [T, 16] fp32 tensor the launch clears.
    persist_probe: (repeats, int32[2] device tensor) -- the measurement probe micf_block_fwd_persistent_probe instead.
    save=False (no backward will follow): the inference form where the kernels have one (block_fuses_sampler shapes: everything but
    the few-token decomposition) -- only 'y' is written, every other entry of the returned dicts is None."""
    B, D, H, W = dims
    T = B * D * H * W
    hidden = groups[0]["P"]["mlp.fc1.weight"].shape[0]
    arr = (_lib.BlockFwdGroup * 2)()
    outs = []
    h_dtype = torch.bfloat16 if _dt() else torch.float32       # the saved fc1 pre-activation: half width in bf16 mode
    st16 = block_saves_bf16(C, heads)                          # ... and everything only matrix cores / the attention backward re-read
    sd = torch.bfloat16 if st16 else torch.float32
    no_h = block_recomputes_h(C, heads)                        # the backward rebuilds h from xn2: not stored
    keep = []            # temporary shadow weights must outlive the launch: the next group's outputs must not reuse them
    nb = fl = 0
    y_all = _new(groups[0]["x"], len(groups) * T, C)    # the groups' outputs are the halves of ONE buffer (functional.JoinFn: no copy)
    for gi, (it, gd) in enumerate(zip(arr, groups)):
        x, P, a = gd["x"], gd["P"], gd["attn"]
        fused_sampler = gd.get("hid") is not None       # cross block that samples its K/V source itself: {hid, samp_src} given
        cross = gd.get("kvsrc") is not None or fused_sampler
        if not save and block_fuses_sampler(C, heads) and persist_probe is None:
            # inference form: nothing saved, every pointer but y NULL
            o = dict.fromkeys(("q", "kv", "o", "x1", "xn2", "h", "g", "stats", "xn", "kvs16", "flow", "xs32"))
            o["y"] = y_all[gi * T:(gi + 1) * T]
        else:
            o = {"y": y_all[gi * T:(gi + 1) * T], "q": _new(x, T, C, dtype=sd), "kv": _new(x, T, 2 * C, dtype=sd),
                 "o": _new(x, T, C, dtype=sd), "x1": _new(x, T, C), "xn2": _new(x, T, C, dtype=sd),
                 "h": None if no_h else _new(x, T, hidden, dtype=h_dtype), "g": _new(x, T, hidden, dtype=sd), "stats": _new(x, 4, T),
                 # (bf16 storage: the q weight gradient pairs a bf16 dq with a bf16 xn, so the kernel always writes its own copy)
                 "xn": _new(x, T, C, dtype=sd) if (gd.get("want_xn", True) or st16) else None,
                 "kvs16": _new(x, T, C, dtype=sd) if (st16 and cross) else None,
                 "flow": _new(x, T, 3) if fused_sampler else None,
                 "xs32": _new(x, T, C) if (fused_sampler and not st16) else None}
        it.x, it.kvsrc, it.s1, it.s2 = f32(x), f32(gd.get("kvsrc")), f32(gd.get("s1")), f32(gd.get("s2"))
        if fused_sampler:
            it.hid, it.samp_src = f32(gd["hid"]), f32(gd["samp_src"])
            it.ln16_g, it.ln16_b, it.w1c = (f32(P[k]) for k in ("conv_offset.1.norm.weight", "conv_offset.1.norm.bias", "conv_offset.3.weight"))
        nl = gd.get("next_ln")
        if nl is not None:
            o["nln"] = (_new(x, T, C), _new(x, T), _new(x, T))
            it.nln_g, it.nln_b, it.zero16 = f32(nl[0]), f32(nl[1]), f32(nl[2])
            it.nln_y, it.nln_mean, it.nln_rstd = (f32(t) for t in o["nln"])
            nb += 4 * T * (C + 2)
        for field, key in FWD_W:
            setattr(it, field, f32(P[key.format(a=a)]))
        wts = block_weights(P, a, backward=False)
        keep.append(wts)
        for field, wt in wts.items():
            setattr(it, field, ptr(wt))
        for k, v in o.items():
            if k != "nln":
                setattr(it, k, ptr(v))
        outs.append(o)
        # bytes the launch moves: read x (+ kvsrc), write everything in `o`; the weights once
        nb += 4 * T * C * (2 if cross else 1) + (64 * T if fused_sampler else 0) \
            + sum(v.numel() * v.element_size() for k, v in o.items() if v is not None and k != "nln") \
            + 12 * C * C * wt.element_size()
        fl += 2 * T * 12 * C * C + 4 * T * C * 8
    if persist_probe is not None:
        call("micf_block_fwd_persistent_probe", ctypes.cast(arr, ctypes.c_void_p), len(groups), B, D, H, W, C, heads, hidden, float(eps),
             float(scale), _dt(), int(persist_probe[0]), ptr(persist_probe[1]))
        return outs
    call("micf_block_fwd", ctypes.cast(arr, ctypes.c_void_p), len(groups), B, D, H, W, C, heads, hidden, float(eps), float(scale),
         _dt_attn(), cost=_block_cost(nb, fl, groups, T, C, hidden, 2, 3))
    return outs


def block_bwd(groups, dims, C, heads, scale):
    """groups: 1 or 2 dicts {dy, x, x1, stats, q, kv, h, P, attn, s1, s2, cross bool, want_copy bool, want_ln1 bool}.  ONE launch.
    Returns per group {dx, dxs | None, dx1, dh, dq, dkv, ln1_part | None, ln2_part, tiles, dx1_copy | None}."""
    B, D, H, W = dims
    T = B * D * H * W
    hidden = groups[0]["P"]["mlp.fc1.weight"].shape[0]
    tm = block_tile_tokens(dims, C, heads, hidden, backward=True)
    tiles = (T + tm - 1) // tm
    arr = (_lib.BlockBwdGroup * 2)()
    outs = []
    keep = []            # temporaries (transposed weights) must outlive the launch: the next group's outputs must not reuse them
    nb = fl = 0
    st16 = block_saves_bf16(C, heads)
    sd = torch.bfloat16 if st16 else torch.float32
    dx_all = _new(groups[0]["dy"], len(groups) * T, C)          # halves of one buffer: the input gradients join without a copy
    copy_all = _new(groups[0]["dy"], len(groups) * T, C) if all(gd.get("want_copy") for gd in groups) else None
    for gi, (it, gd) in enumerate(zip(arr, groups)):
        dy, P, a, cross = gd["dy"], gd["P"], gd["attn"], gd["cross"]
        o = {"dx": dx_all[gi * T:(gi + 1) * T], "dxs": _new(dy, T, C) if cross else None, "dx1": _new(dy, T, C, dtype=sd),
             "dh": _new(dy, T, hidden, dtype=sd), "dq": _new(dy, T, C, dtype=sd), "dkv": _new(dy, T, 2 * C, dtype=sd),
             "ln2_part": _new(dy, tiles, 2 * C), "ln1_part": None if cross else _new(dy, tiles, 2 * C),
             "dx1_copy": (copy_all[gi * T:(gi + 1) * T] if copy_all is not None else _new(dy, T, C)) if gd.get("want_copy") else None,
             "dy16": _new(dy, T, C, dtype=sd) if st16 else None}
        for k in ("dy", "x", "x1", "stats", "s1", "s2"):
            setattr(it, k, f32(gd.get(k)))
        h = gd.get("h")
        if (h is not None and h.dtype != (torch.bfloat16 if _dt() else torch.float32)) or gd["q"].dtype != sd or gd["kv"].dtype != sd \
                or (h is None and gd["xn2"].dtype != sd):
            raise _lib.MicfError("block_bwd: the saved tensors were written in another arithmetic mode")
        it.h, it.q, it.kv = ptr(h), ptr(gd["q"]), ptr(gd["kv"])
        pre = gd.get("pre")
        if pre is not None:                             # the producing LayerNorm backward as the kernel's prologue (bf16 storage, self)
            if not st16 or cross:
                raise _lib.MicfError("block_bwd: the LayerNorm-backward prologue needs a self block with bf16 storage")
            o["pre_part"] = _new(dy, tiles, 2 * C)
            it.pre_d, it.pre_x, it.pre_mean, it.pre_rstd, it.pre_g, it.pre_part = (f32(pre["d"]), f32(pre["x"]), f32(pre["mean"]),
                                                                                  f32(pre["rstd"]), f32(pre["gamma"]), f32(o["pre_part"]))
            nb += 4 * T * C * 2
        if h is None:                                   # recompute: xn2, the FORWARD orientation of fc1's weight, its bias
            w1 = block_weights(P, a, backward=False, only=("w1",))["w1"]
            keep.append(w1)
            it.xn2, it.w1, it.b1 = ptr(gd["xn2"]), ptr(w1), f32(P["mlp.fc1.bias"])
        for field, key in BWD_W:
            setattr(it, field, f32(P[key.format(a=a)]))
        wts = block_weights(P, a, backward=True)
        keep.append(wts)
        for field, wt in wts.items():
            setattr(it, field, ptr(wt))
        for k, v in o.items():
            setattr(it, k, ptr(v))
        # bytes the launch moves: read dy, x1, q, kv, h [+ x: self]; write everything in `o`; the weights once
        nb += 4 * T * C * (2 if cross else 3) + sum(gd[k].numel() * gd[k].element_size() for k in ("q", "kv", "h" if h is not None else "xn2")) \
            + sum(v.numel() * v.element_size() for v in o.values() if v is not None) + 12 * C * C * wt.element_size()
        o["tiles"] = tiles
        outs.append(o)
        fl += 2 * T * 12 * C * C + 8 * T * C * 8
    call("micf_block_bwd", ctypes.cast(arr, ctypes.c_void_p), len(groups), B, D, H, W, C, heads, hidden, float(scale),
         _dt(), cost=_block_cost(nb, fl, groups, T, C, hidden, 3, 5))
    del keep
    return outs


# ----------------------------------------------------------------------------- sliding window (batched) / input pipeline
def _coords(chunk):
    arr = (ctypes.c_int32 * (4 * len(chunk)))()
    for i, (b, z, y, x) in enumerate(chunk):
        arr[4 * i:4 * i + 4] = [b, z, y, x]
    return arr


def sw_window_batch(vol, chunk, roi):
    """vol [B, C, D, H, W]; chunk: list of (b, z0, y0, x0) -> [n, C, rd, rh, rw] crops, ONE launch."""
    B, C, D, H, W = vol.shape
    rd, rh, rw = roi
    win = _new(vol, len(chunk), C, rd, rh, rw)
    arr = _coords(chunk)
    call("micf_sw_window_batch", f32(vol), f32(win), ctypes.cast(arr, ctypes.c_void_p), len(chunk), B, C, D, H, W, rd, rh, rw)
    return win


def sw_accumulate_batch(pred, out, count, chunk):
    """pred [n, K, rd, rh, rw] added into out [B, K, D, H, W] at the windows of `chunk`, count [B, D, H, W] += 1 there; ONE launch."""
    B, K, D, H, W = out.shape
    rd, rh, rw = pred.shape[2:]
    arr = _coords(chunk)
    call("micf_sw_accumulate_batch", f32(pred), f32(out), f32(count), ctypes.cast(arr, ctypes.c_void_p), len(chunk), B, K, D, H, W,
         rd, rh, rw)


def intensity_stats(vol):
    """{sum, sum of squares, count} of the non-zero voxels per (sample, channel): [B * Cm * 3] float64."""
    B, Cm = vol.shape[:2]
    if vol.dtype not in (torch.float16, torch.float32):
        raise TypeError("raw volume must be float16 or float32")
    sums = torch.empty(B * Cm * 3, dtype=torch.float64, device=vol.device)
    call("micf_intensity_stats", ptr(vol), 1 if vol.dtype == torch.float16 else 0, ptr(sums), B, Cm, vol[0, 0].numel())
    return sums


def patch_rows_prepared(vol, sums, params, k):
    """Raw volume [B, Cm <= 2, D, H, W] (fp16 / fp32) -> the [tokens, k^3] patch-row matrices of its modalities with the input
    tail (flips, non-zero normalisation, scale / shift) applied on the fly.  ONE launch."""
    B, Cm, D, H, W = vol.shape
    rows = B * (-(-D // k)) * (-(-H // k)) * (-(-W // k))
    both = torch.empty((Cm * rows, k ** 3), dtype=torch.float32, device=vol.device)      # (halves of one buffer: one GEMM for both)
    outs = [both[m * rows:(m + 1) * rows] for m in range(Cm)]
    call("micf_patch_rows_prepared", ptr(vol), 1 if vol.dtype == torch.float16 else 0, ptr(sums), f32(params), f32(outs[0]),
         f32(outs[1]) if Cm > 1 else None, B, Cm, D, H, W, k, cost=_cost(0, vol, *outs))
    return outs


def flip_labels(label_map, params):
    """label_out = label_in[flips of params] (uint8 [B, D, H, W]); params None: the map itself."""
    if params is None:
        return label_map
    B, D, H, W = label_map.shape
    out = torch.empty_like(label_map)
    dummy = torch.zeros(3 * B, dtype=torch.float64, device=label_map.device)
    call("micf_input_prepare", ptr(label_map), 0, ptr(dummy), f32(params), None, ptr(label_map), ptr(out), B, 1, D, H, W)
    return out


def input_prepare(vol, label_map=None, params=None):
    """Input-pipeline tail on the device.  vol [B, Cm, D, H, W] float16 / float32 raw intensities, label_map uint8 [B, D, H, W] or
    None, params [B, 5] float32 {flip D, flip H, flip W, scale f, shift o} or None (validation).  -> (float32 volume, label map)."""
    B, Cm, D, H, W = vol.shape
    if vol.dtype not in (torch.float16, torch.float32):
        raise TypeError("raw volume must be float16 or float32")
    half = 1 if vol.dtype == torch.float16 else 0
    sums = torch.empty(B * Cm * 3, dtype=torch.float64, device=vol.device)
    call("micf_intensity_stats", ptr(vol), half, ptr(sums), B, Cm, D * H * W)
    out = torch.empty((B, Cm, D, H, W), dtype=torch.float32, device=vol.device)
    lab_out = torch.empty_like(label_map) if label_map is not None else None
    if label_map is not None and label_map.dtype != torch.uint8:
        raise TypeError("label map must be uint8")
    call("micf_input_prepare", ptr(vol), half, ptr(sums), f32(params), f32(out), ptr(label_map), ptr(lab_out), B, Cm, D, H, W)
    return out, lab_out


# ----------------------------------------------------------------------------- step plumbing
def zero_(t):
    """In-place zero fill as one memset node (optimizer.zero_grad() over the flat gradient buffer)."""
    call("micf_zero", ptr(t), t.numel() * t.element_size())
    return t


def drop_path_rng(device, seed):
    """{uint64 seed; uint64 counter} on the device for micf_drop_path_draw."""
    return torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)


def drop_path_draw(rng, keep, batch):
    """keep [n] fp32 keep-probabilities -> [n, batch] per-sample scales (mask / keep); advances the device counter."""
    n = keep.numel()
    out = _new(keep, n, batch)
    call("micf_drop_path_draw", ptr(rng), f32(keep), f32(out), n, batch)
    return out
