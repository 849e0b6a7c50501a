"""torch.autograd.Function wrappers: one per fused unit of the hot path; forward and backward are sequences of C-ABI
launches (micformer_amd.ops).  Activations are channels-last (B, D, H, W, C) fp32; parameters keep the reference's
state_dict layout.  Parameter gradients are accumulated by the kernels into zero-filled buffers and handed to autograd.
"""
import torch

from . import ops
from ._lib import block_region

_zl = torch.zeros_like


def _in_block(fn):
    """Decorator: the launches of this forward / backward are transformer-block kernels (profiler region tag)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with block_region():
            return fn(*a, **k)
    return wrapped


def effective_window(dims, window):
    """get_window_size (MS.py:135-145): a window dim clamps to the volume dim when dim <= window."""
    return tuple(d if d <= w else w for d, w in zip(dims, window))


def _padded(dims, ws):
    return tuple(d + (-d) % w for d, w in zip(dims, ws))


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- deferred weight gradients (engine mode)
# Nothing downstream of a backward pass consumes the parameter gradients until the optimiser step.  In engine mode (the
# kernels accumulate straight into the flat gradient buffer) the weight gradients of the linear layers are therefore not
# launched where autograd reaches them: the (output-gradient, input) pair is queued and TrainEngine flushes the queue once
# after backward through micf_linear_bwd_weight_grouped -- a few chip-filling launches instead of two small ones per
# layer, and the data-gradient chain (the critical path) gets shorter.  The queue holds references, so the caching
# allocator cannot hand the queued buffers to anyone else before the flush.
DEFER_WGRAD = False
DEFER_MAX_TOKENS = 1 << 17          # larger layers launch immediately (their operands are still hot in L2 / MALL)
_DEFERRED = []
_DEFERRED_LN = []                   # (partials, blocks, C, dgamma, dbeta) of LayerNorm backward calls


_QUEUED_DW = set()                  # destinations already in the queue: the grouped kernel owns each dW element exclusively


def _lin_wgrad(defer, dy, a, dw, db, dp_scale=None, rows_per_sample=0):
    if defer and DEFER_WGRAD and dy.shape[0] <= DEFER_MAX_TOKENS and ops.wgrad_groupable(dy, a, dp_scale, rows_per_sample) \
            and dw.data_ptr() not in _QUEUED_DW:          # (a layer applied twice in one step launches its second use at once)
        _QUEUED_DW.add(dw.data_ptr())
        _DEFERRED.append((dy, a, dw, db, dp_scale, rows_per_sample))
    else:
        ops.linear_bwd_weight(dy, a, dw, db, dp_scale=dp_scale, rows_per_sample=rows_per_sample)


def take_deferred():
    """Hand the queued (not yet launched) weight gradients and LayerNorm partials to the caller (TrainEngine's data-parallel step)."""
    items, ln = list(_DEFERRED), list(_DEFERRED_LN)
    _DEFERRED.clear()
    _DEFERRED_LN.clear()
    _QUEUED_DW.clear()
    return items, ln


def drop_deferred():
    """Forget anything still queued (TrainEngine calls this at the start of a step: entries left behind by a backward that
    raised must not be flushed into the next step's gradients)."""
    _DEFERRED.clear()
    _DEFERRED_LN.clear()
    _QUEUED_DW.clear()


def _ln_defer(on):
    """The list LayerNorm backward should queue its parameter-gradient partials in, or None (accumulate immediately)."""
    return _DEFERRED_LN if (on and DEFER_WGRAD) else None


@_in_block
def flush_wgrad():
    """Launch every queued weight gradient on the current stream (which must be ordered after their producers)."""
    if _DEFERRED_LN:
        ln = list(_DEFERRED_LN)
        _DEFERRED_LN.clear()
        ops.layernorm_bwd_finish(ln)
    if _DEFERRED:
        items = list(_DEFERRED)
        _DEFERRED.clear()
        _QUEUED_DW.clear()
        ops.linear_bwd_weight_grouped(items)


def _targets(params):
    """Engine mode: a parameter may carry `_micf_grad`, a view of the flat gradient buffer.  The backward kernels then
    accumulate straight into it (they atomically add anyway) and autograd gets None -- no zero-fill, no `grad += g` pass."""
    return tuple(getattr(p, "_micf_grad", None) for p in params)


def _grad_buf(target, like):
    return target if target is not None else torch.zeros_like(like)


def _ret(target, buf):
    return None if target is not None else buf


# ============================================================================= LayerNorm
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim of [..., C] (optionally over cat[x, x2], MS.py:1033-1034)."""

    @staticmethod
    def forward(ctx, x, x2, gamma, beta, eps):
        x = _c(x)
        C1 = x.shape[-1]
        x2f = _c(x2).reshape(-1, x2.shape[-1]) if x2 is not None else None
        y, mean, rstd = ops.layernorm_fwd(x.reshape(-1, C1), gamma, beta, eps, x2f)
        ctx.save_for_backward(x, x2 if x2 is None else _c(x2), gamma, mean, rstd)
        ctx.tg = _targets((gamma, beta))
        return y.reshape(x.shape[:-1] + (gamma.numel(),))

    @staticmethod
    def backward(ctx, dy):
        x, x2, gamma, mean, rstd = ctx.saved_tensors
        dy = _c(dy).reshape(-1, gamma.numel())
        dg, db = _grad_buf(ctx.tg[0], gamma), _grad_buf(ctx.tg[1], gamma)
        x2f = x2.reshape(-1, x2.shape[-1]) if x2 is not None else None
        r = ops.layernorm_bwd(dy, x.reshape(-1, x.shape[-1]), mean, rstd, gamma, dg, db, x2f,
                              defer=_ln_defer(ctx.tg[0] is not None and ctx.tg[1] is not None))
        dg, db = _ret(ctx.tg[0], dg), _ret(ctx.tg[1], db)
        if x2 is None:
            return r.reshape(x.shape), None, dg, db, None
        return r[0].reshape(x.shape), r[1].reshape(x2.shape), dg, db, None


# ============================================================================= Linear (plain / on a concatenation)
class LinearFn(torch.autograd.Function):
    """F.linear on [..., K] (optionally on cat[a, a2] without materialising it: concat_back_dim, MS.py:1027-1030)."""

    @staticmethod
    def forward(ctx, a, a2, w, b):
        a = _c(a)
        a2 = _c(a2) if a2 is not None else None
        y = ops.linear_fwd(a.reshape(-1, a.shape[-1]), w, b, a2.reshape(-1, a2.shape[-1]) if a2 is not None else None)
        ctx.save_for_backward(a, a2, w)
        ctx.has_bias = b is not None
        ctx.tg = _targets((w, b))
        return y.reshape(a.shape[:-1] + (w.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        a, a2, w = ctx.saved_tensors
        dy = _c(dy).reshape(-1, w.shape[0])
        k1 = a.shape[-1]
        af = a.reshape(-1, k1)
        a2f = a2.reshape(-1, a2.shape[-1]) if a2 is not None else None
        dw = _grad_buf(ctx.tg[0], w)
        db = (ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)) if ctx.has_bias else None
        ops.linear_bwd_weight(dy, af, dw, db, a2f)
        r = ops.linear_bwd_data(dy, w, k1=k1)
        dw, db = _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db)
        if a2 is None:
            return r.reshape(a.shape), None, dw, db
        return r[0].reshape(a.shape), r[1].reshape(a2.shape), dw, db


# ============================================================================= shared pieces of the two block types
def _mlp_fwd(x1f, dims, P, s2, eps):
    """part2 (MS.py:403-404/501-502 + :424/:522): x1 + s2 * fc2(GELU(fc1(LN2(x1)))).  Returns y, saved."""
    B, D, H, W = dims
    rps = D * H * W
    xn2, m2, r2 = ops.layernorm_fwd(x1f, P["norm2.weight"], P["norm2.bias"], eps)
    g, h = ops.linear_fwd(xn2, P["mlp.fc1.weight"], P["mlp.fc1.bias"], act=1, want_pre=True)
    y = ops.linear_fwd(g, P["mlp.fc2.weight"], P["mlp.fc2.bias"], resid=x1f, dp_scale=s2, rows_per_sample=rps)
    return y, (xn2, m2, r2, h, g)


def _mlp_bwd(dy, x1f, saved, dims, P, G, s2, side=False):
    """Returns dx1 = dy + LN2'(...) and accumulates the part2 parameter gradients into G."""
    B, D, H, W = dims
    rps = D * H * W
    xn2, m2, r2, h, g = saved      # g = GELU(h) is kept (HBM is plentiful) so the fc2 weight gradient is a plain GEMM
    _lin_wgrad(side, dy, g, G["mlp.fc2.weight"], G["mlp.fc2.bias"], s2, rps)
    dh = ops.linear_bwd_data(dy, P["mlp.fc2.weight"], dp_scale=s2, rows_per_sample=rps, pre_act=h)
    _lin_wgrad(side, dh, xn2, G["mlp.fc1.weight"], G["mlp.fc1.bias"])
    dxn2 = ops.linear_bwd_data(dh, P["mlp.fc1.weight"])
    return ops.layernorm_bwd(dxn2, x1f, m2, r2, P["norm2.weight"], G["norm2.weight"], G["norm2.bias"], add=dy, defer=_ln_defer(side))


SELF_KEYS = ("norm1.weight", "norm1.bias", "self_attn.q.weight", "self_attn.q.bias", "self_attn.kv.weight",
             "self_attn.kv.bias", "self_attn.proj.weight", "self_attn.proj.bias", "norm2.weight", "norm2.bias",
             "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")
CROSS_KEYS = ("norm1.weight", "norm1.bias", "cross_attn.q.weight", "cross_attn.q.bias", "cross_attn.kv.weight",
              "cross_attn.kv.bias", "cross_attn.proj.weight", "cross_attn.proj.bias", "conv_offset.0.weight",
              "conv_offset.0.bias", "conv_offset.1.norm.weight", "conv_offset.1.norm.bias", "conv_offset.3.weight",
              "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")


def _packed_qkv(P, G):
    """If q.weight | kv.weight and q.bias | kv.bias (and their gradient targets) sit back to back in memory -- TrainEngine lays
    the flat buffers out that way -- return ([3C, C] weight view, [3C] bias view, gradient views); else None."""
    wq, wkv, bq, bkv = (P[k] for k in ("self_attn.q.weight", "self_attn.kv.weight", "self_attn.q.bias", "self_attn.kv.bias"))
    gs = [G.get(k) for k in ("self_attn.q.weight", "self_attn.kv.weight", "self_attn.q.bias", "self_attn.kv.bias")]
    if any(g is None for g in gs):
        return None
    C = wq.shape[0]
    def adj(a, b):
        return a.data_ptr() + 4 * a.numel() == b.data_ptr()
    if not (adj(wq, wkv) and adj(bq, bkv) and adj(gs[0], gs[1]) and adj(gs[2], gs[3])):
        return None
    return (wq.as_strided((3 * C, C), (C, 1)), bq.as_strided((3 * C,), (1,)),
            gs[0].as_strided((3 * C, C), (C, 1)), gs[2].as_strided((3 * C,), (1,)))


# ============================================================================= TransformerBlock3D (MS.py:430-524)
class SelfBlockFn(torch.autograd.Function):
    @staticmethod
    @_in_block
    def forward(ctx, x, s1, s2, heads, window, eps, *params):
        P = dict(zip(SELF_KEYS, params))
        x = _c(x)
        B, D, H, W, C = x.shape
        dims = (B, D, H, W)
        ws = effective_window((D, H, W), window)
        pd = _padded((D, H, W), ws)
        padded = pd != (D, H, W)
        pdims = (B,) + pd
        rps = D * H * W
        scale = (C // heads) ** -0.5
        xf = x.reshape(-1, C)
        xn, m1, r1 = ops.layernorm_fwd(xf, P["norm1.weight"], P["norm1.bias"], eps)
        xnp = ops.pad3d(xn, dims, pd) if padded else xn             # F.pad AFTER the norm (MS.py:477-483)
        ctx.tg = _targets(params)
        fused = _packed_qkv(P, dict(zip(SELF_KEYS, ctx.tg)))
        if fused is not None:
            # engine mode: q.weight | kv.weight (and the biases) are adjacent in the flat buffer -> ONE [3C, C] projection
            q = ops.linear_fwd(xnp, fused[0], fused[1])             # packed [T, 3C] = [q | k | v]
            kv = q
            o = ops.window_attn_fwd_qkv(q, pdims, heads, ws, scale)
        else:
            q = ops.linear_fwd(xnp, P["self_attn.q.weight"], P["self_attn.q.bias"])
            kv = ops.linear_fwd(xnp, P["self_attn.kv.weight"], P["self_attn.kv.bias"])
            o = ops.window_attn_fwd(q, kv, pdims, heads, ws, scale)
        if padded:
            o = ops.crop3d(o, dims, pd)                              # proj is per token: crop before it (MS.py:497-498)
        x1 = ops.linear_fwd(o, P["self_attn.proj.weight"], P["self_attn.proj.bias"], resid=xf, dp_scale=s1,
                            rows_per_sample=rps)
        y, mlp_saved = _mlp_fwd(x1, dims, P, s2, eps)
        ctx.save_for_backward(xf, m1, r1, xnp, q, kv, o, x1, s1, s2, *mlp_saved, *params)
        ctx.meta = (dims, ws, pd, padded, heads, scale, fused is not None)
        return y.reshape(x.shape)

    @staticmethod
    @_in_block
    def backward(ctx, dy):
        sv = ctx.saved_tensors
        xf, m1, r1, xnp, q, kv, o, x1, s1, s2 = sv[:10]
        mlp_saved = sv[10:15]
        params = sv[15:]
        P = dict(zip(SELF_KEYS, params))
        G = {k: _grad_buf(t, v) for (k, v), t in zip(P.items(), ctx.tg)}
        dims, ws, pd, padded, heads, scale, packed = ctx.meta
        B, D, H, W = dims
        pdims = (B,) + pd
        rps = D * H * W
        C = xf.shape[1]
        dy = _c(dy).reshape(-1, C)
        side = all(t is not None for t in ctx.tg)      # engine mode: gradients land in the flat buffer, nobody waits for them
        dx1 = _mlp_bwd(dy, x1, mlp_saved, dims, P, G, s2, side)
        _lin_wgrad(side, dx1, o, G["self_attn.proj.weight"], G["self_attn.proj.bias"], s1, rps)
        do = ops.linear_bwd_data(dx1, P["self_attn.proj.weight"], dp_scale=s1, rows_per_sample=rps)
        if padded:
            do = ops.pad3d(do, dims, pd)
        if packed:
            wqkv, _, gw, gb = _packed_qkv(P, G)
            dqkv = ops.window_attn_bwd_qkv(q, do, pdims, heads, ws, scale)
            _lin_wgrad(side, dqkv, xnp, gw, gb)
            dxn = ops.linear_bwd_data(dqkv, wqkv)
        else:
            dq, dkv = ops.window_attn_bwd(q, kv, do, pdims, heads, ws, scale)
            _lin_wgrad(side, dq, xnp, G["self_attn.q.weight"], G["self_attn.q.bias"])
            _lin_wgrad(side, dkv, xnp, G["self_attn.kv.weight"], G["self_attn.kv.bias"])
            dxn = ops.linear_bwd_data(dq, P["self_attn.q.weight"])
            ops.linear_bwd_data(dkv, P["self_attn.kv.weight"], out=dxn, accumulate=True)
        if padded:
            dxn = ops.crop3d(dxn, dims, pd)
        dx = ops.layernorm_bwd(dxn, xf, m1, r1, P["norm1.weight"], G["norm1.weight"], G["norm1.bias"], add=dx1, defer=_ln_defer(side))
        return (dx.reshape(B, D, H, W, C), None, None, None, None, None) + \
            tuple(_ret(t, G[k]) for k, t in zip(SELF_KEYS, ctx.tg))


# ============================================================================= CrossTransformerBlock3D (MS.py:277-426)
class CrossBlockFn(torch.autograd.Function):
    @staticmethod
    @_in_block
    def forward(ctx, x, xa, s1, s2, heads, window, eps, *params):
        P = dict(zip(CROSS_KEYS, params))
        x, xa = _c(x), _c(xa)
        B, D, H, W, C = x.shape
        dims = (B, D, H, W)
        ws = effective_window((D, H, W), window)
        pd = _padded((D, H, W), ws)
        padded = pd != (D, H, W)
        pdims = (B,) + pd
        rps = D * H * W
        scale = (C // heads) ** -0.5
        xf, xaf = x.reshape(-1, C), xa.reshape(-1, C)
        xn, m1, r1 = ops.layernorm_fwd(xf, P["norm1.weight"], P["norm1.bias"], eps)   # only x is normed (MS.py:343)
        xnp = ops.pad3d(xn, dims, pd) if padded else xn
        xap = ops.pad3d(xaf, dims, pd) if padded else xaf
        hid = ops.conv3_fwd(xnp, P["conv_offset.0.weight"], P["conv_offset.0.bias"], pdims, x2=xap)
        w1 = P["conv_offset.3.weight"]
        flow, xs = ops.offset_sample_fwd(hid, P["conv_offset.1.norm.weight"], P["conv_offset.1.norm.bias"], w1, xap,
                                         pdims, eps)
        q = ops.linear_fwd(xnp, P["cross_attn.q.weight"], P["cross_attn.q.bias"])
        kv = ops.linear_fwd(xs, P["cross_attn.kv.weight"], P["cross_attn.kv.bias"])
        o = ops.window_attn_fwd(q, kv, pdims, heads, ws, scale)
        if padded:
            o = ops.crop3d(o, dims, pd)
        x1 = ops.linear_fwd(o, P["cross_attn.proj.weight"], P["cross_attn.proj.bias"], resid=xf, dp_scale=s1,
                            rows_per_sample=rps)
        y, mlp_saved = _mlp_fwd(x1, dims, P, s2, eps)
        ctx.save_for_backward(xf, m1, r1, xnp, xap, hid, flow, xs, q, kv, o, x1, s1, s2, *mlp_saved, *params)
        ctx.meta = (dims, ws, pd, padded, heads, scale, eps)
        ctx.tg = _targets(params)
        return y.reshape(x.shape)

    @staticmethod
    @_in_block
    def backward(ctx, dy):
        sv = ctx.saved_tensors
        xf, m1, r1, xnp, xap, hid, flow, xs, q, kv, o, x1, s1, s2 = sv[:14]
        mlp_saved = sv[14:19]
        params = sv[19:]
        P = dict(zip(CROSS_KEYS, params))
        G = {k: _grad_buf(t, v) for (k, v), t in zip(P.items(), ctx.tg)}
        dims, ws, pd, padded, heads, scale, eps = ctx.meta
        B, D, H, W = dims
        pdims = (B,) + pd
        rps = D * H * W
        C = xf.shape[1]
        dy = _c(dy).reshape(-1, C)
        side = all(t is not None for t in ctx.tg)
        dx1 = _mlp_bwd(dy, x1, mlp_saved, dims, P, G, s2, side)
        _lin_wgrad(side, dx1, o, G["cross_attn.proj.weight"], G["cross_attn.proj.bias"], s1, rps)
        do = ops.linear_bwd_data(dx1, P["cross_attn.proj.weight"], dp_scale=s1, rows_per_sample=rps)
        if padded:
            do = ops.pad3d(do, dims, pd)
        dq, dkv = ops.window_attn_bwd(q, kv, do, pdims, heads, ws, scale)
        _lin_wgrad(side, dq, xnp, G["cross_attn.q.weight"], G["cross_attn.q.bias"])
        _lin_wgrad(side, dkv, xs, G["cross_attn.kv.weight"], G["cross_attn.kv.bias"])
        dxnp = ops.linear_bwd_data(dq, P["cross_attn.q.weight"])
        dxs = ops.linear_bwd_data(dkv, P["cross_attn.kv.weight"])
        dxap = _zl(xap)                                            # atomic scatter target of the sampler
        dhid = ops.offset_sample_bwd(dxs, hid, P["conv_offset.1.norm.weight"], P["conv_offset.1.norm.bias"],
                                     P["conv_offset.3.weight"], xap, flow, dxap, G["conv_offset.1.norm.weight"],
                                     G["conv_offset.1.norm.bias"], G["conv_offset.3.weight"], pdims, eps)
        ops.conv3_bwd_weight(dhid, xnp, G["conv_offset.0.weight"], G["conv_offset.0.bias"], pdims, x2=xap)
        ops.conv3_bwd_data(dhid, P["conv_offset.0.weight"], pdims, C, C, dx1=dxnp, dx2=dxap, acc1=True, acc2=True)
        if padded:
            dxn = ops.crop3d(dxnp, dims, pd)
            dxa = ops.crop3d(dxap, dims, pd)
        else:
            dxn, dxa = dxnp, dxap
        dx = ops.layernorm_bwd(dxn, xf, m1, r1, P["norm1.weight"], G["norm1.weight"], G["norm1.bias"], add=dx1, defer=_ln_defer(side))
        return (dx.reshape(B, D, H, W, C), dxa.reshape(B, D, H, W, C), None, None, None, None, None) + \
            tuple(_ret(t, G[k]) for k, t in zip(CROSS_KEYS, ctx.tg))


# ============================================================================= patch embed / merging / expand / head
class PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed3D (MS.py:860-878) on modality `mod` of vol [B, nmod, D, H, W] -> (B, D', H', W', E) channels-last."""

    @staticmethod
    def forward(ctx, vol, mod, w, b, p):
        vol = _c(vol)
        y = ops.patch_embed_fwd(vol, mod, w, b, p)
        ctx.save_for_backward(vol, w)
        ctx.meta = (mod, p)
        ctx.tg = _targets((w, b))
        return y

    @staticmethod
    def backward(ctx, dy):
        vol, w = ctx.saved_tensors
        mod, p = ctx.meta
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        ops.patch_embed_bwd_weight(_c(dy), vol, mod, dw, db, p)
        return None, None, _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db), None   # the input volume is data: no gradient (train.py:177-185)


class ConvDownFn(torch.autograd.Function):
    """Conv3d(C->N, k=s=2) of PatchMerging on channels-last x (MS.py:548-557)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _c(x)
        ctx.save_for_backward(x, w)
        ctx.tg = _targets((w, b))
        return ops.conv_down_fwd(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        ops.conv_down_bwd_weight(dy, x, dw, db)
        return ops.conv_down_bwd_data(dy, w, tuple(x.shape)), _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db)


class ConvUpFn(torch.autograd.Function):
    """ConvTranspose3d(C->N, k=s) on channels-last x (PatchExpand MS.py:575-577; reverse_patch_embedding MS.py:1037)."""

    @staticmethod
    def forward(ctx, x, w, b, k):
        x = _c(x)
        ctx.save_for_backward(x, w)
        ctx.k = k
        ctx.tg = _targets((w, b))
        return ops.conv_up_fwd(x, w, b, k)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[1], dtype=w.dtype, device=w.device)
        ops.conv_up_bwd_weight(dy, x, dw, db, ctx.k)
        return ops.conv_up_bwd_data(dy, w, tuple(x.shape), ctx.k), _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db), None


class OutConvFn(torch.autograd.Function):
    """Head.out_conv: Conv3d(E/2 -> classes, 3, padding=1) from the channels-last feature to NCDHW logits (MS.py:1053)."""

    @staticmethod
    def forward(ctx, feat, w, b):
        feat = _c(feat)
        B, D, H, W, C = feat.shape
        ctx.save_for_backward(feat, w)
        ctx.tg = _targets((w, b))
        return ops.conv3_fwd(feat.reshape(-1, C), w, b, (B, D, H, W), ncdhw_out=True)

    @staticmethod
    def backward(ctx, dy):
        feat, w = ctx.saved_tensors
        B, D, H, W, C = feat.shape
        dy = _c(dy)
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        f2 = feat.reshape(-1, C)
        ops.conv3_bwd_weight(dy, f2, dw, db, (B, D, H, W), ncdhw=True)
        dx, _ = ops.conv3_bwd_data(dy, w, (B, D, H, W), C, 0, ncdhw=True)
        return dx.reshape(feat.shape), _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db)


class HeadTailFn(torch.autograd.Function):
    """reverse_patch_embedding (ConvTranspose3d 2E -> E/2, k = s = P; MS.py:1037) + Head.out_conv (Conv3d E/2 -> classes, 3,
    padding=1; MS.py:1053) composed into one linear map on the coarse grid (csrc/head_tail.hip): channels-last coarse
    feature x (B, Dc, Hc, Wc, 2E) -> NCDHW logits (B, classes, P*Dc, P*Hc, P*Wc).  The E/2-channel fine feature is never built."""

    @staticmethod
    def forward(ctx, x, w_up, b_up, w_out, b_out):
        x = _c(x)
        B, Dc, Hc, Wc, Ci = x.shape
        P = w_up.shape[2]
        wb, bf = ops.head_tail_compose(w_up, b_up, w_out)
        xf = x.reshape(-1, Ci)
        t = ops.linear_fwd(xf, wb, bf)
        y = ops.head_tail_col2im(t, b_out, (B, Dc, Hc, Wc), P)
        ctx.save_for_backward(xf, wb, w_up, b_up, w_out)
        ctx.dims = (B, Dc, Hc, Wc)
        ctx.tg = _targets((w_up, b_up, w_out, b_out))
        return y

    @staticmethod
    def backward(ctx, dy):
        xf, wb, w_up, b_up, w_out = ctx.saved_tensors
        B, Dc, Hc, Wc = ctx.dims
        P = w_up.shape[2]
        u = ops.head_tail_im2col(_c(dy), ctx.dims, P)
        dx = ops.linear_bwd_data(u, wb)
        dwb, dbf = torch.zeros_like(wb), torch.zeros(wb.shape[0], dtype=wb.dtype, device=wb.device)
        ops.linear_bwd_weight(u, xf, dwb, dbf)
        grads = [_grad_buf(t, p) for t, p in zip(ctx.tg, (w_up, b_up, w_out, w_out.new_empty(w_out.shape[0])))]
        ops.head_tail_decompose(dwb, dbf, w_up, b_up, w_out, *grads)
        return (dx.reshape(B, Dc, Hc, Wc, -1),) + tuple(_ret(t, g) for t, g in zip(ctx.tg, grads))


class ResizeTrilinearFn(torch.autograd.Function):
    """F.interpolate(mode='trilinear', align_corners=True) on channels-last volumes (MS.py:1018-1025)."""

    @staticmethod
    def forward(ctx, x, size):
        x = _c(x)
        ctx.xshape = tuple(x.shape)
        return ops.resize_trilinear_fwd(x, size)

    @staticmethod
    def backward(ctx, dy):
        return ops.resize_trilinear_bwd(_c(dy), ctx.xshape), None


class DiceBCEFn(torch.autograd.Function):
    """MDiceLoss.forward (dice.py:158-166)."""

    @staticmethod
    def forward(ctx, logits, target):
        logits, target = _c(logits), _c(target)
        loss, sums = ops.dice_bce_fwd(logits, target)
        ctx.save_for_backward(logits, target, sums)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, target, sums = ctx.saved_tensors
        return ops.dice_bce_bwd(logits, target, sums, _c(g).reshape(1)), None
