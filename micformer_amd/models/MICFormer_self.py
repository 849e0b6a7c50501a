"""MI355X-native MicFormer: the nn.Module surface of the reference's ``MicFormer/models/MICFormer_self.py`` (MS.py)
with every forward/backward executed by hand-written HIP kernels through the C-ABI of libmicformer_hip.so.

Drop-in contract:
  * same importable names, constructor signatures and ``state_dict`` keys/shapes as MS.py, so reference checkpoints
    load with ``strict=True`` and ``train_mmwhs_noPad.py`` can ``from models.MICFormer_self import Head`` unchanged;
  * ``Head(...)(x)`` takes float32 ``(B, 2, D, H, W)`` (ch0 = CT "moving", ch1 = MR "fixed") and returns logits
    ``(B, num_classes, D', H', W')``, differentiable w.r.t. every parameter;
  * ``torch.nn`` layers are used ONLY as parameter containers (their ATen forwards are never called); there is no
    CPU / PyTorch fallback -- CPU tensors raise.

Extra keyword-only constructor arguments (``depths``, ``num_heads``, ``drop_path_rate``) expose what the reference
hard-codes, so BASELINE's "tiny" config can be built.
"""
from functools import reduce
from operator import mul

import torch
import torch.nn as nn

from .. import functional as Fn
from .STN import Re_SpatialTransformer, SpatialTransformer

__all__ = ["Head", "MicFormer", "BasicLayer", "BasicLayerUp", "CrossTransformerBlock3D", "TransformerBlock3D",
           "CrossWindowAttention3D", "WindowAttention3D", "PatchEmbed3D", "PatchMerging", "PatchExpand", "Mlp",
           "LayerNormProxy", "DropPath", "window_partition", "window_reverse", "get_window_size"]


# ----------------------------------------------------------------------------- layout helpers (views only)
def window_partition(x, window_size):
    """(B, D, H, W, C) -> (B*nW, wd*wh*ww, C); same contract as MS.py:37-50.  Not used by the kernels (index math)."""
    B, D, H, W, C = x.shape
    wd, wh, ww = window_size
    x = x.reshape(B, D // wd, wd, H // wh, wh, W // ww, ww, C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, reduce(mul, window_size), C)


def window_reverse(windows, window_size, B, D, H, W):
    """Inverse of window_partition; same contract as MS.py:117-132."""
    wd, wh, ww = window_size
    x = windows.reshape(B, D // wd, H // wh, W // ww, wd, wh, ww, -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)


def get_window_size(x_size, window_size, shift_size=None):
    """MS.py:135-145: clamp each window dim to the volume dim when the volume is not larger."""
    return Fn.effective_window(tuple(x_size), tuple(window_size))


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath semantics, scale_by_keep=True; MS.py:5,320,467).

    The kernels take the per-sample scale (mask / keep_prob) as a [B] device vector and fuse it into the residual
    epilogue, so this module only draws that vector.
    """

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep

    def sample_scale(self, batch, device):
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        s = torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            s.div_(keep)
        return s

    def forward(self, x):
        s = self.sample_scale(x.shape[0], x.device)
        return x if s is None else x * s.reshape((-1,) + (1,) * (x.dim() - 1))

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:0.3f}"


class _NoDrop(nn.Identity):
    def sample_scale(self, batch, device):
        return None


def _drop_path(rate):
    return DropPath(rate) if rate > 0.0 else _NoDrop()


# ----------------------------------------------------------------------------- small modules
class _MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        from .. import ops
        x = x.contiguous()
        xf = x.reshape(-1, x.shape[-1])
        g, h = ops.linear_fwd(xf, w1, b1, act=1, want_pre=True)
        y = ops.linear_fwd(g, w2, b2)
        ctx.save_for_backward(xf, h, w1, w2)
        return y.reshape(x.shape[:-1] + (w2.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        from .. import ops
        xf, h, w1, w2 = ctx.saved_tensors
        dy2 = dy.contiguous().reshape(-1, w2.shape[0])
        dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        db1 = torch.zeros(w1.shape[0], device=w1.device)
        db2 = torch.zeros(w2.shape[0], device=w2.device)
        ops.linear_bwd_weight(dy2, h, dw2, db2, a_gelu=True)
        dh = ops.linear_bwd_data(dy2, w2, pre_act=h)
        ops.linear_bwd_weight(dh, xf, dw1, db1)
        dx = ops.linear_bwd_data(dh, w1)
        return dx.reshape(dy.shape[:-1] + (w1.shape[1],)), dw1, db1, dw2, db2


class Mlp(nn.Module):
    """fc1 -> exact GELU -> fc2 (MS.py:16-34); dropout p must be 0 (it is everywhere in the reference)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU or drop != 0.:
            raise NotImplementedError("HIP Mlp implements exact-erf GELU with dropout 0 (the reference's configuration)")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return _MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)


class _WindowAttnFn(torch.autograd.Function):
    """Attention over EXPLICIT windows (nW, N, C) x (nW, N, C): the standalone module contract of MS.py:179-203."""

    @staticmethod
    def forward(ctx, xq, xkv, heads, wq, bq, wkv, bkv, wp, bp):
        from .. import ops
        nW, N, C = xq.shape
        if xkv.shape[1] != N or N > 8:
            raise NotImplementedError("HIP window attention: query/key windows of equal size <= 8 tokens")
        xqf, xkf = xq.contiguous().reshape(-1, C), xkv.contiguous().reshape(-1, C)
        scale = (C // heads) ** -0.5
        dims, ws = (nW, 1, 1, N), (1, 1, N)
        q = ops.linear_fwd(xqf, wq, bq)
        kv = ops.linear_fwd(xkf, wkv, bkv)
        o = ops.window_attn_fwd(q, kv, dims, heads, ws, scale)
        y = ops.linear_fwd(o, wp, bp)
        ctx.save_for_backward(xqf, xkf, q, kv, o, wq, wkv, wp)
        ctx.meta = (dims, ws, heads, scale, bq is not None, xq.shape)
        return y.reshape(nW, N, C)

    @staticmethod
    def backward(ctx, dy):
        from .. import ops
        xqf, xkf, q, kv, o, wq, wkv, wp = ctx.saved_tensors
        dims, ws, heads, scale, qkv_bias, shape = ctx.meta
        C = wq.shape[0]
        dyf = dy.contiguous().reshape(-1, C)
        dwq, dwkv, dwp = torch.zeros_like(wq), torch.zeros_like(wkv), torch.zeros_like(wp)
        dbq = torch.zeros(C, device=wq.device) if qkv_bias else None
        dbkv = torch.zeros(2 * C, device=wq.device) if qkv_bias else None
        dbp = torch.zeros(C, device=wq.device)
        ops.linear_bwd_weight(dyf, o, dwp, dbp)
        do = ops.linear_bwd_data(dyf, wp)
        dq, dkv = ops.window_attn_bwd(q, kv, do, dims, heads, ws, scale)
        ops.linear_bwd_weight(dq, xqf, dwq, dbq)
        ops.linear_bwd_weight(dkv, xkf, dwkv, dbkv)
        dxq = ops.linear_bwd_data(dq, wq)
        dxk = ops.linear_bwd_data(dkv, wkv)
        return dxq.reshape(shape), dxk.reshape(shape), None, dwq, dbq, dwkv, dbkv, dwp, dbp


class _WindowAttentionBase(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        if attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("HIP window attention implements dropout 0 (the reference's configuration)")
        if dim % num_heads != 0:
            raise ValueError("dim must be a multiple of num_heads")
        self.dim = dim
        self.window_size = window_size
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if qk_scale is not None and abs(qk_scale - head_dim ** -0.5) > 1e-12:
            raise NotImplementedError("qk_scale override is not implemented")
        self.scale = head_dim ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.softmax = nn.Softmax(dim=-1)

    def _run(self, x, xa):
        return _WindowAttnFn.apply(x, xa, self.num_heads, self.q.weight, self.q.bias, self.kv.weight, self.kv.bias,
                                   self.proj.weight, self.proj.bias)


class CrossWindowAttention3D(_WindowAttentionBase):
    """Q from x windows, K/V from xa windows (MS.py:148-203)."""

    def forward(self, x, xa):
        return self._run(x, xa)


class WindowAttention3D(_WindowAttentionBase):
    """Plain window MSA, no mask / bias table / shift (MS.py:206-261)."""

    def forward(self, x):
        return self._run(x, x)


class LayerNormProxy(nn.Module):
    """LayerNorm over the channel dim of a channels-first tensor (MS.py:263-273)."""

    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.dim = dim

    def forward(self, x):
        y = Fn.LayerNormFn.apply(x.permute(0, 2, 3, 4, 1), None, self.norm.weight, self.norm.bias, self.norm.eps)
        return y.permute(0, 4, 1, 2, 3)


# Run the two modalities' independent blocks on two HIP streams (engine / bench switch; off = the reference's serial order).
PARALLEL_MODALITIES = False
# ... and whether AUTOGRAD-tracked work may be put on the second stream (the two modalities' resampling convs, the per-op block
# path).  A segmented step capture (functional.StepSegmenter) turns it off: autograd replays an op's backward on its forward
# stream and orders it with events, and a gradient produced on the second stream in one captured segment but consumed in a later
# one (the skip connections) would be an event edge between two different graphs.  Forward-only side work (the head composition,
# weight preparation) is not affected.
FORK_AUTOGRAD_STREAMS = True
# backward flush points inside a stage: every SLOT_FLUSH_STRIDE-th depth slot (measured at base / 128^3 at the end of round 2:
# 1 -> 15.8 ms (the captured graph then replays side and main work serially), 2 -> 12.6, 3 -> 12.6, 4 -> 12.7, none -> 14.0:
# one flush in the middle of the six-slot stage, none inside the two-slot stages)
SLOT_FLUSH_STRIDE = 3
# Head: run reverse_patch_embedding + out_conv as their composition (off = the reference's two separate convolutions).
FUSE_HEAD_TAIL = True
# Both modalities' blocks of a depth slot in one fused launch (off: one fused launch per modality, on two streams when
# PARALLEL_MODALITIES is set).
import os as _os
PAIR_BLOCKS = True
_SIDE_STREAMS = {}
HEAD_WEIGHTS_AFTER = None           # TrainEngine.step_many: [event] behind which the head's weights are current (None: always)


def _side_stream(device):
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _SIDE_STREAMS[device] = s
    return s


def _block_scales(block, x):
    """The two DropPath draws of a block (part1, part2).  MicFormer pre-draws all of them in one batched RNG call per
    forward (`_predraw_drop_path`); a standalone block draws its own."""
    pre = block.__dict__.pop("_pending_scales", None)       # (always popped: a stale draw must not outlive its forward)
    if pre is not None and block.training:
        return pre
    B = x.shape[0]
    return block.drop_path.sample_scale(B, x.device), block.drop_path.sample_scale(B, x.device)


_KEY_PATHS = {}


def _block_params(block, keys):
    """The block's parameters by state_dict-style name, resolved through the modules' own tables at every call (a replaced
    submodule or parameter is seen) without walking named_parameters() -- 96 such walks were 6 ms of host time per forward."""
    out = []
    for k in keys:
        path = _KEY_PATHS.get(k)
        if path is None:
            path = _KEY_PATHS[k] = tuple(k.split("."))
        m = block
        for name in path[:-1]:
            m = m._modules[name]
        out.append(m._parameters[path[-1]])
    return out


class CrossTransformerBlock3D(nn.Module):
    """x <- x + DropPath(CrossAttn(LN(x), deformably re-sampled RAW xa)); x <- x + DropPath(MLP(LN(x)))  (MS.py:277-426)."""

    def __init__(self, dim, num_heads, window_size=(4, 4, 4), hidden_channels=16, kk=3, offset_range_factor=2,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, use_checkpoint=False):
        super().__init__()
        if hidden_channels != 16 or kk != 3 or offset_range_factor < 0 or not qkv_bias or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("HIP cross block implements the reference configuration: hidden 16, 3x3x3 offset "
                                      "conv, offset_range_factor >= 0 (no tanh), qkv_bias=True, nn.LayerNorm")
        self.dim = dim
        self.num_heads = num_heads
        self.window_size = tuple(window_size)
        self.mlp_ratio = mlp_ratio
        self.use_checkpoint = use_checkpoint
        self.hidden_channels = hidden_channels
        self.kk = kk
        self.offset_range_factor = offset_range_factor
        self.norm1 = norm_layer(dim)
        self.cross_attn = CrossWindowAttention3D(dim, window_size=self.window_size, num_heads=num_heads,
                                                 qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.conv_offset = nn.Sequential(
            nn.Conv3d(dim * 2, hidden_channels, kk, 1, kk // 2),
            LayerNormProxy(hidden_channels),
            nn.GELU(),
            nn.Conv3d(hidden_channels, 3, 1, 1, 0, bias=False))
        self.drop_path = _drop_path(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.stn = SpatialTransformer()

    def forward(self, x, xa):
        s1, s2 = _block_scales(self, x)
        return Fn.CrossBlockFn.apply(x, xa, s1, s2, self.num_heads, self.window_size, self.norm1.eps,
                                     *_block_params(self, Fn.CROSS_KEYS))


class TransformerBlock3D(nn.Module):
    """x <- x + DropPath(WindowMSA(LN(x))); x <- x + DropPath(MLP(LN(x)))  (MS.py:430-524).  No shift, no mask."""

    def __init__(self, dim, num_heads, window_size=(4, 4, 4), hidden_channels=16, kk=3, offset_range_factor=2,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, use_checkpoint=False):
        super().__init__()
        if not qkv_bias or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("HIP self block implements qkv_bias=True with nn.LayerNorm (reference configuration)")
        self.dim = dim
        self.num_heads = num_heads
        self.window_size = tuple(window_size)
        self.mlp_ratio = mlp_ratio
        self.use_checkpoint = use_checkpoint
        self.hidden_channels = hidden_channels
        self.kk = kk
        self.offset_range_factor = offset_range_factor
        self.norm1 = norm_layer(dim)
        self.self_attn = WindowAttention3D(dim, window_size=self.window_size, num_heads=num_heads, qkv_bias=qkv_bias,
                                           qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = _drop_path(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x):
        s1, s2 = _block_scales(self, x)
        return Fn.SelfBlockFn.apply(x, s1, s2, self.num_heads, self.window_size, self.norm1.eps,
                                    *_block_params(self, Fn.SELF_KEYS))


class PatchMerging(nn.Module):
    """Conv3d(C->2C, k=s=2) + LN on channels-last tokens (MS.py:527-561)."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.down_conv = nn.Conv3d(dim, 2 * dim, (2, 2, 2), stride=2, padding=0)
        self.norm = norm_layer(2 * dim)

    def forward(self, x):
        y = Fn.ConvDownFn.apply(x, self.down_conv.weight, self.down_conv.bias)
        return Fn.LayerNormFn.apply(y, None, self.norm.weight, self.norm.bias, self.norm.eps)


class PatchExpand(nn.Module):
    """ConvTranspose3d(C->C/2, k=s=2) + LN on channels-last tokens (MS.py:564-579)."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.up_conv = nn.ConvTranspose3d(dim, dim // 2, (2, 2, 2), stride=2, padding=0)
        self.norm = norm_layer(dim // 2)

    def forward(self, x):
        y = Fn.ConvUpFn.apply(x, self.up_conv.weight, self.up_conv.bias, 2)
        return Fn.LayerNormFn.apply(y, None, self.norm.weight, self.norm.bias, self.norm.eps)


JOINT_MODALITIES = True   # the shared per-token modules between the stages on [2B, ...] tensors


def _joint_ok(a, b):
    return JOINT_MODALITIES and torch.is_tensor(a) and torch.is_tensor(b) and a.is_cuda and a.shape == b.shape and a.dtype == b.dtype


def _both(fn, a, b):
    """(fn(a), fn(b)) for the two modalities -- the second call on the side stream when they may overlap (the resampling
    convs between the stages are 40-75 us launches each: a fork / join costs less than running them back to back)."""
    if not (PARALLEL_MODALITIES and FORK_AUTOGRAD_STREAMS and (a[0] if isinstance(a, tuple) else a).is_cuda):
        return fn(a), fn(b)
    main, side = torch.cuda.current_stream(), _side_stream((a[0] if isinstance(a, tuple) else a).device)
    side.wait_stream(main)
    ra = fn(a)
    with torch.cuda.stream(side):
        rb = fn(b)
    main.wait_stream(side)
    rb.record_stream(main)
    return ra, rb


class BasicLayer(nn.Module):
    """One stage: depth x {2 self blocks (one per modality), 2 cross blocks (both read the PRE-update pair)} + optional
    resampling module passed as `downsample` (PatchMerging in the encoder, PatchExpand in the decoder)  (MS.py:582-707)."""

    _resample_attr = "downsample"

    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.window_size = window_size
        self.depth = depth
        self.use_checkpoint = use_checkpoint

        def rate(i):
            return drop_path[i] if isinstance(drop_path, list) else drop_path

        def mk(cls):
            return nn.ModuleList([cls(dim=dim, num_heads=num_heads, window_size=window_size, mlp_ratio=mlp_ratio,
                                      qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                      drop_path=rate(i), norm_layer=norm_layer, use_checkpoint=use_checkpoint)
                                  for i in range(depth)])

        self.blocks1 = mk(CrossTransformerBlock3D)
        self.blocks2 = mk(CrossTransformerBlock3D)
        self.self_blocks1 = mk(TransformerBlock3D)
        self.self_blocks2 = mk(TransformerBlock3D)
        resample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None
        setattr(self, self._resample_attr, resample)

    def _pair_fusable(self, x, xa):
        """Both modalities' blocks of a depth slot in ONE fused launch (functional.SelfPairFn / CrossPairFn)?"""
        if not PAIR_BLOCKS or not x.is_cuda or x.shape != xa.shape or self.depth == 0:
            return False
        blk = self.self_blocks1[0]
        B, D, H, W, C = x.shape
        P = {"mlp.fc1.weight": blk.mlp.fc1.weight}
        return bool(Fn._fusable((B, D, H, W), C, blk.num_heads, Fn.effective_window((D, H, W), blk.window_size), P)) and \
            Fn._padded((D, H, W), (2, 2, 2)) == (D, H, W)

    def _forward_pairs(self, x, xa):
        grad_mode = torch.is_grad_enabled()            # (read HERE: it is always off inside Function.forward)
        for i in range(self.depth):
            if i and i % SLOT_FLUSH_STRIDE == 0 and (Fn.CTX.flush_points or Fn.CTX.defer_calls) and x.requires_grad:
                x, xa = Fn.FlushPointFn.apply(x, xa)   # backward: the later slots' weight gradients start under the earlier slots' chain
            a, b = self.self_blocks1[i], self.self_blocks2[i]
            sa, sb = _block_scales(a, x), _block_scales(b, xa)
            ca, cb = self.blocks1[i], self.blocks2[i]
            # (the cross pair's LayerNorm 1 of the self pair's outputs rides in the self pair's launch: functional.FUSE_NEXT_LN)
            Fn.CTX.next_ln = [(ca.norm1.weight, ca.norm1.bias), (cb.norm1.weight, cb.norm1.bias)] \
                if ca.norm1.eps == a.norm1.eps == cb.norm1.eps else None
            x, xa = Fn.SelfPairFn.apply(x, xa, sa[0], sa[1], sb[0], sb[1], a.num_heads, a.norm1.eps, grad_mode,
                                        *_block_params(a, Fn.SELF_KEYS), *_block_params(b, Fn.SELF_KEYS))
            a, b = self.blocks1[i], self.blocks2[i]
            sa, sb = _block_scales(a, x), _block_scales(b, xa)
            Fn.CTX.cross_after_self = (x.data_ptr(), xa.data_ptr())      # (self pair -> cross pair, nothing in between)
            x, xa = Fn.CrossPairFn.apply(x, xa, sa[0], sa[1], sb[0], sb[1], a.num_heads, a.norm1.eps, grad_mode,
                                         *_block_params(a, Fn.CROSS_KEYS), *_block_params(b, Fn.CROSS_KEYS))
        return x, xa

    def forward(self, x, xa):
        Fn.run_entry_hook()                            # (engine: side work parked for this point of the forward)
        if (Fn.CTX.flush_points or Fn.CTX.defer_calls) and x.requires_grad:
            x, xa = Fn.FlushPointFn.apply(x, xa, id(self))   # backward: launch this stage's queued weight gradients on a side stream
        if self._pair_fusable(x, xa):
            x, xa = self._forward_pairs(x, xa)
            resample = getattr(self, self._resample_attr)
            if resample is not None:
                if _joint_ok(x, xa):     # both modalities through the shared module in ONE pass (the pair kernels wrote them adjacent)
                    return (x, xa) + tuple(Fn.SplitFn.apply(resample(Fn.JoinFn.apply(x, xa))))
                return (x, xa) + _both(resample, x, xa)
            return x, xa, x, xa
        side = _side_stream(x.device) if (PARALLEL_MODALITIES and FORK_AUTOGRAD_STREAMS and x.is_cuda) else None
        if side is None:
            for i in range(self.depth):
                x, xa = self.self_blocks1[i](x), self.self_blocks2[i](xa)
                x, xa = self.blocks1[i](x, xa), self.blocks2[i](xa, x)
        else:
            # The two modalities' blocks of one depth slot are independent (MS.py:700-701): issue them on two HIP streams so the
            # launch-/latency-bound kernels of the 8^3 / 4^3 stages overlap on the 256 CUs.  autograd replays each op's
            # backward on its forward stream, so the backward pass forks the same way; a captured HIP graph keeps the branches.
            main = torch.cuda.current_stream()
            for i in range(self.depth):
                side.wait_stream(main)                              # xa (and, later, x) were produced / consumed on main
                x1 = self.self_blocks1[i](x)
                with torch.cuda.stream(side):
                    xa1 = self.self_blocks2[i](xa)
                main.wait_stream(side)
                side.wait_stream(main)
                xa1.record_stream(main)
                x1.record_stream(side)
                x = self.blocks1[i](x1, xa1)
                with torch.cuda.stream(side):
                    xa = self.blocks2[i](xa1, x1)
                main.wait_stream(side)
                xa.record_stream(main)
        resample = getattr(self, self._resample_attr)
        if resample is not None:
            return (x, xa) + _both(resample, x, xa)
        return x, xa, x, xa


class BasicLayerUp(BasicLayer):
    """Same stage with the resampling module registered as `upsample` (MS.py:710-834; never instantiated by MicFormer)."""

    _resample_attr = "upsample"

    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, upsample=None, use_checkpoint=False):
        super().__init__(dim, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path,
                         norm_layer, upsample, use_checkpoint)


class PatchEmbed3D(nn.Module):
    """Conv3d(in_chans=1 -> E, k=s=patch) with right zero-padding (MS.py:837-878).  Standalone forward keeps the
    reference's channels-first output; MicFormer uses the channels-last kernel output directly."""

    def __init__(self, patch_size=(4, 4, 4), in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        patch_size = tuple(patch_size)
        if len(set(patch_size)) != 1:
            raise NotImplementedError("HIP patch embed implements cubic patches")
        if in_chans != 1:
            raise NotImplementedError("HIP patch embed implements in_chans=1 (Head splits the modalities, MS.py:1050)")
        self.patch_size = patch_size
        self.in_chans = in_chans
        self.embed_dim = embed_dim
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def tokens(self, vol, mod):
        """vol (B, nmod, D, H, W), modality index -> (B, D', H', W', E) channels-last (+ optional norm)."""
        y = Fn.PatchEmbedFn.apply(vol, mod, self.proj.weight, self.proj.bias, self.patch_size[0])
        if self.norm is not None:
            y = Fn.LayerNormFn.apply(y, None, self.norm.weight, self.norm.bias, self.norm.eps)
        return y

    def tokens_from_rows(self, rows, grid):
        """rows [B * D' * H' * W', k^3] (the patch-row matrix of one modality, ops.patch_rows_prepared), grid (B, D', H', W')."""
        y = Fn.PatchRowsEmbedFn.apply(rows, self.proj.weight, self.proj.bias).reshape(tuple(grid) + (self.embed_dim,))
        if self.norm is not None:
            y = Fn.LayerNormFn.apply(y, None, self.norm.weight, self.norm.bias, self.norm.eps)
        return y

    def forward(self, x):
        return self.tokens(x, 0).permute(0, 4, 1, 2, 3)


class MicFormer(nn.Module):
    """4 encoder stages + 4 decoder stages (all BasicLayer) with skip concat+Linear, shared patch embed / merging across
    the two modalities, final LN over cat[moving, fixed] and ConvTranspose3d k4 s4  (MS.py:881-1039)."""

    def __init__(self, pretrained=None, pretrained2d=False, patch_size=(4, 4, 4), in_chans=1, embed_dim=64,
                 depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=(7, 7, 7), mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm,
                 patch_norm=False, frozen_stages=-1, use_checkpoint=False):
        super().__init__()
        if drop_rate != 0.:
            raise NotImplementedError("drop_rate must be 0 (reference configuration)")
        self.pretrained = pretrained
        self.pretrained2d = pretrained2d
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.patch_norm = patch_norm
        self.frozen_stages = frozen_stages
        self.window_size = tuple(window_size)
        self.patch_size = tuple(patch_size)
        self.depths = list(depths)

        self.patch_embed = PatchEmbed3D(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                        norm_layer=norm_layer if patch_norm else None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]      # MS.py:941

        def stage(i, resample):
            return BasicLayer(dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i],
                              window_size=self.window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                              drop=drop_rate, attn_drop=attn_drop_rate,
                              drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                              downsample=resample, use_checkpoint=use_checkpoint)

        self.layers = nn.ModuleList(
            [stage(i, PatchMerging if i < self.num_layers - 1 else None) for i in range(self.num_layers)])
        self.up_layers = nn.ModuleList()
        self.concat_back_dim = nn.ModuleList()
        for i in reversed(range(self.num_layers)):
            c = int(embed_dim * 2 ** i)
            self.up_layers.append(stage(i, PatchExpand if i > 0 else None))
            self.concat_back_dim.append(nn.Linear(2 * c, c))          # index 0 is never used in forward (MS.py:1015-1016)
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.norm = norm_layer(self.num_features)
        self.norm2 = norm_layer(self.embed_dim * 2)
        self.reverse_patch_embedding = nn.ConvTranspose3d(2 * embed_dim, embed_dim // 2, self.patch_size,
                                                          stride=self.patch_size[0])

    def _predraw_drop_path(self, batch, device):
        """One batched draw of every block's two per-sample DropPath scales (mask / keep_prob, timm semantics)."""
        if not self.training:
            return
        every = self.__dict__.get("_dp_blocks")             # (the module tree is walked once: 3 ms of host time per forward)
        if every is None:
            every = self.__dict__["_dp_blocks"] = [b for b in self.modules() if isinstance(b, (TransformerBlock3D, CrossTransformerBlock3D))]
        blocks = [b for b in every if isinstance(b.drop_path, DropPath) and b.drop_path.drop_prob > 0.0]
        if not blocks:
            return
        cache = self.__dict__.setdefault("_dp_keep_cache", {})          # device-resident keep-probabilities + RNG state (not
        ent = cache.get(device)                                         # buffers: state_dict stays the reference's)
        if ent is None or ent[0].shape[0] != 2 * len(blocks):
            from .. import ops
            keep = torch.tensor([1.0 - b.drop_path.drop_prob for b in blocks for _ in (0, 1)], dtype=torch.float32).to(device)
            # seeded from torch's CPU generator: torch.manual_seed(1234 + rank) makes the DropPath stream rank-distinct
            ent = (keep, ops.drop_path_rng(device, int(torch.randint(0, 2 ** 62, (1,)).item())))
            cache[device] = ent
        from .. import ops
        s = ops.drop_path_draw(ent[1], ent[0], batch)                   # ONE launch; the device counter advances per replay
        for i, b in enumerate(blocks):
            b.__dict__["_pending_scales"] = (s[2 * i], s[2 * i + 1])

    def features(self, vol_m, mod_m, vol_f, mod_f):
        """Channels-last (B, D', H', W', E/2) feature that feeds Head.out_conv."""
        x = self.coarse_features(vol_m, mod_m, vol_f, mod_f)
        rp = self.reverse_patch_embedding
        return Fn.ConvUpFn.apply(x, rp.weight, rp.bias, self.patch_size[0])

    def coarse_features(self, vol_m, mod_m, vol_f, mod_f):
        """Channels-last (B, D/P, H/P, W/P, 2E) tokens after norm2, the input of reverse_patch_embedding (MS.py:1033-1036)."""
        self._predraw_drop_path(vol_m.shape[0], vol_m.device)
        if hasattr(vol_m, "patch_rows"):        # data.RawBatch: the input tail runs inside the patch gather
            P = self.patch_size[0]
            rows, grid = vol_m.patch_rows(P)
            if JOINT_MODALITIES and Fn._adjacent(rows[0], rows[1]):
                m, f = Fn.SplitFn.apply(self.patch_embed.tokens_from_rows(Fn._joined(rows[0], rows[1]), (2 * grid[0],) + tuple(grid[1:])))
            else:
                m = self.patch_embed.tokens_from_rows(rows[0], grid)
                f = self.patch_embed.tokens_from_rows(rows[1], grid)
        elif (JOINT_MODALITIES and vol_m is vol_f and (mod_m, mod_f) == (0, 1) and vol_m.is_cuda and vol_m.shape[1] == 2
              and Fn.PATCH_GEMM and self.patch_embed.norm is None and self.patch_size[0] in (2, 4)):
            pe = self.patch_embed                    # both modalities of the input in one GEMM (shared weights)
            m, f = Fn.SplitFn.apply(Fn.PatchEmbedPairFn.apply(vol_m, pe.proj.weight, pe.proj.bias, self.patch_size[0]))
        else:
            m = self.patch_embed.tokens(vol_m, mod_m)
            f = self.patch_embed.tokens(vol_f, mod_f)
        return self._coarse_from_tokens(m, f)

    def _coarse_from_tokens(self, m, f):
        Fn.clear_skip_tokens()
        skips = []
        for layer in self.layers:
            m_out, f_out, m, f = layer(m, f)
            skips.append((m_out, f_out))
        ln = self.norm
        if _joint_ok(m, f):
            m, f = Fn.SplitFn.apply(Fn.LayerNormFn.apply(Fn.JoinFn.apply(m, f), None, ln.weight, ln.bias, ln.eps))
        else:
            m = Fn.LayerNormFn.apply(m, None, ln.weight, ln.bias, ln.eps)
            f = Fn.LayerNormFn.apply(f, None, ln.weight, ln.bias, ln.eps)
        last = self.num_layers - 1
        for inx, up in enumerate(self.up_layers):
            if inx > 0:
                sm, sf = skips[last - inx]
                if m.shape != sm.shape:                                # odd token grids (MS.py:1018-1025)
                    m = Fn.ResizeTrilinearFn.apply(m, tuple(sm.shape[1:4]))
                    f = Fn.ResizeTrilinearFn.apply(f, tuple(sf.shape[1:4]))
                lin = self.concat_back_dim[inx]
                if _joint_ok(m, f) and _joint_ok(sm, sf):
                    skip = Fn.JoinFn.apply(sm, sf)
                    tok = Fn.skip_token(skip) if skip.requires_grad else None
                    if tok is not None:                   # its gradient joins the stage output's other gradient inside PatchMerging's backward
                        skip = Fn.SkipMailFn.apply(skip, tok)
                    m, f = Fn.SplitFn.apply(Fn.LinearFn.apply(Fn.JoinFn.apply(m, f), skip, lin.weight, lin.bias))
                else:
                    m, f = _both(lambda t: Fn.LinearFn.apply(t[0], t[1], lin.weight, lin.bias), (m, sm), (f, sf))
            _, _, m, f = up(m, f)
        return Fn.LayerNormFn.apply(m, f, self.norm2.weight, self.norm2.bias, self.norm2.eps)

    def forward(self, moving, fixed):
        """(B,1,D,H,W) x 2 -> (B, E/2, D', H', W'), as the reference (a channels-first VIEW of the kernel output)."""
        return self.features(moving, 0, fixed, 0).permute(0, 4, 1, 2, 3)


class Head(nn.Module):
    """Segmentation network: MicFormer + Conv3d(E/2 -> num_classes, 3, padding=1)  (MS.py:1042-1055)."""

    def __init__(self, n_channels=1, embed_dim=96, num_classes=14, window_size=(2, 2, 2), *, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), drop_path_rate=0.2):
        super().__init__()
        self.swin = MicFormer(window_size=window_size, in_chans=n_channels, embed_dim=embed_dim, depths=list(depths),
                              num_heads=list(num_heads), drop_path_rate=drop_path_rate)
        self.out_conv = nn.Conv3d(embed_dim // 2, num_classes, 3, padding=1)

    def forward(self, x):
        if x.dim() != 5 or x.shape[1] != 2:
            raise ValueError("Head expects (B, 2, D, H, W): torch.split(x, 1, dim=1) must give (moving, fixed)")
        if not x.is_cuda:
            raise RuntimeError("micformer_amd.Head runs on the MI355X HIP kernels only (no CPU path): move the model "
                               "and the input to cuda")
        if not hasattr(x, "patch_rows"):        # (a data.RawBatch keeps its raw float16 / float32 volume: prepared in the patch gather)
            x = x.float().contiguous()
        rp, oc = self.swin.reverse_patch_embedding, self.out_conv
        if FUSE_HEAD_TAIL and 2 <= rp.kernel_size[0] <= 8 and oc.out_channels <= 32:
            # ConvTranspose3d(k = s = P) and the 3^3 Conv3d have nothing between them: one composed linear map (head_tail.hip)
            # (the composition reads weights only: in engine mode it runs on the side stream under the first stage)
            wb = bf = wut = packs = None
            composed = []
            if PARALLEL_MODALITIES:
                from .. import ops
                main, side = torch.cuda.current_stream(), _side_stream(x.device)

                def compose():
                    if composed:
                        return
                    side.wait_stream(main)
                    if HEAD_WEIGHTS_AFTER is not None and HEAD_WEIGHTS_AFTER[0] is not None and Fn.CTX.segmenter is None:
                        side.wait_event(HEAD_WEIGHTS_AFTER[0])          # (step_many: the head's carried Adam update)
                    with torch.cuda.stream(side):
                        wu = ops.head_tail_transposed_up(rp.weight)
                        w, b = ops.head_tail_compose(rp.weight, rp.bias, oc.weight, wu)
                        pk = None
                        if Fn.FUSE_TAIL_PATCHES and ops.head_tail_fused_supported(
                                (x.shape[0],) + tuple(s // rp.kernel_size[0] for s in x.shape[2:]), rp.in_channels, oc.out_channels,
                                rp.kernel_size[0]) and all(s % rp.kernel_size[0] == 0 for s in x.shape[2:]):
                            pk = ops.head_tail_pack(w, b, oc.bias, rp.kernel_size[0])
                    composed.append((w, b, wu, pk))
                if HEAD_WEIGHTS_AFTER is None:
                    compose()                                           # under the first stage, as always
                else:
                    # the head's weights are updated by work the engine launches at the first stage entry (TrainEngine.step_many):
                    # compose behind it, at the entry of the last stage
                    Fn.park_entry_hook(compose, at=len(self.swin.layers) + len(self.swin.up_layers))
            coarse = self.swin.coarse_features(x, 0, x, 1)
            if PARALLEL_MODALITIES:
                compose()                                               # (fewer stage entries than expected: now)
                wb, bf, wut, packs = composed[0]
                main.wait_stream(side)
                for t in (wb, bf, wut) + (tuple(packs) if packs is not None else ()):
                    t.record_stream(main)
            return Fn.HeadTailFn.apply(coarse, rp.weight, rp.bias, oc.weight, oc.bias, wb, bf, wut, packs)
        feat = self.swin.features(x, 0, x, 1)
        return Fn.OutConvFn.apply(feat, oc.weight, oc.bias)

    def can_accumulate(self, roi=None):
        """True when forward_accumulate can take windows of spatial size `roi`: the composed head is on and supports this
        patch size / class count, `roi` is a multiple of the patch size (else the model pads the coarse grid and the patches
        would spill into the neighbouring accumulator regions), and forward() is Head's own (a subclass that post-processes its
        logits must go through the generic prediction path)."""
        rp, oc = self.swin.reverse_patch_embedding, self.out_conv
        P = rp.kernel_size[0]
        ok = FUSE_HEAD_TAIL and 2 <= P <= 8 and oc.out_channels <= 32 and type(self).forward is Head.forward
        return bool(ok and (roi is None or all(int(r) % P == 0 for r in roi)))

    def forward_accumulate(self, x, out, count, coords):
        """Sliding-window inference (utils.py:226-234) with the accumulate / count epilogue fused into the logits store: x holds n
        windows (n, 2, rd, rh, rw); window i's logits are ADDED into the fp32 volume accumulator `out` (VB, classes, D, H, W) at
        coords[i] = (sample, z0, y0, x0) (int32 device tensor [n, 4]) and `count` (VB, D, H, W) += 1 there.  No gradient; the
        prediction tensor is never materialised.  Returns None."""
        from .. import ops
        rp, oc = self.swin.reverse_patch_embedding, self.out_conv
        if not self.can_accumulate(tuple(x.shape[2:])) or torch.is_grad_enabled():
            raise RuntimeError("forward_accumulate needs the composed head, window dims the patch size divides and torch.no_grad()")
        x = x.float().contiguous()
        P = rp.kernel_size[0]
        wut = ops.head_tail_transposed_up(rp.weight)
        wb, bf = ops.head_tail_compose(rp.weight, rp.bias, oc.weight, wut)
        coarse = self.swin.coarse_features(x, 0, x, 1)
        n, Dc, Hc, Wc, Ci = coarse.shape
        if ops.head_tail_fused_supported((n, Dc, Hc, Wc), Ci, oc.out_channels, P):
            # bf16 mode: the patch-matrix-free head (head_tail_fused.hip) with the accumulate in its logits store -- no
            # [windows * tokens, 6^3 * classes] intermediate (1.6 GB per 7 windows of 128^3)
            pack_fwd, _ = ops.head_tail_pack(wb, bf, oc.bias, P)
            ops.head_tail_fwd_fused_sw(coarse.reshape(-1, Ci), pack_fwd, out, count, coords, (n, Dc, Hc, Wc), oc.out_channels, P)
            return
        t = ops.linear_fwd(coarse.reshape(-1, Ci), wb, bf)
        ops.head_tail_col2im_sw(t, oc.bias, out, count, coords, (n, Dc, Hc, Wc), P)
