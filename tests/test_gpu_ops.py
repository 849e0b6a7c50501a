"""GPU parity tests, op level: every C-ABI entry point against the CPU oracle / a plain PyTorch fp32 reference of the
same op, on seeded inputs, forward and backward.  Tolerance: fp32 kernels vs fp32 CPU -> 2e-5 abs + 1e-4 of the
tensor's max (atomically accumulated weight gradients: 2e-4 of max).  Run on the GPU box: pytest -m gpu.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import fill  # noqa: E402
from oracle import micformer_ref as R  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micformer_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape) * 7919)
    return (torch.randn(*shape, generator=g) * scale).float()


def dev(t):
    return t.cuda().contiguous() if t is not None else None


def close(got, want, atol=2e-5, rtol=1e-4, what=""):
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = max(float(want.abs().max()), 1e-30)
    err = float((got - want).abs().max())
    assert math.isfinite(err), f"{what}: non-finite error"
    assert err <= atol + rtol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.3e})"


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,C,c1", [(37, 48, 48), (130, 24, 24), (65, 96, 48), (9, 384, 384), (5, 16, 16)])
def test_layernorm(ops, rows, C, c1):
    x = rnd(rows, C, seed=1) * 2 + 0.3
    g, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    dy, add = rnd(rows, C, seed=4), rnd(rows, C, seed=5)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = R.layer_norm(xr, gr, br)
    gx, gg, gb = torch.autograd.grad((y * dy).sum(), [xr, gr, br])
    x1, x2 = (x, None) if c1 == C else (x[:, :c1].contiguous(), x[:, c1:].contiguous())
    yy, mean, rstd = ops.layernorm_fwd(dev(x1), dev(g), dev(b), 1e-5, dev(x2))
    close(yy, y, what="ln y")
    close(mean, x.mean(1), what="ln mean")
    dg, db = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    use_add = c1 == C
    r = ops.layernorm_bwd(dev(dy), dev(x1), mean, rstd, dev(g), dg, db, dev(x2), add=dev(add) if use_add else None)
    dx = r if c1 == C else torch.cat([r[0], r[1]], 1)
    close(dx, gx + (add if use_add else 0), what="ln dx")
    close(dg, gg, rtol=2e-4, what="ln dgamma")
    close(db, gb, rtol=2e-4, what="ln dbeta")


def test_layernorm_deferred_parameter_gradients(ops):
    """Partial form of the backward (per-workgroup [2C] partials, no atomics) + micf_layernorm_bwd_finish over several LayerNorms
    at once: same dx, and dgamma / dbeta ACCUMULATED onto non-zero buffers.  70 LayerNorms = two finish launches."""
    queued, want = [], []
    for n in range(70):
        rows, C = [(4096, 48), (1024, 192), (8192, 96), (128, 384), (37, 24)][n % 5]
        x = rnd(rows, C, seed=10 + n) * 1.5 - 0.2
        g = 1 + 0.1 * rnd(C, seed=20 + n)
        dy = rnd(rows, C, seed=30 + n)
        xr, gr = x.clone().requires_grad_(True), g.clone().requires_grad_(True)
        br = torch.zeros(C, requires_grad=True)
        gx, gg, gb = torch.autograd.grad((R.layer_norm(xr, gr, br) * dy).sum(), [xr, gr, br])
        _, mean, rstd = ops.layernorm_fwd(dev(x), dev(g), dev(torch.zeros(C)), 1e-5)
        dg0, db0 = rnd(C, seed=40 + n), rnd(C, seed=50 + n)
        j = len(want) - 5                   # same (rows, C) shape class five calls ago
        if n >= 10 and n % 7 == 0 and want[j][0].numel() == C:   # a module applied twice (shared by both modalities)
            dg, db = want[j][0], want[j][1]
            want[j] = (dg, db, want[j][2] + gg, want[j][3] + gb)
            dx = ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), dg, db, defer=queued)
            close(dx, gx, what=f"deferred ln dx [{n}]")
            continue
        dg, db = dev(dg0), dev(db0)
        dx = ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), dg, db, defer=queued)
        close(dx, gx, what=f"deferred ln dx [{n}]")
        want.append((dg, db, dg0 + gg, db0 + gb))
    assert len(queued) == 70
    ops.layernorm_bwd_finish(queued)
    for n, (dg, db, wg, wb) in enumerate(want):
        close(dg, wg, rtol=3e-4, what=f"deferred dgamma [{n}]")
        close(db, wb, rtol=3e-4, what=f"deferred dbeta [{n}]")


# ----------------------------------------------------------------------------- Linear
@pytest.mark.parametrize("M,N,K,k1", [(200, 48, 48, 48), (513, 96, 48, 48), (70, 192, 48, 48), (64, 48, 192, 192),
                                        (333, 24, 24, 24), (90, 96, 192, 96), (50, 3, 16, 16), (77, 10, 6, 6),
                                        (1000, 384, 768, 384), (8, 1536, 384, 384)])
def test_linear_fwd_bwd(ops, M, N, K, k1):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), 0.1 * rnd(N, seed=3)
    resid, dy = rnd(M, N, seed=4), rnd(M, N, seed=5)
    rps = max(M // 3, 1)
    s = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25][: (M + rps - 1) // rps])
    a1, a2 = (a, None) if k1 == K else (a[:, :k1].contiguous(), a[:, k1:].contiguous())
    srow = s[torch.arange(M) // rps].unsqueeze(1)
    # forward variants
    close(ops.linear_fwd(dev(a1), dev(w), dev(b), dev(a2)), F.linear(a, w, b), what="plain")
    y, pre = ops.linear_fwd(dev(a1), dev(w), dev(b), dev(a2), act=1, want_pre=True)
    close(pre, F.linear(a, w, b), what="pre")
    close(y, R.gelu(F.linear(a, w, b)), what="gelu")
    y = ops.linear_fwd(dev(a1), dev(w), dev(b), dev(a2), resid=dev(resid), dp_scale=dev(s), rows_per_sample=rps)
    close(y, resid + srow * F.linear(a, w, b), what="resid+droppath")
    y = ops.linear_fwd(dev(a1), dev(w), None, dev(a2), resid=dev(resid))
    close(y, resid + F.linear(a, w), what="resid nobias")
    # backward: data
    want = (srow * dy) @ w
    r = ops.linear_bwd_data(dev(dy), dev(w), dp_scale=dev(s), rows_per_sample=rps, k1=k1)
    got = r if k1 == K else torch.cat([r[0], r[1]], 1)
    close(got, want, what="dA")
    h = rnd(M, K, seed=6)
    hr = h.clone().requires_grad_(True)
    gh, = torch.autograd.grad((R.gelu(hr) * (dy @ w)).sum(), hr)
    close(ops.linear_bwd_data(dev(dy), dev(w), pre_act=dev(h)), gh, what="dA * gelu'")
    if k1 == K:
        base = rnd(M, K, seed=7)
        out = dev(base.clone())
        ops.linear_bwd_data(dev(dy), dev(w), out=out, accumulate=True)
        close(out, base + dy @ w, what="dA accumulate")
    # backward: weight
    dw, db = torch.zeros(N, K).cuda(), torch.zeros(N).cuda()
    ops.linear_bwd_weight(dev(dy), dev(a1), dw, db, dev(a2), dp_scale=dev(s), rows_per_sample=rps)
    close(dw, (srow * dy).t() @ a, rtol=2e-4, what="dW")
    close(db, (srow * dy).sum(0), rtol=2e-4, what="db")
    if k1 == K:
        dw.zero_()
        ops.linear_bwd_weight(dev(dy), dev(a), dw, None, a_gelu=True)
        close(dw, dy.t() @ R.gelu(a), rtol=2e-4, what="dW gelu(A)")


def test_linear_large_rows_split_reduction(ops):
    M, N, K = 40000, 48, 192
    a, dy = rnd(M, K, seed=11), rnd(M, N, seed=12)
    dw, db = torch.zeros(N, K).cuda(), torch.zeros(N).cuda()
    ops.linear_bwd_weight(dev(dy), dev(a), dw, db)
    close(dw, (dy.double().t() @ a.double()).float(), rtol=3e-4, what="dW split")
    close(db, dy.double().sum(0).float(), rtol=3e-4, what="db split")


def test_linear_bwd_weight_grouped(ops):
    """micf_linear_bwd_weight_grouped: 40 layers of mixed shape in one call (two launch groups), some with a DropPath
    scale per sample, some longer than one token split (workspace + grouped reduction), accumulating into non-zero dW."""
    shapes = [(128, 384, 384), (128, 1536, 384), (1024, 192, 768), (1024, 192, 192), (2048, 96, 96), (4096, 52, 100),
              (64, 48, 48), (3072, 384, 96), (16, 8, 4), (1024, 768, 192),
              # round 5: token counts that are no multiple of 16 (ragged tails added by the owning lanes): the large model's
              # 5 x 5 x 4 stage (100 tokens per sample, C = 768), fewer than one slab, a long ragged layer with splits
              (100, 768, 768), (200, 3072, 768), (7, 16, 8), (40, 48, 48), (2050, 96, 48), (300, 20, 36)]
    items, want = [], []
    for n in range(48):
        M, N, K = shapes[n % len(shapes)]
        a, dy = rnd(M, K, seed=100 + n), rnd(M, N, seed=200 + n)
        dw0, db0 = rnd(N, K, seed=300 + n), rnd(N, seed=400 + n)
        use_scale, use_bias = n % 3 == 0, n % 4 != 1
        B = 2 if M % 2 == 0 and (M % 32 == 0 or M % 16) else 1          # (ragged layers: two samples of M / 2 rows each)
        s = torch.tensor([0.0, 1.25][:B]) if n % 6 == 0 else torch.tensor([1.1, 0.7][:B])
        rps = M // B
        dys = dy * s.repeat_interleave(rps)[:, None] if use_scale else dy
        want.append((dw0.double() + dys.double().t() @ a.double(), db0.double() + dys.double().sum(0)))
        items.append((dev(dy), dev(a), dev(dw0), dev(db0) if use_bias else None, dev(s) if use_scale else None, rps))
        assert ops.wgrad_groupable(items[-1][0], items[-1][1], items[-1][4], rps)
    ops.linear_bwd_weight_grouped(items)
    for n, ((dy, a, dw, db, sc, rps), (wdw, wdb)) in enumerate(zip(items, want)):
        close(dw, wdw.float(), rtol=3e-4, what=f"grouped dW[{n}] {tuple(dw.shape)} M={dy.shape[0]}")
        if db is not None:
            close(db, wdb.float(), rtol=3e-4, what=f"grouped db[{n}]")
    assert ops.wgrad_groupable(dev(rnd(40, 48)), dev(rnd(40, 48)))              # (round 5: any token count)
    assert not ops.wgrad_groupable(dev(rnd(40, 48)), dev(rnd(40, 48)), dev(rnd(3)), 12)        # M % rows_per_sample != 0


@pytest.mark.parametrize("dims,Ci,Cm,Co,P", [((2, 3, 2, 4), 96, 24, 8, 4), ((1, 1, 1, 1), 48, 12, 8, 4), ((1, 2, 3, 1), 48, 12, 14, 4),
                                             ((1, 3, 3, 3), 32, 8, 5, 2), ((1, 4, 4, 4), 96, 24, 8, 4),
                                             ((1, 2, 3, 8), 48, 12, 8, 4), ((2, 2, 2, 16), 96, 24, 8, 4)])   # (W % 8 == 0: im2col via LDS)
def test_head_tail_composed(dims, Ci, Cm, Co, P):
    """HeadTailFn (ConvTranspose3d(k=s=P) + Conv3d(3, pad 1) composed, csrc/head_tail.hip) against the two torch convolutions:
    logits and all five gradients."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micformer_amd import functional as Fn
    B, Dc, Hc, Wc = dims
    x = rnd(B, Dc, Hc, Wc, Ci, seed=1)
    w_up, b_up = rnd(Ci, Cm, P, P, P, seed=2, scale=0.1), rnd(Cm, seed=3)
    w_out, b_out = rnd(Co, Cm, 3, 3, 3, seed=4, scale=0.1), rnd(Co, seed=5)
    dy = rnd(B, Co, Dc * P, Hc * P, Wc * P, seed=6)
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, w_up, b_up, w_out, b_out)]
    z = F.conv_transpose3d(ref_in[0].permute(0, 4, 1, 2, 3), ref_in[1], ref_in[2], stride=P)
    y_ref = F.conv3d(z, ref_in[3], ref_in[4], padding=1)
    y_ref.backward(dy.double())
    got_in = [dev(t).requires_grad_(True) for t in (x, w_up, b_up, w_out, b_out)]
    y = Fn.HeadTailFn.apply(*got_in)
    y.backward(dev(dy))
    close(y, y_ref.float(), what="logits")
    for name, g, r in zip(("dx", "dw_up", "db_up", "dw_out", "db_out"), got_in, ref_in):
        close(g.grad, r.grad.float(), rtol=3e-4, what=name)


# ----------------------------------------------------------------------------- window attention
def _attn_ref(q, kv, dims, heads, ws):
    B, D, H, W = dims
    C = q.shape[1]
    hd = C // heads
    qw = R._to_windows(q.reshape(B, D, H, W, C), ws)
    kw = R._to_windows(kv[:, :C].reshape(B, D, H, W, C), ws)
    vw = R._to_windows(kv[:, C:].reshape(B, D, H, W, C), ws)
    nW, N, _ = qw.shape
    sp = lambda t: t.reshape(nW, N, heads, hd).transpose(1, 2)
    att = torch.softmax((sp(qw) * hd ** -0.5) @ sp(kw).transpose(-1, -2), -1)
    o = (att @ sp(vw)).transpose(1, 2).reshape(nW, N, C)
    return R._from_windows(o, ws, B, D, H, W).reshape(-1, C)


@pytest.mark.parametrize("dims,C,heads,ws", [((2, 4, 6, 4), 48, 3, (2, 2, 2)), ((1, 2, 2, 2), 24, 3, (2, 2, 2)),
                                              ((1, 1, 1, 1), 192, 24, (1, 1, 1)), ((2, 6, 4, 2), 96, 3, (2, 2, 2)),
                                              ((1, 4, 2, 6), 48, 8, (2, 2, 2)), ((3, 1, 4, 4), 48, 3, (1, 2, 2))])
def test_window_attention(ops, dims, C, heads, ws):
    T = dims[0] * dims[1] * dims[2] * dims[3]
    q = rnd(T, C, seed=1).requires_grad_(True)
    kv = rnd(T, 2 * C, seed=2).requires_grad_(True)
    do = rnd(T, C, seed=3)
    o = _attn_ref(q, kv, dims, heads, ws)
    gq, gkv = torch.autograd.grad((o * do).sum(), [q, kv])
    scale = (C // heads) ** -0.5
    oo = ops.window_attn_fwd(dev(q.detach()), dev(kv.detach()), dims, heads, ws, scale)
    close(oo, o, what="attn o")
    dq, dkv = ops.window_attn_bwd(dev(q.detach()), dev(kv.detach()), dev(do), dims, heads, ws, scale)
    close(dq, gq, what="attn dq")
    close(dkv, gkv, what="attn dkv")


# ----------------------------------------------------------------------------- conv 3x3x3
@pytest.mark.parametrize("dims,c1,c2,N,ncdhw", [((2, 4, 5, 6), 24, 24, 16, False), ((1, 6, 4, 4), 48, 48, 16, False),
                                                ((2, 5, 4, 7), 12, 0, 8, True), ((1, 3, 3, 3), 24, 0, 8, True),
                                                ((1, 2, 1, 3), 6, 6, 16, False),
                                                # large enough token grids for the direct LDS-halo kernels (>= 32 tiles), ragged edges
                                                ((2, 9, 10, 19), 24, 24, 16, False), ((1, 8, 16, 32), 48, 48, 16, False),
                                                ((1, 16, 12, 20), 24, 0, 8, True), ((2, 7, 9, 17), 12, 0, 8, True),
                                                ((1, 10, 16, 16), 24, 24, 16, False),
                                                # two 96-channel slabs / the 8-wide tile of the MFMA data- and weight-gradient kernels
                                                ((1, 8, 8, 8), 96, 96, 16, False), ((2, 4, 8, 9), 48, 48, 16, False)])
def test_conv3(ops, dims, c1, c2, N, ncdhw):
    B, D, H, W = dims
    T = B * D * H * W
    x1 = rnd(T, c1, seed=1)
    x2 = rnd(T, c2, seed=2) if c2 else None
    w = rnd(N, c1 + c2, 3, 3, 3, seed=3) / math.sqrt(27 * (c1 + c2))
    b = 0.1 * rnd(N, seed=4)
    xin = torch.cat([x1] + ([x2] if c2 else []), 1).reshape(B, D, H, W, c1 + c2).permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv3d(xin, wr, br, padding=1)                                  # (B,N,D,H,W)
    dy = rnd(B, N, D, H, W, seed=5)
    gx, gw, gb = torch.autograd.grad((y * dy).sum(), [xin, wr, br])
    gx = gx.permute(0, 2, 3, 4, 1).reshape(T, c1 + c2)
    yy = ops.conv3_fwd(dev(x1), dev(w), dev(b), dims, x2=dev(x2), ncdhw_out=ncdhw)
    close(yy, y if ncdhw else y.permute(0, 2, 3, 4, 1).reshape(T, N), what="conv3 y")
    dyl = dy if ncdhw else dy.permute(0, 2, 3, 4, 1).reshape(T, N).contiguous()
    d1, d2 = ops.conv3_bwd_data(dev(dyl), dev(w), dims, c1, c2, ncdhw=ncdhw)
    close(d1, gx[:, :c1], what="conv3 dx1")
    if c2:
        close(d2, gx[:, c1:], what="conv3 dx2")
        base1, base2 = rnd(T, c1, seed=8), rnd(T, c2, seed=9)
        o1, o2 = dev(base1.clone()), dev(base2.clone())
        ops.conv3_bwd_data(dev(dyl), dev(w), dims, c1, c2, ncdhw=ncdhw, dx1=o1, dx2=o2, acc1=True, acc2=True)
        close(o1, base1 + gx[:, :c1], what="conv3 dx1 acc")
        close(o2, base2 + gx[:, c1:], what="conv3 dx2 acc")
    dw, db = torch.zeros_like(w).cuda(), torch.zeros(N).cuda()
    ops.conv3_bwd_weight(dev(dyl), dev(x1), dw, db, dims, x2=dev(x2), ncdhw=ncdhw)
    close(dw, gw, rtol=2e-4, what="conv3 dw")
    close(db, gb, rtol=2e-4, what="conv3 db")


@pytest.mark.parametrize("dims,C,n", [((1, 4, 4, 4), 96, 6), ((2, 4, 4, 4), 192, 2), ((1, 8, 8, 8), 96, 6), ((1, 16, 16, 16), 48, 4),
                                      ((1, 4, 6, 5), 24, 3), ((1, 8, 8, 8), 24, 13), ((1, 2, 3, 3), 24, 2), ((1, 4, 4, 6), 6, 2)])
def test_conv3_bwd_weight_grouped(ops, dims, C, n):
    """micf_conv3_bwd_weight_grouped: n offset-conv layers of one shape in one launch (4^3 grids on the MFMA kernel's masked
    8-wide tile; > 12 layers chunked; shapes outside the MFMA kernel layer by layer) == F.conv3d's weight gradient, accumulated."""
    B, D, H, W = dims
    T = B * D * H * W
    items, want = [], []
    for k in range(n):
        x1, x2 = rnd(T, C, seed=10 * k + 1), rnd(T, C, seed=10 * k + 2)
        dy = rnd(T, 16, seed=10 * k + 3)
        w0, b0 = rnd(16, 2 * C, 3, 3, 3, seed=10 * k + 4), rnd(16, seed=10 * k + 5)      # accumulation targets' previous content
        xin = torch.cat([x1, x2], 1).reshape(B, D, H, W, 2 * C).permute(0, 4, 1, 2, 3).contiguous()
        wr, br = torch.zeros(16, 2 * C, 3, 3, 3, requires_grad=True), torch.zeros(16, requires_grad=True)
        y = F.conv3d(xin, wr, br, padding=1)
        gw, gb = torch.autograd.grad((y * dy.reshape(B, D, H, W, 16).permute(0, 4, 1, 2, 3)).sum(), [wr, br])
        want.append((w0 + gw, b0 + gb))
        items.append((dev(dy), dev(x1), dev(x2), dev(w0.clone()), dev(b0.clone())))
    ops.conv3_bwd_weight_grouped(items, dims)
    for k, (it, (gw, gb)) in enumerate(zip(items, want)):
        close(it[3], gw, rtol=2e-4, what=f"grouped conv3 dw[{k}]")
        close(it[4], gb, rtol=2e-4, what=f"grouped conv3 db[{k}]")


# ----------------------------------------------------------------------------- offset head + deformable sampling
@pytest.mark.parametrize("case", [((2, 4, 6, 4), 24), ((1, 5, 5, 5), 48), ((1, 2, 2, 2), 24), ((1, 1, 1, 1), 96),
                                  ((1, 3, 5, 2), 96), ((1, 1, 4, 4), 24),
                                  ((1, 16, 16, 16), 48), ((2, 8, 16, 16), 24, 8.0), ((1, 16, 16, 16), 96, 0.5, 1),
                                  ((1, 16, 8, 32), 24, 2.0, 0),
                                  ((1, 16, 16, 16), 48, 0.5, None, 0), ((2, 8, 16, 16), 24, 8.0, None, 1), ((1, 5, 5, 5), 48, 3.0, None, 1),
                                  ((1, 16, 8, 32), 24, 2.0, None, 2), ((2, 6, 10, 12), 96, 1.0, None, 3), ((1, 3, 5, 2), 96, 0.5, None, 0),
                                  ((1, 16, 16, 16), 48, 0.5, None, 3, "40,12"), ((2, 6, 10, 12), 24, 1.0, None, 3, "0,12"), ((2, 4, 4, 4), 384, 0.5, None, 3, "512,2"),
                                  ((1, 16, 16, 16), 96, 0.5, None, 3, "512,0"), ((2, 8, 16, 16), 24, 0.05, None, 3, "100,3"),
                                  ((1, 16, 16, 16), 48, 0.5, None, 3, "512,12,0"), ((2, 6, 10, 12), 24, 0.3, None, 3, "512,12,1"),
                                  ((2, 4, 4, 4), 384, 0.5, None, None, "512,2"), ((2, 8, 8, 8), 192, 1.0, None, None, "0,12"),
                                  ((1, 5, 5, 5), 48, 3.0, None, None, "512,12,1"), ((2, 8, 8, 8), 192, 0.3), ((3, 4, 4, 4), 384, 0.3),
                                  ((1, 16, 16, 16), 96, 0.5, 1, -1), ((1, 16, 8, 32), 24, 2.0, 0, -1), ((2, 8, 16, 16), 24, 8.0, 8, -1)])
def test_offset_sample(ops, case, hook):
    """case = (dims, C[, offset scale[, hook "cell_cap"[, hook "sample_e"]]]).  d(xa) is summed per output tile in LDS from the NEAR
    tokens of a bounded neighbourhood (round 5, every grid), FAR tokens scatter atomically: the offset scale pushes taps far away
    / out of the volume, "sample_e" shrinks the neighbourhood (0: every token takes the atomic path; unset on a grid of <= 512 tokens per sample: the ONE-launch form whose candidate box is the whole sample; non-cubic grids, ragged
    tiles, two samples; a sixth entry = the hooks "tile_cap_hits / _voxel / _cell" as "hits,segment,cell": the capacities of the LDS hit list, of a voxel's list segments and of a cell's index list.  E = -1 switches the tile path off: the cell-list gather of rounds 1-4 on grids >= 4096 tokens, where the
    cap shrinks the per-cell lists so that tokens overflow into its atomic fallback (cap 0: every token)."""
    dims, C = case[0], case[1]
    wscale = case[2] if len(case) > 2 else 0.5
    if len(case) > 3 and case[3] is not None:
        hook("cell_cap", case[3])
    if len(case) > 4 and case[4] is not None:
        if case[4] < 0:
            hook("sample_tile", 0)
        else:
            hook("sample_e", case[4])
    if len(case) > 5:
        for name, v in zip(("tile_cap_hits", "tile_cap_voxel", "tile_cap_cell"), str(case[5]).split(",")):
            hook(name, int(v))                                  # LDS lists shrunk: the scanning lane's own atomic path / the owners walking the hit list
    B, D, H, W = dims
    T = B * D * H * W
    h = rnd(T, 16, seed=1).requires_grad_(True)
    xa = rnd(T, C, seed=2).requires_grad_(True)
    lg = (1 + 0.1 * rnd(16, seed=3)).requires_grad_(True)
    lb = (0.1 * rnd(16, seed=4)).requires_grad_(True)
    w1 = (rnd(3, 16, seed=5) * wscale).requires_grad_(True)
    dxs = rnd(T, C, seed=6)
    off = F.linear(R.gelu(R.layer_norm(h.reshape(B, D, H, W, 16), lg, lb)), w1)
    flow = off + R.reference_points(D, H, W)
    xs = R.trilinear_gather_zero(xa.reshape(B, D, H, W, C), R.sample_coords(flow)).reshape(T, C)
    g = torch.autograd.grad((xs * dxs).sum(), [xa, h, lg, lb, w1])
    fl, xx = ops.offset_sample_fwd(dev(h.detach()), dev(lg.detach()), dev(lb.detach()), dev(w1.detach()), dev(xa.detach()), dims, 1e-5)
    close(fl, flow.reshape(T, 3), what="flow")
    close(xx, xs, what="xs")
    base = rnd(T, C, seed=7)
    dxa = dev(base.clone())
    dlg, dlb, dw1 = torch.zeros(16).cuda(), torch.zeros(16).cuda(), torch.zeros(3, 16).cuda()
    dh = ops.offset_sample_bwd(dev(dxs), dev(h.detach()), dev(lg.detach()), dev(lb.detach()), dev(w1.detach()), dev(xa.detach()),
                               fl, dxa, dlg, dlb, dw1, dims, 1e-5)
    close(dxa, base + g[0], what="dxa")
    for got, want, name in ((dh, g[1], "dh"), (dlg, g[2], "dln_g"), (dlb, g[3], "dln_b"), (dw1, g[4], "dw1")):
        want = want.reshape(got.shape)
        assert torch.equal(torch.isfinite(got.cpu()), torch.isfinite(want)), f"{name}: NaN pattern differs from the oracle"
        fin = torch.isfinite(want)
        close(torch.where(fin, got.cpu(), torch.zeros_like(want)), torch.where(fin, want, torch.zeros_like(want)), rtol=3e-4, what=name)


def test_stn_module_against_reference_goldens():
    """Standalone SpatialTransformer vs vectors produced by the reference's STN.py (tests/golden/f2_stn.npz)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os
    from micformer_amd.models.STN import SpatialTransformer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f2_stn.npz"))
    stn = SpatialTransformer()
    for tag, (D, H, W) in {"s1": (1, 1, 1), "s2": (2, 2, 2), "s3": (3, 3, 3), "s5": (5, 5, 5), "nc": (4, 6, 8),
                           "nc2": (3, 5, 2), "flat": (1, 4, 4)}.items():
        src = fill.lattice((2, 8, D, H, W), f"F2.{tag}.src", 1.0, 0.7).cuda().requires_grad_(True)
        off = fill.lattice((2, D, H, W, 3), f"F2.{tag}.off", 1.3, 0.9)
        ref = torch.from_numpy(g[tag + "_ref"])
        pos = (off + ref).permute(0, 4, 1, 2, 3).contiguous().cuda().requires_grad_(True)
        y = stn(src, pos)
        close(y.permute(0, 2, 3, 4, 1), torch.from_numpy(g[tag + "_y"]), what=tag + " y")
        gy = fill.lattice((2, 8, D, H, W), f"F2.{tag}.gy", 1.0, 0.41).cuda()
        gs, gp = torch.autograd.grad((y * gy).sum(), [src, pos])
        close(gs.permute(0, 2, 3, 4, 1), torch.from_numpy(g[tag + "_gsrc"]), what=tag + " gsrc")
        want = torch.from_numpy(g[tag + "_gpos"])
        got = gp.permute(0, 2, 3, 4, 1).cpu()
        assert torch.equal(torch.isfinite(got), torch.isfinite(want)), tag + ": NaN pattern of d/dflow differs from the reference"
        fin = torch.isfinite(want)
        close(torch.where(fin, got, torch.zeros_like(got)), torch.where(fin, want, torch.zeros_like(want)), what=tag + " gpos")


# ----------------------------------------------------------------------------- patch convs
@pytest.mark.parametrize("shape,E", [((2, 2, 8, 8, 8), 24), ((1, 2, 9, 8, 10), 24), ((1, 2, 16, 12, 8), 48)])
def test_patch_embed(ops, shape, E):
    vol = rnd(*shape, seed=1)
    P = {"proj.weight": (rnd(E, 1, 4, 4, 4, seed=2) / 8).requires_grad_(True), "proj.bias": (0.1 * rnd(E, seed=3)).requires_grad_(True)}
    for mod in (0, 1):
        y = R.patch_embed(vol[:, mod:mod + 1], P, "")
        dy = rnd(*y.shape, seed=4)
        gw, gb = torch.autograd.grad((y * dy).sum(), list(P.values()))
        yy = ops.patch_embed_fwd(dev(vol), mod, dev(P["proj.weight"].detach()), dev(P["proj.bias"].detach()), 4)
        close(yy, y, what="embed y")
        dw, db = torch.zeros(E, 1, 4, 4, 4).cuda(), torch.zeros(E).cuda()
        ops.patch_embed_bwd_weight(dev(dy), dev(vol), mod, dw, db, 4)
        close(dw, gw, rtol=2e-4, what="embed dw")
        close(db, gb, rtol=2e-4, what="embed db")


@pytest.mark.parametrize("shape", [(2, 4, 6, 4, 24), (1, 5, 3, 6, 24), (1, 8, 8, 8, 48), (1, 2, 2, 2, 96)])
def test_conv_down(ops, shape):
    B, D, H, W, C = shape
    x = rnd(*shape, seed=1).requires_grad_(True)
    w = (rnd(2 * C, C, 2, 2, 2, seed=2) / math.sqrt(8 * C)).requires_grad_(True)
    b = (0.1 * rnd(2 * C, seed=3)).requires_grad_(True)
    xc = F.pad(x.permute(0, 4, 1, 2, 3), (0, W % 2, 0, H % 2, 0, D % 2))
    y = F.conv3d(xc, w, b, stride=2).permute(0, 2, 3, 4, 1)
    dy = rnd(*y.shape, seed=4)
    gx, gw, gb = torch.autograd.grad((y * dy).sum(), [x, w, b])
    close(ops.conv_down_fwd(dev(x.detach()), dev(w.detach()), dev(b.detach())), y, what="down y")
    close(ops.conv_down_bwd_data(dev(dy), dev(w.detach()), shape), gx, what="down dx")
    dw, db = torch.zeros_like(w).cuda(), torch.zeros(2 * C).cuda()
    ops.conv_down_bwd_weight(dev(dy), dev(x.detach()), dw, db)
    close(dw, gw, rtol=2e-4, what="down dw")
    close(db, gb, rtol=2e-4, what="down db")


@pytest.mark.parametrize("dims,C,k", [((2, 8, 6, 10), 12, 2), ((1, 5, 7, 9), 4, 2), ((2, 8, 8, 8), 1, 4), ((1, 6, 10, 7), 3, 4)])
def test_space_to_depth_and_back(ops, dims, C, k):
    """micf_space_to_depth: rows of k^3-voxel patches, column = c * k^3 + tap (zeros beyond odd dims); micf_depth_to_space
    inverts it (+ bias), cropping the padding.  Both are pure data movement: bit-exact."""
    B, D, H, W = dims
    x = dev(rnd(B, D, H, W, C, seed=1))
    a = ops.space_to_depth(x, dims, C, k)
    Dp, Hp, Wp = [-(-v // k) * k for v in (D, H, W)]
    xp = torch.zeros(B, Dp, Hp, Wp, C, device="cuda")
    xp[:, :D, :H, :W] = x
    ref = xp.view(B, Dp // k, k, Hp // k, k, Wp // k, k, C).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(-1, C * k ** 3)
    assert torch.equal(a, ref)
    assert torch.equal(ops.depth_to_space(a, dims, C, k), x)
    bias = dev(rnd(C, seed=2))
    assert torch.equal(ops.depth_to_space(a, dims, C, k, bias=bias), x + bias)
    out = torch.zeros(C * k ** 3, device="cuda")
    ops.colsum_(a, out)
    close(out, a.sum(0), rtol=1e-5, what="colsum")


@pytest.mark.parametrize("gemm", [True, False])
def test_patch_conv_functions_match_torch(ops, gemm):
    """PatchEmbedFn / ConvDownFn / ConvUpFn (both forms: space-to-depth + linear GEMMs, element-gather GEMMs) against
    F.conv3d / F.conv_transpose3d, forward and all gradients, odd dims included."""
    from micformer_amd import functional as Fn
    prev = Fn.PATCH_GEMM
    Fn.PATCH_GEMM = gemm
    try:
        # patch embedding, k = 4, modality 1 of 2
        vol = rnd(1, 2, 9, 8, 10, seed=1)
        w = (rnd(24, 1, 4, 4, 4, seed=2) / 8).requires_grad_(True)
        b = (0.1 * rnd(24, seed=3)).requires_grad_(True)
        y = F.conv3d(F.pad(vol[:, 1:2], (0, 2, 0, 0, 0, 3)), w, b, stride=4).permute(0, 2, 3, 4, 1)
        dy = rnd(*y.shape, seed=4)
        gw, gb = torch.autograd.grad((y * dy).sum(), [w, b])
        wd, bd = dev(w.detach()).requires_grad_(True), dev(b.detach()).requires_grad_(True)
        yy = Fn.PatchEmbedFn.apply(dev(vol), 1, wd, bd, 4)
        close(yy, y, what="embed y")
        g = torch.autograd.grad((yy * dev(dy)).sum(), [wd, bd])
        close(g[0], gw, rtol=2e-4, what="embed dw"); close(g[1], gb, rtol=2e-4, what="embed db")
        # PatchMerging conv, k = 2, odd dims
        x = rnd(1, 5, 3, 6, 24, seed=5).requires_grad_(True)
        w = (rnd(48, 24, 2, 2, 2, seed=6) / math.sqrt(192)).requires_grad_(True)
        b = (0.1 * rnd(48, seed=7)).requires_grad_(True)
        y = F.conv3d(F.pad(x.permute(0, 4, 1, 2, 3), (0, 0, 0, 1, 0, 1)), w, b, stride=2).permute(0, 2, 3, 4, 1)
        dy = rnd(*y.shape, seed=8)
        ref = torch.autograd.grad((y * dy).sum(), [x, w, b])
        xd, wd, bd = (dev(t.detach()).requires_grad_(True) for t in (x, w, b))
        yy = Fn.ConvDownFn.apply(xd, wd, bd)
        close(yy, y, what="down y")
        for got, want, tag in zip(torch.autograd.grad((yy * dev(dy)).sum(), [xd, wd, bd]), ref, ("dx", "dw", "db")):
            close(got, want, rtol=2e-4, what="down " + tag)
        # PatchExpand transposed conv, k = 2 and the reverse patch embedding, k = 4
        for k, N in ((2, 24), (4, 12)):
            x = rnd(2, 2, 3, 2, 48, seed=9).requires_grad_(True)
            w = (rnd(48, N, k, k, k, seed=10) / math.sqrt(48)).requires_grad_(True)
            b = (0.1 * rnd(N, seed=11)).requires_grad_(True)
            y = F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), w, b, stride=k).permute(0, 2, 3, 4, 1)
            dy = rnd(*y.shape, seed=12)
            ref = torch.autograd.grad((y * dy).sum(), [x, w, b])
            xd, wd, bd = (dev(t.detach()).requires_grad_(True) for t in (x, w, b))
            yy = Fn.ConvUpFn.apply(xd, wd, bd, k)
            close(yy, y, what=f"up{k} y")
            for got, want, tag in zip(torch.autograd.grad((yy * dev(dy)).sum(), [xd, wd, bd]), ref, ("dx", "dw", "db")):
                close(got, want, rtol=2e-4, what=f"up{k} " + tag)
    finally:
        Fn.PATCH_GEMM = prev


@pytest.mark.parametrize("shape,N,k", [((2, 2, 3, 2, 48), 24, 2), ((1, 4, 4, 4, 96), 48, 2), ((1, 3, 2, 4, 48), 12, 4),
                                        ((2, 2, 2, 2, 96), 24, 4)])
def test_conv_up(ops, shape, N, k):
    B, D, H, W, C = shape
    x = rnd(*shape, seed=1).requires_grad_(True)
    w = (rnd(C, N, k, k, k, seed=2) / math.sqrt(C)).requires_grad_(True)
    b = (0.1 * rnd(N, seed=3)).requires_grad_(True)
    y = F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), w, b, stride=k).permute(0, 2, 3, 4, 1)
    dy = rnd(*y.shape, seed=4)
    gx, gw, gb = torch.autograd.grad((y * dy).sum(), [x, w, b])
    close(ops.conv_up_fwd(dev(x.detach()), dev(w.detach()), dev(b.detach()), k), y, what="up y")
    close(ops.conv_up_bwd_data(dev(dy), dev(w.detach()), shape, k), gx, what="up dx")
    dw, db = torch.zeros_like(w).cuda(), torch.zeros(N).cuda()
    ops.conv_up_bwd_weight(dev(dy), dev(x.detach()), dw, db, k)
    close(dw, gw, rtol=2e-4, what="up dw")
    close(db, gb, rtol=2e-4, what="up db")


# ----------------------------------------------------------------------------- pad / crop / resize
def test_pad_crop_resize(ops):
    dims, pd, C = (2, 5, 3, 4), (6, 4, 4), 24
    x = rnd(2 * 5 * 3 * 4, C, seed=1)
    xp = ops.pad3d(dev(x), dims, pd)
    want = F.pad(x.reshape(2, 5, 3, 4, C), (0, 0, 0, 0, 0, 1, 0, 1)).reshape(-1, C)
    close(xp, want, atol=0, rtol=0, what="pad")
    close(ops.crop3d(xp, dims, pd), x, atol=0, rtol=0, what="crop")
    base = rnd(*x.shape, seed=2)
    o = dev(base.clone())
    ops.crop3d(xp, dims, pd, out=o, accumulate=True)
    close(o, base + x, what="crop acc")
    v = rnd(1, 3, 2, 5, 8, seed=3).requires_grad_(True)
    size = (5, 3, 4)
    y = F.interpolate(v.permute(0, 4, 1, 2, 3), size=size, mode="trilinear", align_corners=True).permute(0, 2, 3, 4, 1)
    dy = rnd(*y.shape, seed=4)
    gv, = torch.autograd.grad((y * dy).sum(), v)
    close(ops.resize_trilinear_fwd(dev(v.detach()), size), y, what="resize")
    close(ops.resize_trilinear_bwd(dev(dy), tuple(v.shape)), gv, what="resize bwd")


# ----------------------------------------------------------------------------- loss / metrics / optimiser
def test_dice_bce_against_reference_golden(ops):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f6_loss.npz"))
    z = torch.from_numpy(g["z"])
    t = fill.one_hot(torch.from_numpy(g["label"]).long())
    loss, sums = ops.dice_bce_fwd(dev(z), dev(t))
    close(loss.reshape(()), torch.from_numpy(g["loss"]), atol=2e-6, what="loss vs reference")
    one = torch.ones(1).cuda()
    dz = ops.dice_bce_bwd(dev(z), dev(t), sums, one)
    close(dz, torch.from_numpy(g["gz"]), atol=1e-9, rtol=2e-4, what="dloss/dz vs reference")
    from micformer_amd.loss.dice import MDiceLoss, MDiceLoss_Val
    zz = dev(z).requires_grad_(True)
    l = MDiceLoss()(zz, dev(t))
    (2 * l).backward()
    close(zz.grad, 2 * torch.from_numpy(g["gz"]), atol=1e-9, rtol=2e-4, what="module grad")
    close(MDiceLoss_Val()(dev(z), dev(t)), torch.from_numpy(g["val_loss"]), atol=2e-6, what="val loss")


def test_dice_bce_from_class_map(ops):
    """uint8 class-map target == one-hot float target (forward loss, sums and gradient), incl. the nn.Module front end."""
    from micformer_amd.loss.dice import MDiceLoss, MDiceLoss_Val
    B, K, D = 2, 8, 20
    z = rnd(B, K, D, D, D, seed=3, scale=2.0)
    lab = fill.make_label_map(B, D, D, D, K)
    onehot = fill.one_hot(lab, K)
    zc = dev(z).requires_grad_(True)
    zo = dev(z).requires_grad_(True)
    l_map = MDiceLoss()(zc, dev(lab))                  # int64 (B, D, H, W) -> uint8 inside
    l_hot = MDiceLoss()(zo, dev(onehot))
    l_map.backward(); l_hot.backward()
    assert float((l_map - l_hot).detach().abs()) <= 1e-7
    close(zc.grad, zo.grad, atol=1e-9, rtol=1e-6, what="dlogits from class map")
    close(MDiceLoss_Val()(zc.detach(), dev(lab).unsqueeze(1)), MDiceLoss_Val()(zo.detach(), dev(onehot)), atol=1e-7, what="val loss")
    want = R.mdice_loss(z, onehot)
    close(l_map.reshape(()), want.reshape(()), atol=2e-6, what="loss vs oracle")


def test_dice_bce_larger(ops):
    z = rnd(2, 8, 12, 10, 14, seed=1) * 3
    t = fill.one_hot(fill.make_label_map(2, 12, 10, 14))
    zr = z.clone().requires_grad_(True)
    l = R.mdice_loss(zr, t)
    gz, = torch.autograd.grad(l, zr)
    loss, sums = ops.dice_bce_fwd(dev(z), dev(t))
    close(loss.reshape(()), l, atol=2e-6, what="loss")
    close(ops.dice_bce_bwd(dev(z), dev(t), sums, torch.full((1,), 0.5).cuda()), 0.5 * gz, atol=1e-9, rtol=2e-4, what="dz")


def test_argmax_meandice(ops):
    z = rnd(2, 8, 6, 5, 7, seed=1)
    z[0, 3, 0, 0, 0] = z[0, 5, 0, 0, 0] = 9.0          # tie: first index wins, as torch.argmax
    lab = fill.make_label_map(2, 6, 5, 7)
    mask, md = ops.argmax_meandice(dev(z), dev(lab.to(torch.uint8)))
    want = R.argmax_mask(z)
    assert torch.equal(mask.cpu().long(), want)
    assert abs(float(md.item()) - float(R.meandice(want, lab, 8))) < 1e-12


def test_adam_against_oracle(ops):
    n = 1003
    p, g1, g2 = rnd(n, seed=1), rnd(n, seed=2) * 1e-3, rnd(n, seed=3) * 1e-3
    m, v = torch.zeros(n), torch.zeros(n)
    pp, mm, vv = p.clone(), m.clone(), v.clone()
    P, M, V = dev(p.clone()), dev(m.clone()), dev(v.clone())
    st = ops.adam_state("cuda")
    for step, g in ((1, g1), (2, g2), (3, g1)):
        lr = R.cosine_lr(1e-4, step - 1, 150)
        pp, mm, vv = R.adam_update(pp, g, mm, vv, step, lr)
        ops.adam_tick(st, 1e-4, 0.0, 150)
        ops.adam_step(P, dev(g), M, V, st)
    assert int(st.cpu()[0]) == 3
    close(P, pp, atol=1e-7, rtol=0, what="adam p")
    close(M, mm, atol=1e-9, rtol=1e-5, what="adam m")
    close(V, vv, atol=1e-12, rtol=1e-5, what="adam v")


def test_dice_metric_kernel(ops):
    """micf_dice_metric against the reference fixture (f9) and the oracle, float one-hot and uint8 class-map targets."""
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "f9_metric.npz")).items()}
    z, lab = g["z"].cuda(), g["label"]
    m1 = ops.dice_metric(z, fill.one_hot(lab.long()).cuda())
    m2 = ops.dice_metric(z, lab.cuda())
    assert float((m1.cpu() - g["metric"]).abs().max()) < 1e-6 and float((m2.cpu() - g["metric"]).abs().max()) < 1e-6
    from micformer_amd.loss.dice import MDiceLoss_Val
    lst = MDiceLoss_Val().metric(z, fill.one_hot(lab.long()).cuda())
    assert len(lst) == 2 and len(lst[0]) == 8 and abs(float(lst[1][6]) - 1.0) < 1e-7
    zz = torch.randn(2, 8, 20, 24, 28, generator=torch.Generator().manual_seed(4))
    ll = torch.randint(0, 8, (2, 20, 24, 28), generator=torch.Generator().manual_seed(5), dtype=torch.uint8)
    want = R.dice_metric(zz, fill.one_hot(ll.long()))
    assert float((ops.dice_metric(zz.cuda(), ll.cuda()).cpu() - want).abs().max()) < 1e-6


@pytest.mark.parametrize("half", [True, False])
def test_input_pipeline_tail_kernel(ops, half):
    """micf_intensity_stats + micf_input_prepare against the oracle's restatement of train.py:116-125."""
    g = torch.Generator().manual_seed(11)
    B, D, H, W = 2, 12, 10, 14
    img = torch.randn(B, 2, D, H, W, generator=g) * 40 + 100
    img[:, :, :3] = 0
    img = img.half() if half else img
    lab = torch.randint(0, 8, (B, D, H, W), generator=g, dtype=torch.uint8)
    params = torch.tensor([[1, 0, 1, 0.07, -0.03], [0, 1, 0, -0.05, 0.09]], dtype=torch.float32)
    x, l = ops.input_prepare(img.cuda(), lab.cuda(), params.cuda())
    for b in range(B):
        flips = tuple(bool(v) for v in params[b, :3])
        wx, wl = R.input_pipeline_tail(img[b], lab[b].long(), flips, float(params[b, 3]), float(params[b, 4]))
        assert float((x[b].cpu() - wx).abs().max()) < (2e-3 if half else 2e-5)
        assert torch.equal(l[b].cpu().long(), wl)
    xv, lv = ops.input_prepare(img.cuda(), None, None)          # validation transform: normalise only
    assert lv is None
    wv = torch.stack([R.normalize_intensity_nonzero(img[b]) for b in range(B)])
    assert float((xv.cpu() - wv).abs().max()) < (2e-3 if half else 2e-5)


@pytest.mark.parametrize("half", [True, False])
def test_input_tail_fused_into_patch_embedding(ops, half):
    """SURVEY 8(f) row 3 as written: micf_patch_rows_prepared (flips as index arithmetic, normalise / scale / shift as one affine map
    inside the patch gather) -- the patch-row matrices equal space-to-depth of the prepared volume bit for bit; Head(RawBatch) equals
    Head(prepared volume) incl. a non-multiple-of-4 volume (right padding), labels are flipped alike, and the patch-embedding weight
    gradient agrees; one TrainEngine graph step on a RawBatch equals the step on the prepared tensor."""
    import micformer_amd.models.MICFormer_self as M
    from micformer_amd import data
    from micformer_amd.engine import TrainEngine
    g = torch.Generator().manual_seed(21)
    B, D, H, W = 2, 30, 32, 34
    img = torch.randn(B, 2, D, H, W, generator=g) * 40 + 100
    img[:, :, :3] = 0
    img = (img.half() if half else img).cuda()
    lab = torch.randint(0, 8, (B, D, H, W), generator=g, dtype=torch.uint8).cuda()
    params = torch.tensor([[1, 0, 1, 0.07, -0.03], [0, 1, 1, -0.05, 0.09]], dtype=torch.float32).cuda()
    x, l = data.prepare_batch(img, lab, params)
    raw, l2 = data.prepare_raw_batch(img, lab, params)
    assert torch.equal(l, l2)
    rows, grid = raw.patch_rows(4)
    for m in (0, 1):
        want = ops.space_to_depth(x, (B, D, H, W), 1, 4, batch_stride=2 * D * H * W, offset=m * D * H * W)
        assert torch.equal(rows[m], want)
    head = M.Head(embed_dim=24, num_classes=8, depths=(1, 1, 1, 1))
    with torch.no_grad():
        for name, t in head.state_dict().items():
            t.copy_(fill.fill_tensor(name, t))
    head = head.cuda().eval()
    ya, yb = head(x), head(raw)
    # (the patch-row matrices are bit-equal; downstream the small grids accumulate the offset conv with fp32 atomics: run-to-run rounding)
    assert float((ya - yb).abs().max()) <= 1e-5
    ga = torch.autograd.grad(ya.square().mean(), head.swin.patch_embed.proj.weight)[0]
    gb = torch.autograd.grad(yb.square().mean(), head.swin.patch_embed.proj.weight)[0]
    # (this volume's last token grid is 1 x 1 x 2: S == 1 sampling axes, where the reference's own backward is NaN -- same pattern both ways)
    assert torch.equal(torch.isnan(ga), torch.isnan(gb))
    fin = ~torch.isnan(ga)
    if bool(fin.any()):
        assert float((ga - gb)[fin].abs().max()) <= 1e-3 * float(ga[fin].abs().max())
    h64 = head(data.RawBatch(torch.nn.functional.pad(img, (0, 30, 0, 32, 0, 34)).contiguous(), params))      # 64 x 64 x 64: finite gradients
    g64 = torch.autograd.grad(h64.square().mean(), head.swin.patch_embed.proj.weight)[0]
    x64p, _ = data.prepare_batch(torch.nn.functional.pad(img, (0, 30, 0, 32, 0, 34)).contiguous(), None, params)
    g64p = torch.autograd.grad(head(x64p).square().mean(), head.swin.patch_embed.proj.weight)[0]
    assert bool(torch.isfinite(g64).all()) and float((g64 - g64p).abs().max()) <= 1e-3 * float(g64p.abs().max())
    # validation transform (no augmentation draws) and the engine's captured step
    xv, _ = data.prepare_batch(img, None, None)
    assert float((head(xv) - head(data.RawBatch(img))).abs().max()) <= 1e-5
    x64 = torch.nn.functional.pad(img, (0, 30, 0, 32, 0, 34)).contiguous()         # (64^3: every sampling axis longer than 1)
    xp, lp = data.prepare_batch(x64, torch.nn.functional.pad(lab, (0, 30, 0, 32, 0, 34)).contiguous(), params)
    rp, _ = data.prepare_raw_batch(x64, None, params)
    losses = []
    for inp in (xp, rp):
        torch.manual_seed(3)
        h2 = M.Head(embed_dim=24, num_classes=8, depths=(1, 1, 1, 1))
        with torch.no_grad():
            for name, t in h2.state_dict().items():
                t.copy_(fill.fill_tensor(name, t))
        e = TrainEngine(h2.cuda().eval(), base_lr=1e-4, t_max=10, use_graph=True)
        losses.append([float(e.step(inp, lp)) for _ in range(2)])
    assert all(abs(a - b) <= 1e-5 for a, b in zip(*losses)) and losses[0][1] == losses[0][1], losses


@pytest.mark.parametrize("dims,C", [((1, 4, 8, 8), 96), ((2, 8, 8, 16), 48), ((1, 4, 4, 4), 384)])
def test_offset_head_pair_matches_per_op_calls(dims, C):
    """micf_offset_head_fwd / _bwd (both heads of a cross pair per launch) against the per-modality entry points they replace."""
    from micformer_amd import ops
    B, D, H, W = dims
    T = B * D * H * W
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    xs = [rnd(T, C), rnd(T, C)]
    xns = [rnd(T, C), rnd(T, C)]
    Ps = [{"conv_offset.0.weight": rnd(16, 2 * C, 3, 3, 3, sc=(54 * C) ** -0.5), "conv_offset.0.bias": rnd(16, sc=0.1),
           "conv_offset.1.norm.weight": 1 + rnd(16, sc=0.1), "conv_offset.1.norm.bias": rnd(16, sc=0.1),
           "conv_offset.3.weight": rnd(3, 16, 1, 1, 1, sc=0.3)} for _ in (0, 1)]
    eps = 1e-5
    ref = []
    for i in (0, 1):
        hid = ops.conv3_fwd(xns[i], Ps[i]["conv_offset.0.weight"], Ps[i]["conv_offset.0.bias"], dims, x2=xs[1 - i])
        flow, samp = ops.offset_sample_fwd(hid, Ps[i]["conv_offset.1.norm.weight"], Ps[i]["conv_offset.1.norm.bias"],
                                           Ps[i]["conv_offset.3.weight"], xs[1 - i], dims, eps)
        ref.append((hid, flow, samp))
    hid0 = torch.zeros(2, T, 16, device="cuda") if ops.offset_head_needs_zero(dims, C) else None
    got = ops.offset_head_fwd([{"xn": xns[i], "xa": xs[1 - i], "P": Ps[i]} for i in (0, 1)], dims, eps, hid0)
    for i in (0, 1):
        for name, a, b in zip(("hid", "flow", "xs"), got[i], ref[i]):
            assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1.0), (i, name)
    # backward
    dxs = [rnd(T, C), rnd(T, C)]
    def grads():
        return [{k: torch.zeros_like(v) for k, v in P.items()} for P in Ps]
    Gr, acc_r, dxn_r, dh_r = grads(), [rnd(T, C), rnd(T, C)], [rnd(T, C), rnd(T, C)], []
    acc_g, dxn_g = [t.clone() for t in acc_r], [t.clone() for t in dxn_r]
    for i in (0, 1):
        dh = ops.offset_sample_bwd(dxs[i], ref[i][0], Ps[i]["conv_offset.1.norm.weight"], Ps[i]["conv_offset.1.norm.bias"],
                                   Ps[i]["conv_offset.3.weight"], xs[1 - i], ref[i][1], acc_r[1 - i], Gr[i]["conv_offset.1.norm.weight"],
                                   Gr[i]["conv_offset.1.norm.bias"], Gr[i]["conv_offset.3.weight"], dims, eps)
        ops.conv3_bwd_data(dh, Ps[i]["conv_offset.0.weight"], dims, C, C, dx1=dxn_r[i], dx2=acc_r[1 - i], acc1=True, acc2=True)
        dh_r.append(dh)
    Gg = grads()
    dh_g = ops.offset_head_bwd([{"dxs": dxs[i], "hid": ref[i][0], "flow": ref[i][1], "xa": xs[1 - i], "P": Ps[i], "G": Gg[i],
                                 "dxa": acc_g[1 - i], "dxn": dxn_g[i]} for i in (0, 1)], dims, eps)
    torch.cuda.synchronize()
    def close(a, b, what):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1.0), what
    for i in (0, 1):
        close(dh_g[i], dh_r[i], f"dhid {i}")
        close(acc_g[i], acc_r[i], f"dxa {i}")
        close(dxn_g[i], dxn_r[i], f"dxn {i}")
        for k in ("conv_offset.1.norm.weight", "conv_offset.1.norm.bias", "conv_offset.3.weight"):
            close(Gg[i][k], Gr[i][k], f"{k} {i}")


@pytest.mark.parametrize("C", [48, 192])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_conv3_prepared_layouts_give_the_same_result(C, mode):
    """micf_conv3_weight_prep_grouped (all layouts of a weight in one launch, once per step) against the calls' own re-layout."""
    from micformer_amd import ops
    dims = (1, 4, 8, 8)
    T = 4 * 8 * 8
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    x1, x2, dy = rnd(T, C), rnd(T, C), rnd(T, 16)
    w, b = rnd(16, 2 * C, 3, 3, 3, sc=(54 * C) ** -0.5), rnd(16, sc=0.1)
    prev = ops.compute_dtype()
    ops.set_compute_dtype(mode)
    try:
        y0 = ops.conv3_fwd(x1, w, b, dims, x2=x2)
        d0 = ops.conv3_bwd_data(dy, w, dims, C, C)
        wp = w.clone()
        wp._micf_c3f, wp._micf_c3b = ops.conv3_prepared_like(wp)
        ops.Conv3PrepPlan([(wp, wp._micf_c3f, wp._micf_c3b)]).launch()
        y1 = ops.conv3_fwd(x1, wp, b, dims, x2=x2)
        d1 = ops.conv3_bwd_data(dy, wp, dims, C, C)
    finally:
        ops.set_compute_dtype(prev)
    tol = 0 if mode == "bf16" else 0            # same kernels, same operands: only the atomic accumulation order may differ
    assert float((y1 - y0).abs().max()) <= 1e-5 * float(y0.abs().max())
    assert torch.equal(d1[0], d0[0]) and torch.equal(d1[1], d0[1])


def test_grouped_wgrad_column_blocks(ops):
    """micf_wgrad_item.ldw: a layer on a concatenation [a1 | a2] as two grouped items, each a column block of one dw (long layers:
    split + reduce through the workspace; short ones: direct) == the two-input micf_linear_bwd_weight, accumulated."""
    for M in (4096, 512):
        N, k1, k2 = 48, 96, 32
        dy, a1, a2 = dev(rnd(M, N, seed=1)), dev(rnd(M, k1, seed=2)), dev(rnd(M, k2, seed=3))
        w0, b0 = rnd(N, k1 + k2, seed=4), rnd(N, seed=5)
        dw_ref, db_ref = dev(w0.clone()), dev(b0.clone())
        ops.linear_bwd_weight(dy, a1, dw_ref, db_ref, a2)
        dw, db = dev(w0.clone()), dev(b0.clone())
        ops.linear_bwd_weight_grouped([(dy, a1, dw[:, :k1], db, None, 0), (dy, a2, dw[:, k1:], None, None, 0)])
        close(dw, dw_ref.cpu(), rtol=2e-5, what=f"column-block dw M={M}")
        close(db, db_ref.cpu(), rtol=2e-5, what=f"column-block db M={M}")
