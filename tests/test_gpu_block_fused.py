"""GPU parity of the fused window-local block kernels (csrc/block_fwd.hip, block_bwd.hip) against the per-op entry points of
round 1 (each of which is pinned against the oracle / the reference goldens in test_gpu_ops.py): every tensor the fused
forward saves and every gradient the fused backward emits, self and cross form, one and two groups per launch, every tile
width, partial tiles, with and without DropPath scales.  fp32 mode: same arithmetic, different summation order -> 1e-5-class."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micformer_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).cuda()


def make_params(C, hidden, attn, seed):
    k = 1.0 / math.sqrt(C)
    kh = 1.0 / math.sqrt(hidden)
    P = {"norm1.weight": 1 + rnd((C,), seed + 1, 0.1), "norm1.bias": rnd((C,), seed + 2, 0.1),
         f"{attn}.q.weight": rnd((C, C), seed + 3, k), f"{attn}.q.bias": rnd((C,), seed + 4, 0.1),
         f"{attn}.kv.weight": rnd((2 * C, C), seed + 5, k), f"{attn}.kv.bias": rnd((2 * C,), seed + 6, 0.1),
         f"{attn}.proj.weight": rnd((C, C), seed + 7, k), f"{attn}.proj.bias": rnd((C,), seed + 8, 0.1),
         "norm2.weight": 1 + rnd((C,), seed + 9, 0.1), "norm2.bias": rnd((C,), seed + 10, 0.1),
         "mlp.fc1.weight": rnd((hidden, C), seed + 11, k), "mlp.fc1.bias": rnd((hidden,), seed + 12, 0.1),
         "mlp.fc2.weight": rnd((C, hidden), seed + 13, kh), "mlp.fc2.bias": rnd((C,), seed + 14, 0.1)}
    return P


def ref_fwd(ops, x, kvsrc, P, attn, s1, s2, dims, heads, eps, scale):
    """The round-1 launch sequence (functional.SelfBlockFn / CrossBlockFn tail)."""
    B, D, H, W = dims
    rps = D * H * W
    xn, m1, r1 = ops.layernorm_fwd(x, P["norm1.weight"], P["norm1.bias"], eps)
    q = ops.linear_fwd(xn, P[f"{attn}.q.weight"], P[f"{attn}.q.bias"])
    kv = ops.linear_fwd(kvsrc if kvsrc is not None else xn, P[f"{attn}.kv.weight"], P[f"{attn}.kv.bias"])
    o = ops.window_attn_fwd(q, kv, dims, heads, (2, 2, 2), scale)
    x1 = ops.linear_fwd(o, P[f"{attn}.proj.weight"], P[f"{attn}.proj.bias"], resid=x, dp_scale=s1, rows_per_sample=rps)
    xn2, m2, r2 = ops.layernorm_fwd(x1, P["norm2.weight"], P["norm2.bias"], eps)
    g, h = ops.linear_fwd(xn2, P["mlp.fc1.weight"], P["mlp.fc1.bias"], act=1, want_pre=True)
    y = ops.linear_fwd(g, P["mlp.fc2.weight"], P["mlp.fc2.bias"], resid=x1, dp_scale=s2, rows_per_sample=rps)
    return {"y": y, "xn": xn, "q": q, "kv": kv, "o": o, "x1": x1, "xn2": xn2, "h": h, "g": g,
            "stats": torch.stack([m1, r1, m2, r2])}


def ref_bwd(ops, dy, x, sv, P, attn, s1, s2, dims, heads, scale, cross):
    B, D, H, W = dims
    rps = D * H * W
    C = x.shape[1]
    m1, r1, m2, r2 = sv["stats"]
    dh = ops.linear_bwd_data(dy, P["mlp.fc2.weight"], dp_scale=s2, rows_per_sample=rps, pre_act=sv["h"])
    dxn2 = ops.linear_bwd_data(dh, P["mlp.fc1.weight"])
    dg2, db2 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx1 = ops.layernorm_bwd(dxn2, sv["x1"], m2.contiguous(), r2.contiguous(), P["norm2.weight"], dg2, db2, add=dy)
    do = ops.linear_bwd_data(dx1, P[f"{attn}.proj.weight"], dp_scale=s1, rows_per_sample=rps)
    dq, dkv = ops.window_attn_bwd(sv["q"], sv["kv"], do, dims, heads, (2, 2, 2), scale)
    dxq = ops.linear_bwd_data(dq, P[f"{attn}.q.weight"])
    dxk = ops.linear_bwd_data(dkv, P[f"{attn}.kv.weight"])
    out = {"dx1": dx1, "dh": dh, "dq": dq, "dkv": dkv, "dg2": dg2, "db2": db2}
    if cross:
        out["dx"], out["dxs"] = dxq, dxk
    else:
        dg1, db1 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        out["dx"] = ops.layernorm_bwd(dxq + dxk, x, m1.contiguous(), r1.contiguous(), P["norm1.weight"], dg1, db1, add=dx1)
        out["dg1"], out["db1"] = dg1, db1
    return out


def check(name, got, want, tol, errs):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    if got.shape != want.shape:
        errs.append(f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}")
        return
    scale = max(float(want.abs().max()), 1e-20)
    err = float((got - want).abs().max())
    if not (err <= tol * scale + 1e-7):
        bad = (got - want).abs() > tol * scale + 1e-7
        errs.append(f"{name}: max abs err {err:.3e} (scale {scale:.3e}), {int(bad.sum())}/{bad.numel()} bad, first bad index {bad.nonzero()[0].tolist() if bad.any() else None}")


CASES = [  # (B, D, H, W, C, heads)
    (2, 4, 4, 4, 48, 3),      # TM 32, 4 full tiles
    (1, 2, 2, 2, 48, 3),      # one window in a 32-token tile: masked rows
    (1, 4, 6, 4, 96, 6),      # TM 16
    (1, 2, 6, 2, 96, 3),      # head_dim 32, 3 windows: partial last tile
    (2, 4, 4, 2, 192, 12),    # TM 16
    (1, 4, 2, 2, 192, 6),     # head_dim 32 at C 192
    (2, 4, 4, 4, 384, 24),    # few-token decomposition (block_wide.hip): the base model's 4^3 stage
    (1, 2, 6, 2, 384, 12),    # C 384 / head_dim 32 (round 6: tile kernels), 3 windows: masked rows in the last 16-token tile
    (1, 10, 10, 8, 384, 12),  # ... the large model's third stage (config 4): 50 tiles per group
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("ngroups", [1, 2])
@pytest.mark.parametrize("save_h", [True, False])
def test_fused_block_matches_per_op_path(ops, case, cross, ngroups, save_h, hook):
    """save_h True: the default, the fc1 pre-activation is stored; False: the option "block_recompute_h" -- the tile kernels do not
    store it, the backward rebuilds it from xn2 (micf_block_recomputes_h; the few-token decomposition at C = 384 always stores)."""
    B, D, H, W, C, heads = case
    wide = not ops.block_fuses_sampler(C, heads)       # the few-token decomposition (block_wide.hip): C = 384 with head_dim 16
    if not save_h:
        if wide:
            pytest.skip("the few-token decomposition stores h either way: covered by save_h True")
        hook("block_recompute_h", 1)
    assert ops.block_recomputes_h(C, heads) == (not wide and not save_h)
    dims = (B, D, H, W)
    T = B * D * H * W
    hidden = 4 * C
    tm = ops.block_tile_tokens(dims, C, heads, hidden)
    assert tm in (16, 32, 64), f"shape {case} should be handled by the fused kernels"
    attn = "cross_attn" if cross else "self_attn"
    eps, scale = 1e-5, (C // heads) ** -0.5
    groups_f, refs, extra = [], [], []
    for gi in range(ngroups):
        seed = 1000 * gi + 17 * C + T
        P = make_params(C, hidden, attn, seed)
        x = rnd((T, C), seed + 50)
        kvsrc = rnd((T, C), seed + 51) if cross else None
        if B == 2:
            s1 = torch.tensor([0.0, 1.25] if gi == 0 else [1.25, 1.25]).cuda()
            s2 = torch.tensor([1.25, 0.0]).cuda()
        else:
            s1 = None if gi == 0 else torch.tensor([1.25]).cuda()
            s2 = None if gi == 0 else torch.tensor([0.8]).cuda()
        groups_f.append({"x": x, "kvsrc": kvsrc, "P": P, "attn": attn, "s1": s1, "s2": s2, "want_xn": True})
        refs.append(ref_fwd(ops, x, kvsrc, P, attn, s1, s2, dims, heads, eps, scale))
        extra.append((x, kvsrc, P, s1, s2, seed))
    outs = ops.block_fwd(groups_f, dims, C, heads, eps, scale)
    errs = []
    for gi, (o, r) in enumerate(zip(outs, refs)):
        assert (o["h"] is None) == ops.block_recomputes_h(C, heads)
        for k in ("xn", "q", "kv", "o", "x1", "xn2", "h", "g", "y", "stats"):
            if o[k] is not None:
                check(f"fwd g{gi} {k}", o[k], r[k], 2e-5, errs)
    assert not errs, "\n".join(errs)

    groups_b, brefs = [], []
    for gi, (o, (x, kvsrc, P, s1, s2, seed)) in enumerate(zip(outs, extra)):
        dy = rnd((T, C), seed + 60)
        groups_b.append({"dy": dy, "x": x, "x1": o["x1"], "stats": o["stats"], "q": o["q"], "kv": o["kv"], "h": o["h"], "xn2": o["xn2"], "P": P,
                         "attn": attn, "s1": s1, "s2": s2, "cross": cross, "want_copy": cross})
        brefs.append(ref_bwd(ops, dy, x, refs[gi], P, attn, s1, s2, dims, heads, scale, cross))
    bouts = ops.block_bwd(groups_b, dims, C, heads, scale)
    for gi, (o, r) in enumerate(zip(bouts, brefs)):
        for k in ("dh", "dx1", "dq", "dkv", "dx"):
            check(f"bwd g{gi} {k}", o[k], r[k], 5e-5, errs)
        if cross:
            check(f"bwd g{gi} dxs", o["dxs"], r["dxs"], 5e-5, errs)
            check(f"bwd g{gi} dx1_copy", o["dx1_copy"], r["dx1"], 5e-5, errs)
        p2 = o["ln2_part"].sum(0)
        check(f"bwd g{gi} dgamma2", p2[:C], r["dg2"], 2e-4, errs)
        check(f"bwd g{gi} dbeta2", p2[C:], r["db2"], 2e-4, errs)
        if not cross:
            p1 = o["ln1_part"].sum(0)
            check(f"bwd g{gi} dgamma1", p1[:C], r["dg1"], 2e-4, errs)
            check(f"bwd g{gi} dbeta1", p1[C:], r["db1"], 2e-4, errs)
        # the partial rows feed the round-1 finish kernel unchanged
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        ops.layernorm_bwd_finish([(o["ln2_part"], o["tiles"], C, dg, db)])
        check(f"bwd g{gi} finish dgamma2", dg, r["dg2"], 2e-4, errs)
        check(f"bwd g{gi} finish dbeta2", db, r["db2"], 2e-4, errs)
    assert not errs, "\n".join(errs)


def test_unsupported_shapes_are_reported(ops):
    assert ops.block_tile_tokens((1, 3, 4, 4), 48, 3, 192) == 0          # odd token grid: pad-to-window path
    assert ops.block_tile_tokens((1, 4, 4, 4), 24, 3, 96) == 0           # head_dim 8 (tiny config)
    assert ops.block_tile_tokens((1, 4, 4, 4), 768, 24, 3072) == 0       # C = 768 (large model, 5^3 stage): per-op path
    assert ops.block_tile_tokens((1, 4, 4, 4), 384, 24, 1536) == 16      # few-token decomposition
    assert ops.block_tile_tokens((2, 32, 32, 32), 48, 3, 192) == 32


@pytest.mark.parametrize("case", [(2, 4, 4, 4, 48, 3), (1, 4, 6, 4, 96, 6), (2, 4, 4, 2, 192, 12), (2, 4, 4, 4, 384, 24), (1, 4, 6, 4, 384, 12)])
def test_fused_block_bf16_mode_is_close(ops, case):
    """bf16 matrix-core operands (fp32 accumulate, fp32 everywhere else): 8-bit mantissas on the GEMM inputs only."""
    B, D, H, W, C, heads = case
    dims = (B, D, H, W)
    T = B * D * H * W
    attn = "self_attn"
    P = make_params(C, 4 * C, attn, 5)
    x = rnd((T, C), 7)
    eps, scale = 1e-5, (C // heads) ** -0.5
    ref = ref_fwd(ops, x, None, P, attn, None, None, dims, heads, eps, scale)
    ops.set_compute_dtype("bf16")
    try:
        o = ops.block_fwd([{"x": x, "kvsrc": None, "P": P, "attn": attn, "s1": None, "s2": None}], dims, C, heads, eps, scale)[0]
        dy = rnd((T, C), 9)
        b = ops.block_bwd([{"dy": dy, "x": x, "x1": o["x1"], "stats": o["stats"], "q": o["q"], "kv": o["kv"], "h": o["h"], "xn2": o["xn2"], "P": P,
                            "attn": attn, "s1": None, "s2": None, "cross": False}], dims, C, heads, scale)[0]
    finally:
        ops.set_compute_dtype("fp32")
    errs = []
    for k in ("q", "kv", "o", "x1", "h", "y"):
        if o[k] is not None:
            check(f"bf16 fwd {k}", o[k], ref[k], 2e-2, errs)
    rb = ref_bwd(ops, dy, x, ref, P, attn, None, None, dims, heads, scale, False)
    for k in ("dh", "dx1", "dq", "dkv", "dx"):
        check(f"bf16 bwd {k}", b[k], rb[k], 3e-2, errs)
    assert float((o["y"] - ref["y"]).abs().max()) > 0                      # it really is a different arithmetic
    assert not errs, "\n".join(errs)
    # storage: the tile-per-workgroup kernels (everything but C = 384 / head_dim 16) leave what only matrix cores / the attention
    # backward re-read as bf16
    st16 = bool(ops.block_fuses_sampler(C, heads))
    assert st16 == (not (C == 384 and C // heads == 16))
    for k in ("xn", "q", "kv", "o", "xn2", "g"):
        assert (o[k].dtype == torch.bfloat16) == st16, k
    for k in ("dq", "dkv", "dh", "dx1"):
        assert (b[k].dtype == torch.bfloat16) == st16, k
    assert o["x1"].dtype == o["y"].dtype == b["dx"].dtype == torch.float32
    if st16:
        assert torch.equal(b["dy16"], dy.bfloat16())                        # the bf16 copy of dy is its RNE rounding
    else:
        assert b["dy16"] is None


def test_bf16_storage_cross_block_operand_copies(ops):
    """bf16 storage of a cross block: kvs16 is the RNE rounding of the K/V source, xn is written although the caller did not ask
    for it (the q weight gradient pairs two bf16 operands), and the five weight gradients from the stored operands equal the
    fp32-operand weight gradients of the same (rounded) values."""
    B, D, H, W, C, heads = 2, 4, 4, 4, 96, 6
    dims, T, attn = (B, D, H, W), 128, "cross_attn"
    P = make_params(C, 4 * C, attn, 11)
    x, kvsrc, dy = rnd((T, C), 12), rnd((T, C), 13), rnd((T, C), 14)
    eps, scale = 1e-5, (C // heads) ** -0.5
    ops.set_compute_dtype("bf16")
    try:
        o = ops.block_fwd([{"x": x, "kvsrc": kvsrc, "P": P, "attn": attn, "s1": None, "s2": None, "want_xn": False}], dims, C, heads, eps,
                          scale)[0]
        b = ops.block_bwd([{"dy": dy, "x": None, "x1": o["x1"], "stats": o["stats"], "q": o["q"], "kv": o["kv"], "h": o["h"], "xn2": o["xn2"], "P": P,
                            "attn": attn, "s1": None, "s2": None, "cross": True, "want_copy": True}], dims, C, heads, scale)[0]
        assert torch.equal(o["kvs16"], kvsrc.bfloat16()) and o["xn"] is not None and o["xn"].dtype == torch.bfloat16
        assert b["dx1_copy"].dtype == torch.float32 and float((b["dx1_copy"] - b["dx1"].float()).abs().max()) <= 2 ** -8 * float(b["dx1_copy"].abs().max())
        errs = []
        for dyo, a, name in ((b["dy16"], o["g"], "fc2"), (b["dh"], o["xn2"], "fc1"), (b["dx1"], o["o"], "proj"), (b["dq"], o["xn"], "q"),
                             (b["dkv"], o["kvs16"], "kv")):
            N, K = dyo.shape[1], a.shape[1]
            w16, w32 = torch.zeros(N, K, device="cuda"), torch.zeros(N, K, device="cuda")
            ops.linear_bwd_weight_grouped([(dyo, a, w16, None, None, 0)])
            ops.linear_bwd_weight_grouped([(dyo.float(), a.float(), w32, None, None, 0)])
            check(f"wgrad {name}", w16, w32, 1e-5, errs)
        assert not errs, "\n".join(errs)
    finally:
        ops.set_compute_dtype("fp32")


@pytest.mark.parametrize("rows,cols", [(48, 48), (96, 48), (192, 768), (33, 70), (130, 4)])
def test_weight_prep_outputs(ops, rows, cols):
    """micf_weight_prep_grouped: the fp32 copies are permutations (bit-exact); the bf16 copies equal torch's round-to-nearest-even."""
    w = torch.randn(rows, cols, device="cuda")
    wt, wc = ops.shadow_like(w, True, torch.float32), ops.shadow_like(w, False, torch.float32)
    w16, wt16 = ops.shadow_like(w, False, torch.bfloat16), ops.shadow_like(w, True, torch.bfloat16)
    only_t = ops.shadow_like(w, True, torch.bfloat16)
    ops.WeightPrepPlan([(w, wc, wt), (w, w16, wt16), (w, None, only_t)]).launch()
    torch.cuda.synchronize()
    assert torch.equal(wc, w) and torch.equal(wt, w.t().contiguous())
    assert torch.equal(w16, w.to(torch.bfloat16))
    assert torch.equal(wt16, w.t().contiguous().to(torch.bfloat16)) and torch.equal(only_t, wt16)


@pytest.mark.parametrize("rows,cols", [(48, 48), (96, 48), (192, 768), (768, 384)])
def test_weight_prep_blocked_outputs(ops, rows, cols):
    """bf16 = 2: the same bf16 values in the K16-blocked order the block kernels stream ([R/16][C/16][16][16]); shapes that are
    not multiples of 16 are refused."""
    w = torch.randn(rows, cols, device="cuda")
    w16, wt16 = ops.shadow_like(w, False, torch.bfloat16), ops.shadow_like(w, True, torch.bfloat16)
    ops.WeightPrepPlan([(w, w16, wt16)], blocked=True).launch()
    torch.cuda.synchronize()
    assert torch.equal(w16, ops.blocked16(w.to(torch.bfloat16)))
    assert torch.equal(wt16, ops.blocked16(w.t().contiguous().to(torch.bfloat16)))
    w32, wt32 = ops.shadow_like(w, False, torch.float32), ops.shadow_like(w, True, torch.float32)
    ops.WeightPrepPlan([(w, w32, wt32)], blocked=True).launch()                  # bf16 = 3: the fp32 mode's blocked copies
    torch.cuda.synchronize()
    assert torch.equal(w32, ops.blocked16(w)) and torch.equal(wt32, ops.blocked16(w.t().contiguous()))
    bad = torch.randn(40, 48, device="cuda")
    with pytest.raises(RuntimeError):
        ops.WeightPrepPlan([(bad, ops.shadow_like(bad, False, torch.bfloat16), None)], blocked=True).launch()


def test_inference_shadow_cache_follows_weight_updates(ops):
    """Without an engine the shadow weights are cached on the weight tensor (keyed by torch's version counter, PARAM_EPOCH and the
    arithmetic mode), BOTH orientations prepared by one launch, and re-derived when torch code writes the weight -- in either grad
    mode (inside autograd.Function.forward grad mode is always off: the drop-in's training step prepares every weight once)."""
    ops.set_compute_dtype("bf16")
    try:
        keys = ("self_attn.q.weight", "self_attn.kv.weight", "self_attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")
        P = {k: torch.randn(96, 48, device="cuda") for k in keys}
        with torch.no_grad():
            a = ops.block_weights(P, "self_attn", backward=False)
            b = ops.block_weights(P, "self_attn", backward=False)
            assert all(a[f].data_ptr() == b[f].data_ptr() for f in a)                  # cached, not re-made
            assert torch.equal(a["wq"], ops.blocked16(P["self_attn.q.weight"].to(torch.bfloat16)))
            P["self_attn.q.weight"].mul_(2.0)                                          # in-place update: version counter moves
            c = ops.block_weights(P, "self_attn", backward=False)
            torch.cuda.synchronize()
            assert torch.equal(c["wq"], ops.blocked16(P["self_attn.q.weight"].to(torch.bfloat16)))
            assert c["wkv"].data_ptr() == a["wkv"].data_ptr()                          # untouched weights keep their copies
        from micformer_amd import _lib
        n0 = _lib.LAUNCHES[0]
        g = ops.block_weights(P, "self_attn", backward=False)                          # autograd recording: the same cache
        t = ops.block_weights(P, "self_attn", backward=True)                           # ... and the transposed copies came with it
        assert g["wq"].data_ptr() == c["wq"].data_ptr() and _lib.LAUNCHES[0] == n0
        assert torch.equal(t["wqt"], ops.blocked16(P["self_attn.q.weight"].t().contiguous().to(torch.bfloat16)))
        ops.PARAM_EPOCH[0] += 1                                                        # a write torch cannot see (.data, a kernel)
        assert ops.block_weights(P, "self_attn", backward=False)["wq"].data_ptr() != c["wq"].data_ptr()
    finally:
        ops.set_compute_dtype("fp32")


def test_persistent_probe_matches_the_launch(ops):
    """micf_block_fwd_persistent_probe (the round-5 measurement of a persistent 8^3 stage, tools/bench_persist.py): N passes of the
    forward tile body inside one launch with device-wide barriers give the launch's outputs, no barrier times out, and shapes
    outside the probe's (base 8^3, bf16) are refused."""
    from micformer_amd import _lib
    B, n, C, heads = 2, 8, 192, 12
    dims, T = (B, n, n, n), B * n ** 3
    ops.set_compute_dtype("bf16")
    try:
        gs = [{"x": rnd((T, C), gi), "kvsrc": None, "P": make_params(C, 4 * C, "self_attn", 100 * gi), "attn": "self_attn",
               "s1": None, "s2": None} for gi in range(2)]
        want = [o["y"].clone() for o in ops.block_fwd(gs, dims, C, heads, 1e-5, 0.25)]
        sync = torch.ones(2, dtype=torch.int32, device="cuda")
        got = ops.block_fwd(gs, dims, C, heads, 1e-5, 0.25, persist_probe=(5, sync))
        torch.cuda.synchronize()
        assert sync.tolist() == [4 * 128, 0]                      # 4 barriers x 128 workgroups arrived, none timed out
        for a, b in zip(got, want):
            assert torch.equal(a["y"], b)
        small = [{"x": rnd((2 * 64, 96), 3), "kvsrc": None, "P": make_params(96, 384, "self_attn", 7), "attn": "self_attn",
                  "s1": None, "s2": None}]
        with pytest.raises(_lib.MicfError):
            ops.block_fwd(small, (2, 4, 4, 4), 96, 6, 1e-5, 0.25, persist_probe=(2, sync))
    finally:
        ops.set_compute_dtype("fp32")


@pytest.mark.parametrize("case", [(2, 4, 4, 4, 48, 3), (1, 2, 2, 2, 48, 3), (1, 4, 6, 4, 96, 6), (2, 4, 4, 2, 192, 12), (2, 8, 8, 8, 192, 12),
                                  (1, 4, 6, 4, 384, 12)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("save", [True, False])
def test_next_layernorm_as_the_forward_epilogue(ops, case, mode, save):
    """micf_block_fwd_group.nln_g: the LayerNorm the next block applies to y (a cross block's norm1), written by the launch that
    writes y, against micf_layernorm_fwd on that y: same arithmetic, 1e-6-class; the [T, 16] buffer it is asked to clear is clear;
    two groups with different gains, tile kernels (C = 96 / 192, fp32-mode C = 48), wave-private kernels (bf16 C = 48), training
    and inference form.  The other outputs of the launch do not change."""
    B, D, H, W, C, heads = case
    dims, T = (B, D, H, W), B * D * H * W
    eps, scale = 1e-5, (C // heads) ** -0.5
    ops.set_compute_dtype(mode)
    try:
        groups, plain = [], []
        for gi in range(2):
            P = make_params(C, 4 * C, "self_attn", 300 * gi + C)
            x = rnd((T, C), 300 * gi + 7)
            gam, bet = 1 + rnd((C,), 300 * gi + 8, 0.2), rnd((C,), 300 * gi + 9, 0.2)
            z16 = torch.full((T, 16), 3.0, device="cuda")
            base = {"x": x, "kvsrc": None, "P": P, "attn": "self_attn", "s1": None, "s2": None}
            plain.append(dict(base))
            groups.append(dict(base, next_ln=(gam, bet, z16 if gi == 0 else None)))
        ref = ops.block_fwd(plain, dims, C, heads, eps, scale, save=save)
        out = ops.block_fwd(groups, dims, C, heads, eps, scale, save=save)
        errs = []
        for gi, (o, r, gd) in enumerate(zip(out, ref, groups)):
            for k in ("y", "x1", "q", "g"):
                if r[k] is not None:
                    assert torch.equal(o[k], r[k]), f"g{gi} {k} changed with the epilogue"
            yn, mean, rstd = ops.layernorm_fwd(o["y"], gd["next_ln"][0], gd["next_ln"][1], eps)
            check(f"g{gi} nln", o["nln"][0], yn, 2e-6, errs)
            check(f"g{gi} mean", o["nln"][1], mean, 2e-6, errs)
            check(f"g{gi} rstd", o["nln"][2], rstd, 2e-6, errs)
        assert not errs, "\n".join(errs)
        assert float(groups[0]["next_ln"][2].abs().max()) == 0.0
    finally:
        ops.set_compute_dtype("fp32")


def test_next_layernorm_epilogue_is_refused_on_the_few_token_path(ops):
    from micformer_amd._lib import MicfError
    C, heads, dims = 384, 24, (2, 4, 4, 4)
    T = 128
    P = make_params(C, 4 * C, "self_attn", 5)
    g = {"x": rnd((T, C), 1), "kvsrc": None, "P": P, "attn": "self_attn", "s1": None, "s2": None,
         "next_ln": (1 + rnd((C,), 2, 0.1), rnd((C,), 3, 0.1), None)}
    with pytest.raises(MicfError):
        ops.block_fwd([g], dims, C, heads, 1e-5, (C // heads) ** -0.5)
