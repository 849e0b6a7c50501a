"""Sliding-window inference (reference call site utils.py:226-240).

MONAI is not under /root/reference, so the referee is the CPU restatement oracle/micformer_ref.py::sliding_window_inference
(parity with MONAI itself is unpinned, as its header says).  CPU: the scan-start arithmetic of the host module.
GPU: the device path (window crop, fp32 accumulate + visit count, normalise kernels) against the oracle with (a) a synthetic
window-position-dependent predictor -- so a wrong start, a wrong count or a missed window changes the result -- and
(b) the tiny MicFormer Head itself as the predictor.
"""
import pytest
import torch

from oracle import micformer_ref as R


@pytest.mark.parametrize("L,roi,overlap", [(128, 128, 0.5), (100, 128, 0.5), (256, 128, 0.5), (512, 128, 0.5), (200, 128, 0.5),
                                           (129, 128, 0.5), (70, 32, 0.25), (33, 32, 0.9), (96, 32, 0.0)])
def test_scan_starts_match_the_oracle(L, roi, overlap):
    from micformer_amd.inference import sliding_window_starts
    assert sliding_window_starts(L, roi, overlap) == R.sliding_window_starts(L, roi, overlap)
    s = sliding_window_starts(L, roi, overlap)
    assert s[0] == 0 and (L <= roi or s[-1] == L - roi)            # the volume is covered end to end
    assert all(b - a <= roi for a, b in zip(s, s[1:]))             # no gaps between consecutive windows


def test_base_config5_window_count():
    """BASELINE config 5: 512 x 512 x 256 whole-heart volume, roi 128^3, overlap 0.5 -> 7 * 7 * 3 = 147 windows."""
    from micformer_amd.inference import sliding_window_starts as st
    assert len(st(512, 128)) * len(st(512, 128)) * len(st(256, 128)) == 147


def _ramp_predictor(x):
    """K = 3 channels: a pointwise function of the window content plus a ramp in WINDOW coordinates."""
    n, c, d, h, w = x.shape
    zz = torch.arange(d, device=x.device, dtype=torch.float32).view(1, d, 1, 1) / d
    yy = torch.arange(h, device=x.device, dtype=torch.float32).view(1, 1, h, 1) / h
    xx = torch.arange(w, device=x.device, dtype=torch.float32).view(1, 1, 1, w) / w
    a, b = x[:, 0].float(), x[:, 1].float()
    return torch.stack([2 * a - b + zz, a * b + yy * xx, torch.tanh(a) + zz - xx], 1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,roi,overlap,sw", [((2, 2, 40, 24, 56), (16, 16, 16), 0.5, 1), ((2, 2, 40, 24, 56), (16, 16, 16), 0.5, 7),
                                                  ((1, 2, 20, 12, 37), (16, 16, 16), 0.25, 3), ((1, 2, 33, 16, 16), (16, 16, 16), 0.5, 4),
                                                  ((1, 2, 24, 40, 17), (16, 24, 8), 0.5, 2)])
def test_device_path_matches_oracle_with_synthetic_predictor(shape, roi, overlap, sw):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micformer_amd.inference import sliding_window_inference
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    want = R.sliding_window_inference(x, _ramp_predictor, roi=roi, overlap=overlap)
    got = sliding_window_inference(x.cuda(), roi, sw, _ramp_predictor, overlap=overlap)
    assert got.shape == want.shape and got.dtype == torch.float32
    assert float((got.cpu() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
def test_tiny_head_as_predictor_matches_oracle():
    """End to end: eval-mode tiny MicFormer (embed 24, depths 1-1-1-1) on 32^3 windows of a 48 x 32 x 64 volume."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import fill
    from oracle.shapes import filled_params
    import micformer_amd.models.MICFormer_self as M
    from micformer_amd.inference import sliding_window_inference
    cfg = R.Cfg(embed_dim=24, depths=(1, 1, 1, 1))
    P = filled_params(cfg)
    head = M.Head(embed_dim=24, num_classes=8, depths=(1, 1, 1, 1))
    with torch.no_grad():
        for name, t in head.state_dict().items():
            t.copy_(fill.fill_tensor(name, t))
    head = head.cuda().eval()
    x = fill.make_volume(1, 48, 32, 64, "SW.x")
    with torch.no_grad():
        want = R.sliding_window_inference(x, lambda w: R.head_forward(P, w, cfg), roi=(32, 32, 32), overlap=0.5)
        with torch.autocast("cuda", dtype=torch.float16):            # utils.py:236-238 runs the predictor under autocast
            got = sliding_window_inference(x.cuda(), (32, 32, 32), 4, head, overlap=0.5)
            got_g = sliding_window_inference(x.cuda(), (32, 32, 32), 3, head, overlap=0.5, graph=True)   # HIP-graph replayed predictor
    assert got.shape == want.shape
    assert float((got_g - got).abs().max()) <= 1e-5
    assert float((got.cpu() - want).abs().max()) <= 1e-4
    assert torch.equal(got.argmax(1).cpu(), want.argmax(1)) or \
        float((got.argmax(1).cpu() != want.argmax(1)).float().mean()) < 1e-3


def _head(E, depths):
    from oracle import fill
    import micformer_amd.models.MICFormer_self as M
    head = M.Head(embed_dim=E, num_classes=8, depths=depths)
    with torch.no_grad():
        for name, t in head.state_dict().items():
            t.copy_(fill.fill_tensor(name, t))
    return head.cuda().eval()


@pytest.mark.gpu
def test_fused_accumulate_epilogue_matches_the_separate_launches():
    """SURVEY 8(f) row 1 as written: the accumulate / count epilogue inside the head's logits store (Head.forward_accumulate ->
    micf_head_tail_col2im_sw, window origins as a device argument) against the unfused path (prediction tensor + batched accumulate
    launch), eager and graph-replayed, fp32."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import fill
    from micformer_amd import inference as I
    head = _head(48, (1, 1, 1, 1))
    x = fill.make_volume(2, 96, 64, 80, "SW.fused").cuda()
    with torch.no_grad():
        fused = I.sliding_window_inference(x, (64, 64, 64), 3, head, overlap=0.5)
        fused_g = I.sliding_window_inference(x, (64, 64, 64), 2, head, overlap=0.5, graph=True)
        I.FUSE_ACCUMULATE = False
        try:
            plain = I.sliding_window_inference(x, (64, 64, 64), 3, head, overlap=0.5)
        finally:
            I.FUSE_ACCUMULATE = True
    scale = float(plain.abs().max())
    # (the same fp32 terms summed in another order -- atomics, and the head's patch sum folded into the accumulation)
    assert float((fused - plain).abs().max()) <= 1e-5 * max(1.0, scale)
    assert float((fused_g - plain).abs().max()) <= 1e-5 * max(1.0, scale)


@pytest.mark.gpu
def test_config5_whole_heart_volume_full_size():
    """BASELINE config 5 at full size: base Head on a 512 x 512 x 256 two-modality volume, roi 128^3, overlap 0.5 -> 147 windows,
    HIP-graph replayed predictor with the fused accumulate epilogue.  Size-independent properties: any sw_batch_size gives the
    same volume (the network is batch-independent; fp32: to accumulation-order rounding), every voxel is finite, the region only
    ONE window covers (the first 64 voxels of every axis) equals the direct prediction of that window, and the bf16 predictor (the
    reference wraps its predictor in autocast, utils.py:236-238) stays within the bf16 logits gate of the fp32 one."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micformer_amd import inference as I
    head = _head(48, (2, 2, 6, 2))
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.randn((1, 2, 512, 512, 256), generator=g, device="cuda")
    starts = [I.sliding_window_starts(n, 128) for n in (512, 512, 256)]
    assert len(starts[0]) * len(starts[1]) * len(starts[2]) == 147
    gp = I.GraphedPredictor(head)
    with torch.no_grad():
        y7 = I.sliding_window_inference(x, 128, 7, gp, overlap=0.5)
        y1 = I.sliding_window_inference(x, 128, 1, gp, overlap=0.5)
        assert y7.shape == (1, 8, 512, 512, 256) and bool(torch.isfinite(y7).all())
        scale = max(1.0, float(y7.abs().max()))
        assert float((y7 - y1).abs().max()) <= 1e-5 * scale
        direct = head(x[:, :, :128, :128, :128].contiguous())
        assert float((y7[:, :, :64, :64, :64] - direct[:, :, :64, :64, :64]).abs().max()) <= 1e-5 * scale
        del y1, direct
        y16 = I.sliding_window_inference(x, 128, 7, gp, overlap=0.5, autocast=True)
        err = float((y16 - y7).abs().max())
        assert 0 < err <= 2e-2, err


@pytest.mark.gpu
def test_fused_head_sliding_window_form():
    """micf_head_tail_fwd_fused_sw (bf16 mode: the patch-matrix-free head with the accumulate / count epilogue in its logits store)
    against micf_head_tail_col2im_sw on the same windows -- overlapping windows, an origin off every alignment, two volume samples:
    identical visit counts, sums within the two bf16 paths' distance; a voxel only ONE window covers holds bit for bit what
    micf_head_tail_fwd_fused writes for that window."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micformer_amd import ops
    n, Dc, Hc, Wc, Ci, Cm, Co, P = 3, 4, 8, 16, 96, 24, 8, 4
    g = torch.Generator().manual_seed(5)
    rn = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).cuda()
    x = rn(n * Dc * Hc * Wc, Ci)
    w_up, b_up, w_out, b_out = rn(Ci, Cm, P, P, P, k=0.1), rn(Cm), rn(Co, Cm, 3, 3, 3, k=0.1), rn(Co)
    dims = (n, Dc, Hc, Wc)
    VB, VD, VH, VW = 2, 24, 40, 72
    coords = torch.tensor([[0, 0, 0, 0], [0, 8, 4, 3], [1, 8, 8, 8]], dtype=torch.int32).cuda()
    ops.set_compute_dtype("bf16")
    try:
        assert ops.head_tail_fused_supported(dims, Ci, Co, P)
        wut = ops.head_tail_transposed_up(w_up)
        wb, bf = ops.head_tail_compose(w_up, b_up, w_out, wut)
        out_a, cnt_a = torch.zeros(VB, Co, VD, VH, VW, device="cuda"), torch.zeros(VB, VD, VH, VW, device="cuda")
        ops.head_tail_col2im_sw(ops.linear_fwd(x, wb, bf), b_out, out_a, cnt_a, coords, dims, P)
        pack = ops.head_tail_pack(wb, bf, b_out, P)[0]
        out_b, cnt_b = torch.zeros_like(out_a), torch.zeros_like(cnt_a)
        ops.head_tail_fwd_fused_sw(x, pack, out_b, cnt_b, coords, dims, Co, P)
        y = ops.head_tail_fwd_fused(x, pack, dims, Co, P)
    finally:
        ops.set_compute_dtype("fp32")
    assert torch.equal(cnt_a, cnt_b) and float(cnt_b.max()) == 2.0 and float(cnt_b.sum()) == n * 64 * Dc * Hc * Wc
    scale = float(out_a.abs().max())
    assert 0 < float((out_a - out_b).abs().max()) <= 1.5e-2 * scale
    assert torch.equal(out_b[1, :, 8:24, 8:40, 8:72], y[2])                  # window 2 is alone in its volume sample
    assert float(out_b[1].abs().sum()) == float(y[2].abs().sum())             # ... and nothing else was written there
    both = (cnt_b[0] == 2)
    assert bool(both.any())                                                  # (windows 0 and 1 do overlap)
    v0 = torch.zeros(Co, VD, VH, VW, device="cuda")
    v0[:, 0:16, 0:32, 0:64] += y[0]
    v0[:, 8:24, 4:36, 3:67] += y[1]
    assert float((out_b[0] - v0).abs().max()) <= 1e-6 * scale                  # (fp32 adds in either order)
