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
