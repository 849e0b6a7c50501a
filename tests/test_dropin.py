"""The reference's own import lines against this package (SURVEY.md 8(b) "Import").

`MicFormer/test.ipynb:11` and the intent of `train_mmwhs_noPad.py:26` are `from models.MICFormer_self import Head`;
`train_mmwhs_noPad.py:19-20` are `from loss import MDiceLoss` / `from loss.dice import MDiceLoss_Val`.  With ONE sys.path entry
(`micformer_amd/dropin`) those lines must resolve to the HIP modules -- from any working directory, in a fresh interpreter, without
the repository root on the path.  The GPU half runs the reference's literal loop body (train_mmwhs_noPad.py:108-114, 148, 183-207:
stock torch.optim.Adam + CosineAnnealingLR) on the modules imported that way, against the reference's own two-iteration fixture f5.
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "micformer_amd", "dropin")

_SCRIPT = r"""
import sys
assert not any(p.rstrip('/') == {root!r} for p in sys.path), sys.path      # the repository root is NOT on the path
sys.path.insert(0, {dropin!r})
from models.MICFormer_self import Head                     # MicFormer/test.ipynb:11
from models.MICFormer_self import MicFormer, BasicLayer, BasicLayerUp, CrossTransformerBlock3D, TransformerBlock3D
from models.MICFormer_self import CrossWindowAttention3D, WindowAttention3D, PatchEmbed3D, PatchMerging, PatchExpand, Mlp
from models.MICFormer_self import LayerNormProxy, window_partition, window_reverse, get_window_size
from models.STN import SpatialTransformer, Re_SpatialTransformer      # MicFormer/models/MICFormer_self.py:10
from loss import MDiceLoss                                 # train_mmwhs_noPad.py:19
from loss.dice import MDiceLoss_Val                        # train_mmwhs_noPad.py:20
import micformer_amd.models.MICFormer_self as impl
import micformer_amd.loss.dice as limpl
assert Head is impl.Head and MicFormer is impl.MicFormer and MDiceLoss is limpl.MDiceLoss and MDiceLoss_Val is limpl.MDiceLoss_Val
model_1 = Head(embed_dim=48, num_classes=8)                # train_mmwhs_noPad.py:92
assert sum(p.numel() for p in model_1.parameters()) == 61722608
assert len(model_1.state_dict()) == 1626
criterion = MDiceLoss()
print("dropin ok")
"""


def test_reference_import_lines_with_one_sys_path_entry(tmp_path):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT, dropin=DROPIN)], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "dropin ok" in r.stdout


@pytest.mark.gpu
def test_reference_loop_body_on_the_hip_modules_against_f5():
    """optimizer.zero_grad(); segs = model(x); loss = criterion(segs, y); loss.backward(); optimizer.step(); scheduler.step()
    (train_mmwhs_noPad.py:183-207) with torch.optim.Adam(lr=1e-4) (:114) and CosineAnnealingLR (:148), two iterations, on the
    modules imported through the drop-in path: parameters after each iteration, the second loss and the learning rates against
    the reference's own run (tests/golden/f5_adam.npz; tolerances of test_two_train_steps_against_reference)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import fill
    if DROPIN not in sys.path:
        sys.path.insert(0, DROPIN)
    from models.MICFormer_self import Head
    from loss import MDiceLoss
    from micformer_amd import ops
    assert ops.compute_dtype() == "fp32"
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "f5_adam.npz")).items()}
    model_1 = Head(embed_dim=48, num_classes=8)
    with torch.no_grad():
        for name, t in model_1.state_dict().items():
            t.copy_(fill.fill_tensor(name, t))
    model_1 = model_1.cuda().eval()                       # (the fixture ran the reference in eval(): DropPath off)
    criterion = MDiceLoss().cuda()
    optimizer = torch.optim.Adam(model_1.parameters(), lr=1e-4, weight_decay=0)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, 150)
    inputs_S1 = fill.make_volume(1, 64, 64, 64).cuda()
    labels_S1 = fill.one_hot(fill.make_label_map(1, 64, 64, 64)).cuda()
    names = [k[3:] for k in g if k.startswith("w1.")]
    sub = lambda v: v.reshape(-1)[::17] if v.numel() > 20000 else v

    def close(got, want, atol, what):
        err = float((got.detach().cpu().double() - want.double()).abs().max())
        assert math.isfinite(err) and err <= atol, f"{what}: {err:.3e} > {atol:.1e}"

    losses = []
    for it in (1, 2):
        optimizer.zero_grad()
        segs_S1 = model_1(inputs_S1)
        loss_ = criterion(segs_S1, labels_S1)
        losses.append(loss_.item())
        loss_.backward()
        optimizer.step()
        scheduler.step()
        sd = model_1.state_dict()
        for n in names:
            close(sub(sd[n]), g[f"w{it}." + n], 3e-7 if it == 1 else 2e-6, f"w{it}." + n)
        assert abs(optimizer.param_groups[0]["lr"] - float(g[f"lr_after_{it}"])) < 1e-12
    close(torch.tensor(losses[1]), g["loss2"], 2e-5, "loss2")
    # the never-used concat_back_dim.0 has no gradient at all in this loop (MS.py:1015-1016), exactly as in the reference
    assert model_1.swin.concat_back_dim[0].weight.grad is None


def test_block_parameters_resolved_by_state_dict_name_without_walking_the_module():
    """The per-call parameter lookup of a block (host code of the engine-less path) returns exactly named_parameters()' tensors for
    every key the launches use, and sees a parameter replaced after construction."""
    from micformer_amd import functional as Fn
    import micformer_amd.models.MICFormer_self as M
    for blk, keys in ((M.TransformerBlock3D(48, 3, window_size=(2, 2, 2)), Fn.SELF_KEYS),
                      (M.CrossTransformerBlock3D(48, 3, window_size=(2, 2, 2)), Fn.CROSS_KEYS)):
        want = dict(blk.named_parameters())
        got = M._block_params(blk, keys)
        assert len(got) == len(keys) and all(g is want[k] for g, k in zip(got, keys))
        blk.mlp.fc1.weight = torch.nn.Parameter(torch.zeros_like(blk.mlp.fc1.weight))
        assert M._block_params(blk, ("mlp.fc1.weight",))[0] is blk.mlp.fc1.weight
