#!/usr/bin/env python3
"""Round-2 golden fixtures from the REAL reference (CPU, this container only; see make_golden.py for the import shim):
  f7_base128.npz   BASELINE config 2's network and size: base Head(48, 8) on one 128^3 CT+MR pair, eval mode -- strided logits,
                   argmax mask, top-2 margin, MDiceLoss, meandice, and the loss gradient's norm for every parameter
  f8_large160.npz  BASELINE config 4's network and size: large Head(96, 8) on one 160x160x128 pair, forward only -- strided
                   logits, argmax mask, margin
usage:  python tests/golden/make_golden_r2.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as G  # noqa: E402
from oracle import fill  # noqa: E402


def metric_fixture(dice):
    """f9_metric.npz: MDiceLoss_Val().metric (dice.py:223-230) on lattice logits.  The reference's empty-target branch builds its
    return value with device="cuda" (dice.py:139-141), which cannot run in this GPU-less container: those planes are NOT called
    through the reference; the fixture marks them (empty_target) and stores the value that branch returns by inspection."""
    z = fill.lattice((2, 8, 6, 5, 7), "F9.z", 3.0, 0.77)
    lab = fill.make_label_map(2, 6, 5, 7)
    lab[0][lab[0] == 5] = 0                 # class 5 absent in sample 0 (prediction non-empty -> 0)
    lab[1][lab[1] == 6] = 0                 # class 6 absent in sample 1 ...
    z[1, 6] = -z[1, 6].abs() - 0.1          # ... and never predicted -> 1
    t = fill.one_hot(lab)
    crit = dice.MDiceLoss_Val()
    out = np.zeros((2, 8), np.float32)
    empty = np.zeros((2, 8), np.uint8)
    for j in range(2):
        for i in range(8):
            if float(t[j, i].sum()) == 0:
                empty[j, i] = 1
                out[j, i] = 1.0 if int((torch.sigmoid(z[j, i]) > 0.5).sum()) == 0 else 0.0
            else:
                out[j, i] = float(crit.binary_dice(z[j, i], t[j, i], i, True))
    assert empty[0, 5] and empty[1, 6] and out[0, 5] == 0.0 and out[1, 6] == 1.0 and empty.sum() < 8
    G.save("f9_metric.npz", z=G.np32(z), label=lab.numpy().astype(np.uint8), metric=out, empty_target=empty)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    MS, dice = G.import_reference()
    crit = dice.MDiceLoss()
    if "--metric-only" in sys.argv:
        metric_fixture(dice)
        return
    metric_fixture(dice)

    print("F7 base 128^3 (forward + backward)")
    base = MS.Head(embed_dim=48, num_classes=8).eval()
    fill.fill_state_dict(base)
    x = fill.make_volume(1, 128, 128, 128)
    lab = fill.make_label_map(1, 128, 128, 128)
    tgt = fill.one_hot(lab)
    logits = base(x)
    loss = crit(logits, tgt)
    loss.backward()
    mask = torch.argmax(logits, 1)
    top2 = logits.topk(2, 1).values
    gn = {n: (float(p.grad.double().norm()) if p.grad is not None else "none") for n, p in base.named_parameters()}
    with open(os.path.join(HERE, "f7_base128_gradnorms.json"), "w") as f:
        json.dump(gn, f)
    G.save("f7_base128.npz", logits_stride=G.np32(logits[:, :, ::8, ::8, ::8]), mask=mask.numpy().astype(np.uint8),
           margin_stride=(top2[:, 0] - top2[:, 1])[:, ::2, ::2, ::2].detach().numpy().astype(np.float16),
           loss=G.np32(loss), meandice=np.float64(G.meandice_ref(mask, lab, 8).item()))
    del base, logits, loss

    print("F8 large 160x160x128 (forward)")
    large = MS.Head(embed_dim=96, num_classes=8).eval()
    fill.fill_state_dict(large)
    x = fill.make_volume(1, 160, 160, 128)
    with torch.no_grad():
        logits = large(x)
    mask = torch.argmax(logits, 1)
    top2 = logits.topk(2, 1).values
    G.save("f8_large160.npz", logits_stride=G.np32(logits[:, :, ::8, ::8, ::8]), mask=mask.numpy().astype(np.uint8),
           margin_stride=(top2[:, 0] - top2[:, 1])[:, ::2, ::2, ::2].numpy().astype(np.float16))
    print("done")


if __name__ == "__main__":
    main()
