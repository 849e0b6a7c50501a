#!/usr/bin/env python3
"""Round-3 golden fixture from the REAL reference (CPU, this container only; see make_golden.py for the import shim):
  f7_base128_gradnorms_autocast.json   the reference's OWN reduced-precision behaviour on BASELINE config 2's network and size:
      base Head(48, 8), one 128^3 CT+MR pair, eval mode, forward + MDiceLoss + backward under torch.autocast("cpu", bfloat16)
      (the reference's one reduced-precision site is the autocast around its predictor, utils.py:236-238) -- the norm of every
      parameter's loss gradient, the loss, and max |logits - fp32 logits| on the stride-8 lattice of f7_base128.npz.
      tests/test_gpu_bf16.py anchors its per-tensor bf16 gradient-norm gates on these deviations from the fp32 norms.
usage:  python tests/golden/make_golden_r3.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as G  # noqa: E402
from oracle import fill  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    MS, dice = G.import_reference()
    crit = dice.MDiceLoss()
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    base = MS.Head(embed_dim=48, num_classes=8).eval()
    fill.fill_state_dict(base)
    x = fill.make_volume(1, size, size, size)
    tgt = fill.one_hot(fill.make_label_map(1, size, size, size))
    t0 = time.time()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        logits = base(x)
    loss = crit(logits.float(), tgt)
    loss.backward()
    print(f"autocast fwd+bwd at {size}^3: {time.time() - t0:.1f} s, logits dtype {logits.dtype}, loss {float(loss):.6f}")
    gn = {n: (float(p.grad.double().norm()) if p.grad is not None else "none") for n, p in base.named_parameters()}
    out = {"loss": float(loss), "gradnorms": gn}
    if size == 128:
        ref = np.load(os.path.join(HERE, "f7_base128.npz"))
        out["logits_max_abs_err_vs_fp32_stride8"] = float(np.abs(logits.float().detach().numpy()[:, :, ::8, ::8, ::8] - ref["logits_stride"]).max())
        out["fp32_loss"] = float(ref["loss"])
        name = "f7_base128_gradnorms_autocast.json"
    else:
        name = f"_probe_autocast_{size}.json"
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(out, f)
    print("wrote", name)


if __name__ == "__main__":
    main()
