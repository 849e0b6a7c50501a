#!/usr/bin/env python3
"""Round-5 golden fixture from the REAL reference (CPU, this container only; see make_golden.py for the import shim):
  f10_base128_sched.npz / f10_base128_sched_gradnorms.json / f10_base128_sched_gradnorms_autocast.json
      BASELINE config 2's network and size -- base Head(48, 8), one 128^3 CT+MR pair, eval mode -- filled with the closed-form
      weights TIMES oracle.fill.stage_amplitude (the gains of the three PatchExpand LayerNorms x 30): unlike f7, whose deep-stage
      gradient norms are 1e-4 ... 1e-6 of the head's, every stage group of this fixture carries 1e-2 ... 2e-1 of the largest
      gradient norm, so the per-tensor bf16 gradient gates of tests/test_gpu_bf16.py see the 8^3 / 4^3 stages (93 % of the
      parameters).  Stored: strided logits, argmax mask, top-2 margin, MDiceLoss, the norm of every parameter's loss gradient in
      fp32, the same norms (+ loss, logits error) from the reference's own torch.autocast("cpu", bfloat16) run, and from the
      reference in fp32 arithmetic with only its nn.Linear weights rounded to bfloat16 (..._w16.json: the conditioning anchor).
usage:  python tests/golden/make_golden_r5.py [size]
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
    x = fill.make_volume(1, size, size, size)
    lab = fill.make_label_map(1, size, size, size)
    tgt = fill.one_hot(lab)

    def model():
        m = MS.Head(embed_dim=48, num_classes=8).eval()
        fill.fill_state_dict(m, fill.stage_amplitude)
        return m

    t0 = time.time()
    base = model()
    logits = base(x)
    loss = crit(logits, tgt)
    loss.backward()
    print(f"fp32 fwd+bwd at {size}^3: {time.time() - t0:.1f} s, loss {float(loss.detach()):.6f}, |logits| max {float(logits.abs().max()):.3f}")
    mask = torch.argmax(logits, 1)
    top2 = logits.topk(2, 1).values
    gn = {n: (float(p.grad.double().norm()) if p.grad is not None else "none") for n, p in base.named_parameters()}
    tag = "f10_base128_sched" if size == 128 else f"_probe_f10_{size}"
    with open(os.path.join(HERE, tag + "_gradnorms.json"), "w") as f:
        json.dump(gn, f)
    G.save(tag + ".npz", logits_stride=G.np32(logits[:, :, ::8, ::8, ::8]), mask=mask.numpy().astype(np.uint8),
           margin_stride=(top2[:, 0] - top2[:, 1])[:, ::2, ::2, ::2].detach().numpy().astype(np.float16),
           loss=G.np32(loss), meandice=np.float64(G.meandice_ref(mask, lab, 8).item()))
    ref_logits = logits.detach()
    del base, logits, loss

    t0 = time.time()
    base = model()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        logits = base(x)
    loss = crit(logits.float(), tgt)
    loss.backward()
    print(f"autocast fwd+bwd: {time.time() - t0:.1f} s, loss {float(loss.detach()):.6f}")
    gna = {n: (float(p.grad.double().norm()) if p.grad is not None else "none") for n, p in base.named_parameters()}
    out = {"loss": float(loss.detach()), "gradnorms": gna,
           "logits_max_abs_err_vs_fp32_stride8": float((logits.float().detach() - ref_logits)[:, :, ::8, ::8, ::8].abs().max())}
    with open(os.path.join(HERE, tag + "_gradnorms_autocast.json"), "w") as f:
        json.dump(out, f)
    del base, logits, loss

    # the conditioning anchor: the reference in FP32 ARITHMETIC with nothing changed but its nn.Linear weights rounded to bfloat16
    # (the least any bf16 matrix-core mode does to them).  How far that alone moves the gradient norms is what the closed-form
    # network's conditioning costs at bf16 precision, before any activation or accumulation is touched.
    t0 = time.time()
    base = model()
    with torch.no_grad():
        for p in base.parameters():
            if p.dim() == 2:
                p.copy_(p.bfloat16().float())
    logits = base(x)
    loss = crit(logits, tgt)
    loss.backward()
    print(f"fp32 arithmetic, bf16-rounded linear weights: {time.time() - t0:.1f} s, loss {float(loss.detach()):.6f}")
    gnw = {n: (float(p.grad.double().norm()) if p.grad is not None else "none") for n, p in base.named_parameters()}
    with open(os.path.join(HERE, tag + "_gradnorms_w16.json"), "w") as f:
        json.dump({"loss": float(loss.detach()), "gradnorms": gnw,
                   "logits_max_abs_err_vs_fp32_stride8": float((logits.detach() - ref_logits)[:, :, ::8, ::8, ::8].abs().max())}, f)
    print("wrote", tag)


if __name__ == "__main__":
    main()
