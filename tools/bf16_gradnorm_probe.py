#!/usr/bin/env python3
"""Dump the bf16-mode loss-gradient norm of every parameter of the base Head on the f7 fixture input (one 128^3 pair, eval mode)
to a JSON file (argv[1]) -- the data behind the per-tensor gates of tests/test_gpu_bf16.py (profiles/r03_bf16_gradnorms.txt)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import fill  # noqa: E402
from micformer_amd import MDiceLoss, ops  # noqa: E402
import micformer_amd.models.MICFormer_self as M  # noqa: E402

out = {}
sched = fill.stage_amplitude if "--f10" in sys.argv else None      # (--f10: the round-5 fixture with non-vanishing deep-stage gradients)
for mode in ("bf16", "fp32"):
    ops.set_compute_dtype(mode)
    h = M.Head(embed_dim=48, num_classes=8)
    fill.fill_state_dict(h, sched)
    h = h.cuda().eval()
    x = fill.make_volume(1, 128, 128, 128).cuda()
    tgt = fill.one_hot(fill.make_label_map(1, 128, 128, 128)).cuda()
    loss = MDiceLoss()(h(x), tgt)
    loss.backward()
    out[mode] = {"loss": float(loss.detach()), "gradnorms": {n: (float(p.grad.double().norm()) if p.grad is not None else "none")
                                                              for n, p in h.named_parameters()}}
json.dump(out, open(sys.argv[1], "w"))
print("wrote", sys.argv[1])
