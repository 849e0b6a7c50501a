#!/usr/bin/env python3
"""BASELINE config 4's fp8 leg as numbers: the large network (embed 96, head_dim 32) on a 160 x 160 x 128 pair in the bf16 mode and in
bf16 + fp8 attention (MICF_DTYPE_BF16_ATTN_FP8), forward error against the reference's fp32 logits (tests/golden/f8_large160.npz),
margin-enforced argmax agreement, and ms per train step (graph replay, batch 1) of both modes on this box.
  python tools/fp8_report.py > profiles/r04_fp8_report.txt"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import fill
from micformer_amd import ops
from micformer_amd.engine import TrainEngine
import micformer_amd.models.MICFormer_self as M
import bench

g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "f8_large160.npz")).items()}
h = M.Head(embed_dim=96, num_classes=8, depths=(2, 2, 6, 2))
with torch.no_grad():
    for name, t in h.state_dict().items():
        t.copy_(fill.fill_tensor(name, t))
h = h.cuda().eval()
x = fill.make_volume(1, 160, 160, 128).cuda()
out = {}
logits = {}
for mode in ("fp32", "bf16", "bf16+fp8attn"):
    ops.set_compute_dtype(mode)
    with torch.no_grad():
        l = h(x)
    logits[mode] = l
    s = l[:, :, ::8, ::8, ::8].cpu()
    bad = (l.argmax(1).cpu() != g["mask"].long())[:, ::2, ::2, ::2]
    m = g["margin_stride"].float()
    out[mode] = {"max_abs_logit_err_vs_reference": float((s - g["logits_stride"]).abs().max()),
                 "rms_logit_err_vs_reference": float((s - g["logits_stride"]).pow(2).mean().sqrt()),
                 "argmax_mismatch_rate": float(bad.float().mean()),
                 "argmax_mismatch_where_margin_gt_4e-2": int((bad & (m > 4e-2)).sum()),
                 "logit_range": [float(g["logits_stride"].min()), float(g["logits_stride"].max())]}
out["fp8attn_vs_bf16_max_abs"] = float((logits["bf16+fp8attn"] - logits["bf16"]).abs().max())
del logits
vol = (160, 160, 128)
xt, tt = bench.synthetic_batch(1, vol, 8, torch.device("cuda"), 1234)
for mode in ("bf16", "bf16+fp8attn", "bf16", "bf16+fp8attn"):
    ops.set_compute_dtype(mode)
    torch.manual_seed(0)
    model = M.Head(embed_dim=96, num_classes=8).cuda().train()
    eng = TrainEngine(model, use_graph=True)
    for _ in range(3):
        l = eng.step(xt, tt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        l = eng.step(xt, tt)
    torch.cuda.synchronize()
    out.setdefault("train_step_ms_160x160x128_B1", {}).setdefault(mode, []).append(round(1e3 * (time.perf_counter() - t0) / 10, 2))
    out.setdefault("train_loss_after_13_steps", {})[mode] = float(l)
    del eng, model
print(json.dumps(out, indent=1))
