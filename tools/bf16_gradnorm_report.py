#!/usr/bin/env python3
"""profiles/r03_bf16_gradnorms.txt: per-tensor loss-gradient norms of the base Head on the f7 fixture (one 128^3 pair) in this
build's bf16 mode (tools/bf16_gradnorm_probe.py, on MI355X) and in the reference's own CPU bf16 autocast
(tests/golden/f7_base128_gradnorms_autocast.json), both against the reference's fp32 norms.
usage: python tools/bf16_gradnorm_report.py gpurun_out/r03/bf16_gradnorms.json > profiles/r03_bf16_gradnorms.txt"""
import json
import math
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
o = json.load(open(sys.argv[1]))
FX = "f10_base128_sched" if "--f10" in sys.argv else "f7_base128"       # (--f10: the round-5 fixture, probe run with --f10 too)
FLOOR = 1e-4 if "--f10" in sys.argv else 1.5e-3
a = json.load(open(os.path.join(G, FX + "_gradnorms_autocast.json")))
r = json.load(open(os.path.join(G, FX + "_gradnorms.json")))
a.setdefault("fp32_loss", float("nan"))
W = json.load(open(os.path.join(G, FX + "_gradnorms_w16.json"))) if "--f10" in sys.argv else None
rows = [(r[n], abs(v - r[n]), abs(a["gradnorms"][n] - r[n]), abs(o["fp32"]["gradnorms"][n] - r[n]), n)
        for n, v in o["bf16"]["gradnorms"].items() if r[n] != "none"]
mx = max(x[0] for x in rows)
print(f"fixture {FX} (base Head, one 128^3 pair, eval): {len(rows)} parameter tensors with a gradient; largest fp32 norm {mx:.4e}")
print(f"loss: reference fp32 {a['fp32_loss']:.7f} | this build bf16 {o['bf16']['loss']:.7f}, fp32 {o['fp32']['loss']:.7f} | reference autocast {a['loss']:.7f}")
print(f"this build fp32 mode: worst per-tensor relative deviation {max(x[3] / x[0] for x in rows):.2e}")
print(f"median per-tensor relative deviation: this build bf16 {statistics.median(x[1] / x[0] for x in rows):.3f}, reference autocast "
      f"{statistics.median(x[2] / x[0] for x in rows):.3f}")
print("\ntensors by fp32 norm relative to the largest | n | worst relative deviation (build, autocast) | worst |dev| / largest norm (build, autocast)")
for lo, hi in ((1e-1, 10), (1e-2, 1e-1), (1e-3, 1e-2), (1e-4, 1e-3), (1e-5, 1e-4), (1e-6, 1e-5), (0, 1e-6)):
    sel = [x for x in rows if lo * mx <= x[0] < hi * mx]
    if sel:
        print(f"  [{lo:g}, {hi:g})  {len(sel):4d}  {max(x[1] / x[0] for x in sel):7.3f} {max(x[2] / x[0] for x in sel):7.3f}   "
              f"{max(x[1] for x in sel) / mx:.2e} {max(x[2] for x in sel) / mx:.2e}")
print(f"\ngate of tests/test_gpu_bf16.py: |norm - ref| <= 0.03 ref + {FLOOR:g} largest norm")
print(f"  worst use of the bound: build {max(x[1] / (0.03 * x[0] + FLOOR * mx) for x in rows):.2f}, "
      f"reference autocast {max(x[2] / (0.03 * x[0] + FLOOR * mx) for x in rows):.1f} "
      f"({sum(x[2] > 0.03 * x[0] + FLOOR * mx for x in rows)} tensors outside)")
print(f"  pure 3 % gate (no floor): build {sum(x[1] <= 0.03 * x[0] for x in rows)} of {len(rows)} tensors inside, "
      f"reference autocast {sum(x[2] <= 0.03 * x[0] for x in rows)}; 10 %: build {sum(x[1] <= 0.10 * x[0] for x in rows)}, "
      f"autocast {sum(x[2] <= 0.10 * x[0] for x in rows)}")
if W is not None:
    wd = [abs(W["gradnorms"][x[4]] - x[0]) / x[0] for x in rows]
    print(f"\nconditioning anchor (the reference in fp32 arithmetic, only its nn.Linear weights rounded to bfloat16): median per-tensor deviation "
          f"{statistics.median(wd):.3f}; inside 3 %: {sum(v <= 0.03 for v in wd)}, inside 10 %: {sum(v <= 0.10 for v in wd)} of {len(wd)}; "
          f"logits error {W['logits_max_abs_err_vs_fp32_stride8']:.3e} (reference autocast {a['logits_max_abs_err_vs_fp32_stride8']:.3e})")
grp = {}
for x in rows:
    k = ".".join(x[4].split(".")[:3])
    g = grp.setdefault(k, [0.0, 0.0, 0.0])
    g[0] += x[0] ** 2
    g[1] += o["bf16"]["gradnorms"][x[4]] ** 2
    g[2] += a["gradnorms"][x[4]] ** 2
print("\nnorm over a module group: fp32 reference | build bf16 (relative) | reference autocast (relative)")
for k, (p, q, s) in grp.items():
    print(f"  {k:38s} {math.sqrt(p):.3e}  {math.sqrt(q / p) - 1:+.4f}  {math.sqrt(s / p) - 1:+.4f}")
