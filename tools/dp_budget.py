#!/usr/bin/env python3
"""Per-step budget of the data-parallel gradient exchange: the slices dist.module_buckets cuts the flat gradient
into at the base model, their bytes, and a modelled ring all-reduce time at 8 ranks for the fp32 and the bf16 wire format.
Runs without a GPU (parameter shapes only).  Model: t = latency + bytes * 2 (N - 1) / N / bus_bw; bus_bw 300 GB/s is the figure
VERDICT r3 priced the exchange with (RCCL over 7 x 153 GB/s xGMI links per GPU), latency 30 us per collective.
  python tools/dp_budget.py [embed_dim]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micformer_amd.models.MICFormer_self import Head
from micformer_amd.dist import flatten_views, module_buckets

E = int(sys.argv[1]) if len(sys.argv) > 1 else 48
m = Head(embed_dim=E, num_classes=8)
names, params = zip(*[(n, p) for n, p in m.named_parameters()])
offs, total = flatten_views(params, align=4)
buckets = module_buckets(list(names), offs, [p.numel() for p in params], total)
N, BUS, LAT = 8, 300e9, 30e-6
print(f"# base Head(embed_dim={E}): {total} gradient elements = {total * 4 / 1e6:.1f} MB fp32; {len(buckets)} slices; ring all-reduce model: N = {N}, bus {BUS / 1e9:.0f} GB/s, {LAT * 1e6:.0f} us per collective")
print(f"{'slice (first parameter)':58s} {'MB fp32':>9s} {'t fp32 us':>10s} {'t bf16 us':>10s}")
t32 = t16 = 0.0
for a, b in buckets:
    first = min((o, n) for o, n in zip(offs, names) if o >= a)[1]
    by = (b - a) * 4
    x32 = LAT + by * 2 * (N - 1) / N / BUS
    x16 = LAT + by / 2 * 2 * (N - 1) / N / BUS
    t32 += x32; t16 += x16
    print(f"{first[:58]:58s} {by / 1e6:9.2f} {x32 * 1e6:10.0f} {x16 * 1e6:10.0f}")
print(f"{'total (serial on the RCCL stream)':58s} {total * 4 / 1e6:9.2f} {t32 * 1e6:10.0f} {t16 * 1e6:10.0f}")
