#!/usr/bin/env python3
"""BASELINE config 4's shapes on one GPU: MicFormer large (embed 96) on a 160 x 160 x 128 CT+MR pair, fp32 or bf16-mode train steps.
Checks finiteness / batch independence of the forward and prints ms per step (eager and graph replay)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micformer_amd.models.MICFormer_self import Head
from micformer_amd.engine import TrainEngine
import bench
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
from micformer_amd import ops
ops.set_compute_dtype(sys.argv[2] if len(sys.argv) > 2 else "fp32")          # usage: run_large.py [batch] [fp32|bf16]
vol = (160, 160, 128)
model = Head(embed_dim=96, num_classes=8).cuda().train()
x, t = bench.synthetic_batch(B, vol, 8, torch.device("cuda"), 1234)
with torch.no_grad():
    y = model.eval()(x)
    assert torch.isfinite(y).all() and y.shape == (B, 8) + vol
    y0 = model(x[:1].contiguous())
    assert float((y[:1] - y0).abs().max()) < (1e-4 if ops.compute_dtype() == "fp32" else 2e-2)
model.train()
eng = TrainEngine(model, use_graph=True)
for _ in range(2):
    l = eng.step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    l = eng.step(x, t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
assert torch.isfinite(l)
print(json.dumps({"workload": f"MicFormer large (embed 96) train step, {vol} pair(s) x {B}, {ops.compute_dtype()}, graph replay", "ms_per_step": round(1e3 * dt, 2),
                  "pairs_per_s": round(B / dt, 2), "loss": float(l), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
