#!/usr/bin/env python3
"""BASELINE config 5: sliding-window inference of MicFormer base on a synthetic 512 x 512 x 256 two-modality volume
(roi 128^3, overlap 0.5 -> 147 windows), one MI355X.  Prints seconds per volume and windows/s for a few sw_batch_size values.
usage: python tools/bench_infer.py [D H W] [--autocast]     (--autocast: bf16 matrix-core mode for the predictor, utils.py:236-238)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micformer_amd.models.MICFormer_self import Head
from micformer_amd.inference import sliding_window_inference, sliding_window_starts

AUTOCAST = "--autocast" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
dims = tuple(int(a) for a in argv[:3]) if len(argv) >= 3 else (512, 512, 256)
torch.manual_seed(0)
model = Head(embed_dim=48, num_classes=8).cuda().eval()
x = torch.randn((1, 2) + dims, device="cuda")
nwin = 1
for L in dims:
    nwin *= len(sliding_window_starts(L, 128))
from micformer_amd.inference import GraphedPredictor
res = {}
for graphed in (False, True):
    for sw in (1, 3, 7):
        pred = GraphedPredictor(model) if graphed else model
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            # warm-up on the SAME volume and overlap (graph capture incl. the volume accumulators it bakes in), then the better of two runs
            sliding_window_inference(x, (128, 128, 128), sw, pred, overlap=0.5, autocast=AUTOCAST)
            dt = 1e9
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = sliding_window_inference(x, (128, 128, 128), sw, pred, overlap=0.5, autocast=AUTOCAST)
                torch.cuda.synchronize()
                dt = min(dt, time.perf_counter() - t0)
        res[f"{'graph' if graphed else 'eager'}_sw_batch_{sw}"] = {"s_per_volume": round(dt, 3), "windows_per_s": round(nwin / dt, 1)}
print(json.dumps({"workload": f"sliding-window inference, MicFormer base, volume {dims}, roi 128^3, overlap 0.5, {nwin} windows, {'bf16 mode' if AUTOCAST else 'fp32 kernels'}",
                  "results": res, "checksum": float(out.double().mean())}))
