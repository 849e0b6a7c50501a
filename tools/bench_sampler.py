#!/usr/bin/env python3
"""Time the deformable sampler's adjoint (micf_offset_head_bwd: sampler backward + d(xa) sum + conv data gradient, both modalities of
a cross pair per call) at the four stage shapes of the base 128^3 / batch-2 step, graph-replayed.  Diagnostic tool.
   python tools/bench_sampler.py [--reps 50] [--offset-scale 0.3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from micformer_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--offset-scale", type=float, default=0.3)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    ops.set_compute_dtype(args.dtype)
    g = torch.Generator(device="cuda").manual_seed(1)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    for dims, C in (((2, 32, 32, 32), 48), ((2, 16, 16, 16), 96), ((2, 8, 8, 8), 192), ((2, 4, 4, 4), 384)):
        B, D, H, W = dims
        T = B * D * H * W
        groups = []
        for m in range(2):
            P = {"conv_offset.0.weight": r(16, 2 * C, 3, 3, 3) * 0.05, "conv_offset.1.norm.weight": 1 + 0.1 * r(16),
                 "conv_offset.1.norm.bias": 0.1 * r(16), "conv_offset.3.weight": r(3, 16) * args.offset_scale}
            G = {k: torch.zeros_like(v) for k, v in P.items()}
            hid, xa = r(T, 16), r(T, C)
            flow, _ = ops.offset_sample_fwd(hid, P["conv_offset.1.norm.weight"], P["conv_offset.1.norm.bias"], P["conv_offset.3.weight"], xa, dims, 1e-5)
            groups.append({"dxs": r(T, C), "hid": hid, "flow": flow, "xa": xa, "P": P, "G": G, "dxa": torch.zeros(T, C, device="cuda"),
                           "dxn": torch.zeros(T, C, device="cuda")})
        fn = lambda: ops.offset_head_bwd(groups, dims, 1e-5)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(args.reps):
                    fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr.replay()
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"offset_head_bwd {dims} C={C}: {e0.elapsed_time(e1) / args.reps * 1e3:8.1f} us per pair", flush=True)


if __name__ == "__main__":
    main()
