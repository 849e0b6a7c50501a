#!/usr/bin/env python3
"""Microbenchmark of the fused block kernels against the per-op launch sequence at the base model's four stages
(batch 2, both modalities): HIP-event time per block pair, forward and backward.  python tools/bench_block.py [--dtype bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def timeit(fn, n=10, warm=2, reps=5):
    """us per call of fn, measured on a captured HIP graph of n calls (the Python / allocator overhead of a call is ~80 us,
    far more than the small-stage kernels themselves: eager timing would be host-bound)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--cross", action="store_true")
    ap.add_argument("--fused-only", action="store_true")
    ap.add_argument("--stage", type=int, default=-1, help="only this stage (0..3)")
    args = ap.parse_args()
    from micformer_amd import ops
    import test_gpu_block_fused as tb
    B = 2
    for si, (n, C, heads) in enumerate(((32, 48, 3), (16, 96, 6), (8, 192, 12), (4, 384, 24))):
        if args.stage >= 0 and si != args.stage:
            continue
        dims = (B, n, n, n)
        T = B * n ** 3
        if ops.block_tile_tokens(dims, C, heads, 4 * C) == 0:
            continue
        attn = "cross_attn" if args.cross else "self_attn"
        eps, scale = 1e-5, 16 ** -0.5
        gs = []
        for gi in range(2):
            P = tb.make_params(C, 4 * C, attn, 100 * gi)
            gs.append({"x": tb.rnd((T, C), gi), "kvsrc": tb.rnd((T, C), 5 + gi) if args.cross else None, "P": P, "attn": attn,
                       "s1": None, "s2": None})
        ref_f = lambda: [tb.ref_fwd(ops, g["x"], g["kvsrc"], g["P"], attn, None, None, dims, heads, eps, scale) for g in gs]
        t_ref_f = 0.0 if args.fused_only else timeit(ref_f)
        ops.set_compute_dtype(args.dtype)
        fus_f = lambda: ops.block_fwd(gs, dims, C, heads, eps, scale)
        t_fus_f = timeit(fus_f)
        outs = fus_f()
        ops.set_compute_dtype("fp32")
        refs = ref_f()
        bg = [{"dy": tb.rnd((T, C), 9 + i), "x": g["x"], "x1": o["x1"], "stats": o["stats"], "q": o["q"], "kv": o["kv"], "h": o["h"], "xn2": o["xn2"],
               "P": g["P"], "attn": attn, "s1": None, "s2": None, "cross": args.cross} for i, (g, o) in enumerate(zip(gs, outs))]
        ref_b = lambda: [tb.ref_bwd(ops, b["dy"], b["x"], r, b["P"], attn, None, None, dims, heads, scale, args.cross) for b, r in zip(bg, refs)]
        t_ref_b = 0.0 if args.fused_only else timeit(ref_b)
        ops.set_compute_dtype(args.dtype)
        t_fus_b = timeit(lambda: ops.block_bwd(bg, dims, C, heads, scale))
        ops.set_compute_dtype("fp32")
        fl = 2 * 2 * T * 12 * C * C
        print(f"stage {n}^3 C={C} T={2 * T} tile={ops.block_tile_tokens(dims, C, heads, 4 * C)}: fwd per-op {t_ref_f:8.1f} us  fused {t_fus_f:8.1f} us "
              f"({fl / t_fus_f / 1e6:6.1f} TF/s) | bwd per-op {t_ref_b:8.1f} us  fused {t_fus_b:8.1f} us ({2 * fl / t_fus_b / 1e6:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
