#!/usr/bin/env python3
"""Judge's round-4 item 5(c): MEASURE the persistent form of the 8^3 stage instead of pricing it.
N dependent passes of the fused forward block kernel at the base model's 8^3 stage (C = 192, 2 samples x 2 modalities = 128 workgroups
of 1024 threads, one per CU) as
   (a) N launches of micf_block_fwd replayed from ONE HIP graph (what the step does today), and
   (b) ONE launch of the same tile body walking the N passes with a device-wide barrier between them
       (micf_block_fwd_persistent_probe: atomic arrive + bounded spin; every workgroup resident).
Prints us per pass for both, self and cross (fused sampler) blocks.   python tools/bench_persist.py [--passes 12]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def replay_us(fn, per, reps=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
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
        best = min(best, e0.elapsed_time(e1) * 1e3 / per)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=12)
    args = ap.parse_args()
    from micformer_amd import ops
    import test_gpu_block_fused as tb
    ops.set_compute_dtype("bf16")
    B, n, C, heads = 2, 8, 192, 12
    dims, T = (B, n, n, n), B * n ** 3
    eps, scale = 1e-5, 16 ** -0.5
    sync = torch.zeros(2, dtype=torch.int32, device="cuda")
    for cross in (False, True):
        attn = "cross_attn" if cross else "self_attn"
        gs = []
        for gi in range(2):
            P = tb.make_params(C, 4 * C, attn, 100 * gi)
            gd = {"x": tb.rnd((T, C), gi), "kvsrc": None, "P": P, "attn": attn, "s1": None, "s2": None}
            if cross:                                              # the cross block samples its K/V source itself (as in the step)
                P.update({"conv_offset.1.norm.weight": 1 + 0.1 * tb.rnd((16,), 7), "conv_offset.1.norm.bias": 0.1 * tb.rnd((16,), 8),
                          "conv_offset.3.weight": 0.1 * tb.rnd((3, 16), 9)})
                gd.update(hid=tb.rnd((T, 16), 20 + gi), samp_src=tb.rnd((T, C), 30 + gi))
            gs.append(gd)
        N = args.passes
        a = replay_us(lambda: [ops.block_fwd(gs, dims, C, heads, eps, scale) for _ in range(N)], N)
        one = replay_us(lambda: ops.block_fwd(gs, dims, C, heads, eps, scale), 1)
        b = replay_us(lambda: ops.block_fwd(gs, dims, C, heads, eps, scale, persist_probe=(N, sync)), N)
        torch.cuda.synchronize()
        assert int(sync[1].item()) == 0, "a device-wide barrier timed out (workgroups not all resident?)"
        # the probe's passes compute what the launches compute
        y0 = ops.block_fwd(gs, dims, C, heads, eps, scale)[0]["y"].clone()
        y1 = ops.block_fwd(gs, dims, C, heads, eps, scale, persist_probe=(3, sync))[0]["y"]
        torch.cuda.synchronize()
        assert torch.equal(y0, y1) or float((y0 - y1).abs().max()) < 1e-5
        print(f"{'cross (fused sampler)' if cross else 'self':22s} 8^3 C=192, {N} dependent passes: graph of {N} launches {a:6.1f} us / pass "
              f"(a single launch alone: {one:5.1f} us) | one persistent launch with device-wide barriers {b:6.1f} us / pass", flush=True)


if __name__ == "__main__":
    main()
