"""Times micf_block_fwd / micf_block_bwd at the base model's 32^3 stage (2 x 65536 tokens, C = 48, bf16 mode) with the wave-private
kernels (default) and the tile-per-workgroup kernels (test hook "block_wave" = 0): self pair, cross pair with the sampling fused in.
    python tools/bench_block_wave.py [--reps 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from micformer_amd import _lib, ops  # noqa: E402
from test_gpu_block_fused import make_params, rnd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--dims", type=int, nargs=4, default=[2, 32, 32, 32])
args = ap.parse_args()
ops.set_compute_dtype("bf16")
C, HEADS = 48, 3
dims = tuple(args.dims)
T = dims[0] * dims[1] * dims[2] * dims[3]
eps, scale = 1e-5, 0.25


def groups(kind):
    attn = "self_attn" if kind == "self" else "cross_attn"
    gs = []
    for i in range(2):
        P = make_params(C, 4 * C, attn, 20 + 40 * i)
        gd = {"x": rnd((T, C), 3 + i), "kvsrc": None, "P": P, "attn": attn, "s1": None, "s2": None}
        if kind == "sampled":
            P.update({"conv_offset.1.norm.weight": 1 + rnd((16,), 31 + i, 0.1), "conv_offset.1.norm.bias": rnd((16,), 32 + i, 0.1),
                      "conv_offset.3.weight": rnd((3, 16), 33 + i, 0.3)})
            gd.update(hid=rnd((T, 16), 13 + i), samp_src=rnd((T, C), 15 + i), want_xn=False)
        gs.append(gd)
    return gs


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / args.reps * 1e3


for kind in ("self", "sampled"):
    gs = groups(kind)
    for wave in ("0", "1"):
        _lib.set_option("block_wave", int(wave))
        o = ops.block_fwd([dict(g) for g in gs], dims, C, HEADS, eps, scale)
        t_f = timed(lambda: ops.block_fwd([dict(g) for g in gs], dims, C, HEADS, eps, scale))
        bg = [{"dy": rnd((T, C), 9 + i), "x": gs[i]["x"] if kind == "self" else None, "x1": o[i]["x1"], "stats": o[i]["stats"], "q": o[i]["q"],
               "kv": o[i]["kv"], "h": o[i]["h"], "xn2": o[i]["xn2"], "P": gs[i]["P"], "attn": gs[i]["attn"], "s1": None, "s2": None,
               "cross": kind != "self", "want_copy": kind != "self"} for i in range(2)]
        t_b = timed(lambda: ops.block_bwd([dict(g) for g in bg], dims, C, HEADS, scale))
        if kind == "self":                      # ... and with the producing LayerNorm's backward as the prologue
            for i, g in enumerate(bg):
                px = rnd((T, C), 60 + i)
                g["pre"] = {"d": rnd((T, C), 62 + i), "x": px, "mean": px.mean(1).contiguous(),
                            "rstd": (px.var(1, unbiased=False) + eps).rsqrt().contiguous(), "gamma": 1 + rnd((C,), 64 + i, 0.1)}
            timed(lambda: ops.block_bwd([dict(g) for g in bg], dims, C, HEADS, scale))
        print(f"{kind:8s} block_wave={wave}: fwd {t_f:7.1f} us   bwd {t_b:7.1f} us (eager call incl. host overhead)", flush=True)
