#!/usr/bin/env python3
"""Time individual C-ABI entry points at the shapes of the base 128^3 / batch-2 training step (diagnostic tool).

  python tools/microbench.py linear_fwd 65536 192 48 [--reps 50]
  python tools/microbench.py all
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from micformer_amd import ops  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def r(*s):
    return torch.randn(*s, device="cuda")


def case(name, args, reps):
    if name == "linear_fwd":
        M, N, K = args
        a, w, b = r(M, K), r(N, K), r(N)
        us = timeit(lambda: ops.linear_fwd(a, w, b), reps)
        byt, fl = 4 * (M * K + M * N + N * K), 2 * M * N * K
    elif name == "linear_fwd_gelu":
        M, N, K = args
        a, w, b = r(M, K), r(N, K), r(N)
        us = timeit(lambda: ops.linear_fwd(a, w, b, act=1, want_pre=True), reps)
        byt, fl = 4 * (M * K + 2 * M * N + N * K), 2 * M * N * K
    elif name == "linear_fwd_resid":
        M, N, K = args
        a, w, b, res = r(M, K), r(N, K), r(N), r(M, N)
        us = timeit(lambda: ops.linear_fwd(a, w, b, resid=res), reps)
        byt, fl = 4 * (M * K + 2 * M * N + N * K), 2 * M * N * K
    elif name == "linear_bwd_data":
        M, N, K = args
        dy, w = r(M, N), r(N, K)
        us = timeit(lambda: ops.linear_bwd_data(dy, w), reps)
        byt, fl = 4 * (M * K + M * N + N * K), 2 * M * N * K
    elif name == "linear_bwd_weight":
        M, N, K = args
        dy, a = r(M, N), r(M, K)
        dw, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
        us = timeit(lambda: ops.linear_bwd_weight(dy, a, dw, db), reps)
        byt, fl = 4 * (M * K + M * N + N * K), 2 * M * N * K
    elif name == "layernorm_fwd":
        M, C = args
        x, g, b = r(M, C), r(C), r(C)
        us = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5), reps)
        byt, fl = 8 * M * C, 8 * M * C
    elif name == "layernorm_bwd":
        M, C = args
        x, g, b, dy = r(M, C), r(C), r(C), r(M, C)
        y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        us = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, add=dy), reps)
        byt, fl = 12 * M * C, 12 * M * C
    elif name in ("conv3_fwd", "conv3_bwd_data", "conv3_bwd_weight"):
        B, D, C, N = args          # cubic grid D^3, inputs [xn|xa] of C channels each (C2 = 0 when N == 8: out_conv)
        dims = (B, D, D, D)
        T = B * D ** 3
        two = N == 16
        x1 = r(T, C)
        x2 = r(T, C) if two else None
        Cin = 2 * C if two else C
        w, bias = r(N, Cin, 3, 3, 3), r(N)
        nc = not two
        dy = r(B, N, D, D, D) if nc else r(T, N)
        if name == "conv3_fwd":
            us = timeit(lambda: ops.conv3_fwd(x1, w, bias, dims, x2=x2, ncdhw_out=nc), reps)
        elif name == "conv3_bwd_data":
            us = timeit(lambda: ops.conv3_bwd_data(dy, w, dims, C, C if two else 0, ncdhw=nc), reps)
        else:
            dw, db = torch.zeros_like(w), torch.zeros(N, device="cuda")
            us = timeit(lambda: ops.conv3_bwd_weight(dy, x1, dw, db, dims, x2=x2, ncdhw=nc), reps)
        byt, fl = 4 * (T * Cin + T * N), 2 * T * 27 * Cin * N
    elif name in ("attn_fwd", "attn_bwd"):
        B, D, C, heads = args
        dims = (B, D, D, D)
        T = B * D ** 3
        q, kv, do = r(T, C), r(T, 2 * C), r(T, C)
        ws = (2, 2, 2) if D > 1 else (1, 1, 1)
        if name == "attn_fwd":
            us = timeit(lambda: ops.window_attn_fwd(q, kv, dims, heads, ws, 0.25), reps)
            byt = 16 * T * C
        else:
            us = timeit(lambda: ops.window_attn_bwd(q, kv, do, dims, heads, ws, 0.25), reps)
            byt = 28 * T * C
        fl = 32 * T * C
    elif name in ("osample_fwd", "osample_bwd"):
        B, D, C = args
        dims = (B, D, D, D)
        T = B * D ** 3
        h, lg, lb, w1, xa, dxs = r(T, 16), r(16), r(16), r(3, 16) * 0.3, r(T, C), r(T, C)
        fl_, xs = ops.offset_sample_fwd(h, lg, lb, w1, xa, dims, 1e-5)
        if name == "osample_fwd":
            us = timeit(lambda: ops.offset_sample_fwd(h, lg, lb, w1, xa, dims, 1e-5), reps)
            byt = 4 * T * (2 * C + 19)
        else:
            dxa = torch.zeros_like(xa)
            dlg, dlb, dw1 = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda"), torch.zeros(3, 16, device="cuda")
            us = timeit(lambda: ops.offset_sample_bwd(dxs, h, lg, lb, w1, xa, fl_, dxa, dlg, dlb, dw1, dims, 1e-5), reps)
            byt = 4 * T * (4 * C + 35)
        fl = 40 * T * C
    elif name in ("conv_up_fwd", "conv_up_bwd_data", "conv_up_bwd_weight"):
        B, D, C, N, k = args
        x, w, bias = r(B, D, D, D, C), r(C, N, k, k, k), r(N)
        dy = r(B, D * k, D * k, D * k, N)
        if name == "conv_up_fwd":
            us = timeit(lambda: ops.conv_up_fwd(x, w, bias, k), reps)
        elif name == "conv_up_bwd_data":
            us = timeit(lambda: ops.conv_up_bwd_data(dy, w, tuple(x.shape), k), reps)
        else:
            dw, db = torch.zeros_like(w), torch.zeros(N, device="cuda")
            us = timeit(lambda: ops.conv_up_bwd_weight(dy, x, dw, db, k), reps)
        byt, fl = 4 * (x.numel() + dy.numel()), 2 * dy.numel() * C
    else:
        raise SystemExit(f"unknown case {name}")
    print(f"{name:20s} {str(args):28s} {us:9.1f} us  {byt / us / 1e3:8.1f} GB/s  {fl / us / 1e6:7.2f} TFLOP/s", flush=True)


ALL = [
    ("linear_fwd", (131072, 48, 48)), ("linear_fwd", (131072, 96, 48)), ("linear_fwd_gelu", (131072, 192, 48)), ("linear_fwd_resid", (131072, 48, 192)),
    ("linear_fwd", (16384, 96, 96)), ("linear_fwd_gelu", (16384, 384, 96)), ("linear_fwd", (2048, 192, 192)), ("linear_fwd_gelu", (2048, 768, 192)),
    ("linear_fwd_resid", (2048, 192, 768)), ("linear_fwd", (256, 384, 384)), ("linear_fwd_gelu", (256, 1536, 384)), ("linear_fwd_resid", (256, 384, 1536)),
    ("linear_bwd_data", (131072, 192, 48)), ("linear_bwd_data", (131072, 48, 192)), ("linear_bwd_data", (2048, 768, 192)), ("linear_bwd_data", (256, 1536, 384)),
    ("linear_bwd_weight", (131072, 192, 48)), ("linear_bwd_weight", (131072, 48, 48)), ("linear_bwd_weight", (2048, 192, 192)), ("linear_bwd_weight", (256, 1536, 384)),
    ("layernorm_fwd", (131072, 48)), ("layernorm_bwd", (131072, 48)), ("layernorm_fwd", (2048, 192)), ("layernorm_bwd", (2048, 192)), ("layernorm_bwd", (256, 384)),
    ("conv3_fwd", (2, 32, 48, 16)), ("conv3_bwd_data", (2, 32, 48, 16)), ("conv3_bwd_weight", (2, 32, 48, 16)),
    ("conv3_fwd", (2, 8, 192, 16)), ("conv3_bwd_data", (2, 8, 192, 16)), ("conv3_bwd_weight", (2, 8, 192, 16)),
    ("conv3_fwd", (2, 128, 24, 8)), ("conv3_bwd_data", (2, 128, 24, 8)), ("conv3_bwd_weight", (2, 128, 24, 8)),
    ("attn_fwd", (2, 32, 48, 3)), ("attn_bwd", (2, 32, 48, 3)), ("attn_bwd", (2, 8, 192, 12)),
    ("osample_fwd", (2, 32, 48)), ("osample_bwd", (2, 32, 48)), ("osample_bwd", (2, 8, 192)),
    ("conv_up_fwd", (2, 32, 96, 24, 4)), ("conv_up_bwd_data", (2, 32, 96, 24, 4)), ("conv_up_bwd_weight", (2, 32, 96, 24, 4)),
]

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("args", nargs="*", type=int)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=("fp32", "bf16"))
    a = ap.parse_args()
    ops.set_compute_dtype(a.dtype)
    if a.case == "all":
        for n, g in ALL:
            case(n, g, a.reps)
    else:
        case(a.case, tuple(a.args), a.reps)
