#!/usr/bin/env python3
"""Time the patch-matrix-free head tail kernels alone (base network at 128^3, batch 2: 65536 coarse voxels x 96 channels)."""
import sys, torch
sys.path.insert(0, ".")
from micformer_amd import ops
ops.set_compute_dtype("bf16")
B, Dc, Hc, Wc, Ci, Co, P = 2, 32, 32, 32, 96, 8, 4
g = torch.Generator().manual_seed(0)
x = torch.randn(B * Dc * Hc * Wc, Ci, generator=g).cuda()
wb = (torch.randn(216 * Co, Ci, generator=g) * 0.05).cuda()
bf = torch.randn(216 * Co, generator=g).cuda()
bo = torch.randn(Co, generator=g).cuda()
dy = torch.randn(B, Co, 4 * Dc, 4 * Hc, 4 * Wc, generator=g).cuda()
pf, pq = ops.head_tail_pack(wb, bf, bo, P)
dims = (B, Dc, Hc, Wc)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("fwd  us", round(timeit(lambda: ops.head_tail_fwd_fused(x, pf, dims, Co, P)), 1))
print("bwd  us", round(timeit(lambda: ops.head_tail_bwd_data_fused(dy, pq, dims, Ci, P)), 1))
