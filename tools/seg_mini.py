"""Minimal multi-graph / shared-pool experiment with torch ops only."""
import torch
x = torch.randn(1 << 20, device="cuda")
pool = torch.cuda.graph_pool_handle()
cs = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(cs):
    g1 = torch.cuda.CUDAGraph(); g1.capture_begin(pool=pool, capture_error_mode="relaxed")
    a = x * 2
    tmp = a + 1
    s1 = tmp.sum()
    del tmp
    g1.capture_end()
    g2 = torch.cuda.CUDAGraph(); g2.capture_begin(pool=pool, capture_error_mode="relaxed")
    b = a * 3
    c = b + 5
    del a
    d = c * c
    e = torch.zeros(1 << 22, device="cuda")
    e[: 1 << 20] += d
    out = e.sum() + s1
    g2.capture_end()
torch.cuda.synchronize()
for i in range(4):
    x.normal_()
    g1.replay(); g2.replay()
    torch.cuda.synchronize()
    want = (((x * 2) * 3 + 5) ** 2).sum() + (x * 2 + 1).sum()
    print(i, float(out), float(want))
    junk = [torch.randn(1 << 22, device="cuda") for _ in range(8)]
    del junk
    torch.cuda.empty_cache()
print("ok")
