#!/usr/bin/env python3
"""Stage timeline of one replayed training step from a rocprofv3 kernel-trace CSV.  The fused block kernels carry the channel
count in their template arguments, so the step is cut wherever that count changes (stage 3 / the heads run no fused kernel: the
gaps between the C=192 encoder run and the C=192 decoder run, and around the loss).  Per segment: wall time, time covered by
the fused block kernels, by any other kernel, and idle.   usage: trace_stages.py trace.csv [marker=drop_path_draw_kernel] [--detail]
(--detail: per segment, the kernels by total time: count x mean us)"""
import csv, re, sys
detail = "--detail" in sys.argv
argv = [a for a in sys.argv if a != "--detail"]
rows = list(csv.DictReader(open(argv[1])))
marker = argv[2] if len(argv) > 2 else "drop_path_draw_kernel"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
marks = [i for i, e in enumerate(ev) if marker in e[2]]
if len(marks) < 3:
    sys.exit("need >= 3 steps")
step = ev[marks[-2]:marks[-1]]          # one step = from one DropPath draw (first launch of the forward) to the next
t0 = ev[marks[-2]][0]
def _blk(n):          # (direction, C) of a fused block kernel: block_fwd_kernel<C, ...> / the wave-private block_fwd_wave48_kernel<...>
    m = re.search(r"block_(fwd|bwd)_kernel<(\d+)", n)
    if m:
        return m.group(1), int(m.group(2))
    m = re.search(r"block_(fwd|bwd)_wave(\d+)_kernel", n)
    return (m.group(1), int(m.group(2))) if m else None


blk = [(s, e, _blk(n)) for s, e, n in step]
blk = [(s, e, m[0], m[1]) for s, e, m in blk if m]
# stages: maximal runs of fused launches with the same (direction, C), per-op kernels between them included
runs = []
for s, e, d, c in blk:
    if runs and tuple(runs[-1][2:]) == (d, c):
        runs[-1][1] = e
    else:
        runs.append([s, e, d, c])
segs, prev = [], t0
for s, e, d, c in runs:
    if s > prev:
        segs.append(("  (between stages)", prev, s))
    segs.append((f"{d} C={c}", s, e))
    prev = e
segs.append(("  (between stages)", prev, step[-1][1]))


def covered(lo, hi, sel):
    iv = sorted((max(s, lo), min(e, hi)) for s, e, n in step if sel(n) and e > lo and s < hi)
    tot, cur = 0, lo
    for s, e in iv:
        if e > cur:
            tot += e - max(s, cur)
            cur = e
    return tot


print(f"step wall {(step[-1][1] - t0) / 1e6:.3f} ms, {len(step)} kernels")
print(f"{'segment':22s} {'wall ms':>8s} {'fused':>8s} {'other':>8s} {'idle':>8s} {'kernels':>8s}")
for name, lo, hi in segs:
    if hi - lo < 20000:
        continue
    fz = covered(lo, hi, lambda n: _blk(n) is not None)
    al = covered(lo, hi, lambda n: True)
    nk = sum(1 for s, e, n in step if lo <= s < hi)
    print(f"{name:22s} {(hi - lo) / 1e6:8.3f} {fz / 1e6:8.3f} {(al - fz) / 1e6:8.3f} {(hi - lo - al) / 1e6:8.3f} {nk:8d}")
    if detail:
        agg = {}
        for s, e, n in step:
            if lo <= s < hi:
                k = re.sub(r"\(.*", "", n)[:70]
                a = agg.setdefault(k, [0, 0])
                a[0] += 1
                a[1] += e - s
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"      {t / 1e3:8.1f} us  {c:3d} x {t / c / 1e3:6.1f}  {k}")
