#!/usr/bin/env python3
"""Kernel sequence of one depth slot of the C=192 stage (forward and backward) from a rocprofv3 kernel-trace CSV of a graph-
replayed step: start offset, duration, gap to the previous kernel's end on the critical stream, queue.  A slot runs from one
self-pair fused block launch to the next (the self pair is every second block launch of the stage).
usage: trace_slot.py trace.csv [C=192] [slot|stage] [marker=drop_path_draw_kernel]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
C = int(sys.argv[2]) if len(sys.argv) > 2 else 192
whole = len(sys.argv) > 3 and sys.argv[3] == "stage"
marker = sys.argv[4] if len(sys.argv) > 4 else "drop_path_draw_kernel"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
marks = [i for i, e in enumerate(ev) if marker in e[2]]
step = ev[marks[-2]:marks[-1]]


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("micf::", "")
    return n[:78]


for d in ("fwd", "bwd"):
    idx = [i for i, e in enumerate(step) if f"block_{d}_kernel<{C}," in e[2]]
    if len(idx) < 6:
        continue
    lo, hi = idx[2], idx[4]                 # third block launch of the stage -> fifth: one full slot (two block launches)
    if whole:                               # every launch of the (first) run of this stage
        run = [i for i in idx if i - idx[0] < 400 and all("block_" not in step[j][2] or f"<{C}," in step[j][2] for j in range(idx[0], i))]
        lo, hi = run[0], run[-1] + 1
    print(f"== {d} C={C}: {'stage' if whole else 'one slot'} = {(step[hi][0] - step[lo][0]) / 1e3:.1f} us, {hi - lo} kernels")
    t0, prev_end = step[lo][0], step[lo][0]
    for s, e, n, q in step[lo:hi]:
        print(f"  +{(s - t0) / 1e3:7.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{q:>3s}  {short(n)}")
        prev_end = max(prev_end, e)
