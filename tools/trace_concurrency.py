#!/usr/bin/env python3
"""Concurrency of one replayed step from a rocprofv3 kernel-trace CSV: wall time, time with 0 / 1 / >=2 kernels in flight,
and the kernels that run alone the longest.  usage: trace_concurrency.py trace.csv [marker_kernel=drop_path_draw_kernel]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "drop_path_draw_kernel"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
marks = [i for i, e in enumerate(ev) if marker in e[2]]
if len(marks) < 3:
    sys.exit("need >= 3 steps")
a, b = marks[-2], marks[-1]          # one step = from one DropPath draw (first launch of the forward) to the next
step = ev[a:b]
t0, t1 = ev[a][0], max(e for _, e, _ in step)
pts = []
for s, e, n in step:
    pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
busy = collections.Counter(); alone = collections.Counter(); cur = 0; last = t0; active = {}
for t, d, n in pts:
    dt = t - last
    busy[min(cur, 2)] += dt
    if cur == 1:
        k = re.sub(r"\(.*", "", next(iter(active))).replace("void ", "").replace("micf::", "")[:70]
        alone[k] += dt
    if d == 1: active[n] = active.get(n, 0) + 1
    else:
        active[n] -= 1
        if active[n] == 0: del active[n]
    cur += d; last = t
tot = t1 - t0
print(f"step wall {tot/1e6:.3f} ms, kernels {len(step)}, sum of kernel time {sum(e-s for s,e,_ in step)/1e6:.3f} ms")
for k in (0, 1, 2):
    print(f"  {k}{'+' if k == 2 else ' '} kernels in flight: {busy[k]/1e6:7.3f} ms ({100*busy[k]/tot:4.1f} %)")
cnt = collections.Counter(re.sub(r"\(.*", "", n).replace("void ", "")[:60] for _, _, n in step)
print("non-micf kernels in the step:", {k: v for k, v in cnt.items() if "micf" not in k})
def first(name):
    return next(((st, en) for st, en, n in step if name in n), None)
marks_ = [("forward (until tail_col2im ends)", "tail_col2im"), ("loss fwd+bwd (until dice_bce_bwd ends)", "dice_bce_bwd"),
          ("head backward (until tail_dwout ends)", "tail_dwout"), ("blocks backward (until first wgrad_grouped starts)", "wgrad_grouped_kernel"),
          ("grouped weight gradients (until adam_tick starts)", "adam_tick"), ("adam", None)]
prev = t0
for label, name in marks_:
    if name is None:
        print(f"  phase {label}: {(t1 - prev)/1e6:.3f} ms"); break
    f = first(name)
    if f is None: continue
    edge = f[0] if ("starts" in label) else f[1]
    print(f"  phase {label}: {(edge - prev)/1e6:.3f} ms")
    prev = edge
print("kernels running ALONE (ms):")
for k, v in alone.most_common(25):
    print(f"  {v/1e6:7.3f}  {k}")
