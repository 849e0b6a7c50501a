#!/usr/bin/env python3
"""Per-kernel SQ wave-time split from a rocprofv3 --pmc pass (counter_collection CSV): where the waves' cycles go.
usage: pmc_sq_summary.py <dir> [name substring ...]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
subs = sys.argv[2:] or ["block_bwd_wave48", "block_fwd_wave48", "block_bwd_kernel<48", "block_fwd_kernel<48", "block_bwd_kernel<192", "block_fwd_kernel<192"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    for s in subs:
        if s in n:
            agg[s][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                cnt[s] += 1
for s in subs:
    a = agg[s]
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{s}: dispatches {cnt[s]}")
    for k in sorted(a):
        print(f"    {k:28s} {a[k]:16.0f}  {100 * a[k] / wc:6.1f} % of wave cycles")
