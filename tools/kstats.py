"""Per-kernel launch count / average / minimum duration from a rocprofv3 --kernel-trace csv (or every *_kernel_trace.csv under a directory).
    python tools/kstats.py DIR_OR_CSV [substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit(f"no kernel trace under {src}")
pats = sys.argv[2:]
d = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if pats and not any(p in n for p in pats):
            continue
        d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:100]:100s} n={len(v):5d} avg={sum(v) / len(v):8.1f} us  min={min(v):8.1f}")
