#!/usr/bin/env python3
"""Group a rocprofv3 kernel-trace CSV by (kernel, grid): calls, total ms, avg us.  usage: trace_by_shape.py trace.csv steps [filter]"""
import csv, re, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
flt = sys.argv[3].split(",") if len(sys.argv) > 3 else None
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("micf::", "")
    if flt and not any(f in n for f in flt):
        continue
    key = (n[:80], r["Grid_Size_X"], r["Grid_Size_Y"])
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[0]:82s} grid=({k[1]},{k[2]}) calls={v[0]/steps:6.1f} tot={v[1]/steps/1e3:7.3f}ms avg={v[1]/v[0]:7.1f}us")
