#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats CSV: per kernel calls / total ms / avg us, short names."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("micf::", "")
    n = n.replace("at::native::", "aten::")
    return n[:118]
print(f"total kernel time {tot/1e6/div:.2f} ms per step (divisor {div})")
micf = sum(float(r["TotalDurationNs"]) for r in rows if "micf::" in r["Name"])
print(f"  micf kernels {micf/1e6/div:.2f} ms, other {(tot-micf)/1e6/div:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{short(r['Name']):120s} {int(r['Calls'])/div:8.1f} {float(r['TotalDurationNs'])/1e6/div:8.3f} ms {float(r['AverageNs'])/1e3:9.1f} us")
