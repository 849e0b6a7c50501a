#!/usr/bin/env python3
"""Matrix-core utilisation per kernel from a rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVES and
GRBM_GUI_ACTIVE.   usage: mfma_summary.py DIR [top=25]

  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 4 SIMDs * 256 CUs)   (the derived metric of the same name falls back to
  this formula; MI355X_MICROARCH.md: MFMA_BUSY counts cycles, summed over all SIMDs)
Counters are summed over the launches of a kernel (name without arguments), sorted by GPU-active cycles."""
import collections, csv, glob, re, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
acc = collections.defaultdict(collections.Counter)
n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("micf::", "")[:80]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])
tot = sum(c["GRBM_GUI_ACTIVE"] for _, c in rows) or 1
print(f"{'kernel':82s} {'launches':>8s} {'share':>6s} {'MfmaUtil':>8s} {'SQ busy':>8s}")
for k, c in rows[:top]:
    act = c["GRBM_GUI_ACTIVE"] or 1
    print(f"{k:82s} {n[k]:8d} {100 * act / tot:5.1f}% {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (act * 4 * 256):7.2f}% "
          f"{100 * c['SQ_BUSY_CYCLES'] / (act * 8 * 4):7.1f}%")
