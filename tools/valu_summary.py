#!/usr/bin/env python3
"""Per-kernel instruction mix from a rocprofv3 PMC pass with SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
SQ_WAVE_CYCLES GRBM_GUI_ACTIVE: VALU instructions per MFMA, VALU-busy share of the SIMD time, share of wave time spent waiting.
usage: valu_summary.py DIR [top=25]"""
import collections, csv, glob, re, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
acc = collections.defaultdict(collections.Counter); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("micf::", "")[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k] += r["Counter_Name"] == "GRBM_GUI_ACTIVE"
rows = sorted(acc.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])
print(f"{'kernel':72s} {'launches':>8s} {'VALU/MFMA':>9s} {'VALU busy':>9s} {'waiting':>8s} {'waves/SIMD':>10s}")
for k, c in rows[:top]:
    simd = (c["GRBM_GUI_ACTIVE"] or 1) / 8 * 1024
    print(f"{k:72s} {n[k]:8d} {c['SQ_INSTS_VALU'] / max(c['SQ_INSTS_MFMA'], 1):9.1f} {100 * 4 * c['SQ_ACTIVE_INST_VALU'] / simd:8.1f}% "
          f"{100 * c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):7.1f}% {4 * c['SQ_WAVE_CYCLES'] / simd:10.2f}")
