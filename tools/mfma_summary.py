#!/usr/bin/env python3
"""Matrix-core utilisation per kernel from a rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA, SQ_WAVE_CYCLES,
SQ_WAVES and GRBM_GUI_ACTIVE.   usage: mfma_summary.py DIR [top=25]

  MfmaUtil   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
  waves/SIMD = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
Calibration on this box (block_fwd_kernel<48,..,bf16>, 132.4 us): GRBM_GUI_ACTIVE = 2.51 M = 8 XCDs x 2.37 GHz x 132 us (the
counter is summed over the XCDs); SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA for v_mfma_f32_16x16x32_bf16 (cycles, summed over
the 1024 SIMDs); SQ_WAVE_CYCLES counts quad-cycles (MI355X_MICROARCH.md).  Counters are summed over the launches of a kernel
(name without arguments), sorted by GPU-active cycles."""
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
rows = sorted(((k, c) for k, c in acc.items() if not k.startswith("__amd_rocclr")),       # (runtime copy / fill helpers: eager warm-up steps)
              key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])
tot = sum(c["GRBM_GUI_ACTIVE"] for _, c in rows) or 1
print(f"{'kernel':82s} {'launches':>8s} {'share':>6s} {'MfmaUtil':>8s} {'waves/SIMD':>10s} {'MFMA instr/launch':>18s}")
for k, c in rows[:top]:
    simd_cycles = (c["GRBM_GUI_ACTIVE"] or 1) / 8 * 1024
    print(f"{k:82s} {n[k]:8d} {100 * c['GRBM_GUI_ACTIVE'] / tot:5.1f}% {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:7.2f}% "
          f"{4 * c['SQ_WAVE_CYCLES'] / simd_cycles:10.2f} {c['SQ_INSTS_MFMA'] / max(n[k], 1):18.0f}")
