#!/usr/bin/env python3
"""One replayed step of a rocprofv3 kernel-trace CSV as a small CSV (start_ns relative to the step, end_ns, queue, grid,
kernel name): small enough to bring back from the GPU box and analyse offline with trace_stages.py / trace_slot.py (both accept it).
usage: trace_dump_step.py trace.csv out.csv [marker=drop_path_draw_kernel]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[3] if len(sys.argv) > 3 else "drop_path_draw_kernel"
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(ev) if marker in r["Kernel_Name"]]
if len(marks) < 3:
    sys.exit("need >= 3 steps")
lo, hi = marks[-3], marks[-1]            # two whole steps, so the consumers' "last whole step" logic still finds its markers
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Start_Timestamp", "End_Timestamp", "Queue_Id", "Grid_Size", "Kernel_Name"])
for r in ev[lo:hi + 1]:
    w.writerow([r["Start_Timestamp"], r["End_Timestamp"], r.get("Queue_Id", ""), r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Kernel_Name"]])
