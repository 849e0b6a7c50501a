#!/bin/bash
# kernels of the LAST replayed step of BASELINE config 4 (large, 160x160x128, batch 1, bf16) by (kernel, grid): run on the GPU box through gpurun
# (the table in profiles/r06_large160_kernels.txt)
OUT=gpurun_out/large; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp -o l -- python tools/run_large.py 1 bf16 > $OUT/run.json 2> $OUT/err.log
python - <<'PY'
import csv, re, collections
rows = list(csv.DictReader(open("gpurun_out/large/tmp/l_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last replayed step: take the last 1/7th by adam_tick markers
ticks = [i for i, r in enumerate(rows) if "adam_tick" in r["Kernel_Name"]]
lo, hi = ticks[-2] + 1, ticks[-1] + 1
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
print(f"last step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("micf::", "")
    agg[(n[:70], r["Grid_Size_X"], r["Grid_Size_Y"])][0] += 1
    agg[(n[:70], r["Grid_Size_X"], r["Grid_Size_Y"])][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"sum of kernel time {tot / 1e3:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k[0]:72s} grid=({k[1]},{k[2]}) calls={v[0]:4d} tot={v[1] / 1e3:7.3f}ms avg={v[1] / v[0]:7.1f}us")
PY
rm -rf $OUT/tmp
