#!/bin/bash
# kernel stats of BASELINE config 4's train step (large, 160x160x128, batch 1, bf16): top kernels per step
set -u
OUT=gpurun_out/large; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp -o l -- python tools/run_large.py 1 bf16 > $OUT/run.json 2> $OUT/rocprof.err
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/large/tmp/l_kernel_stats.csv')))
n=9.0   # 2 eager warm-ups + capture warm-ups + 7 replays ~ steps
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:45]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms total {int(r['Calls']):6d} calls avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
PY
python tools/trace_stages.py $OUT/tmp/l_kernel_trace.csv > $OUT/stages.txt 2>&1; cat $OUT/stages.txt
python tools/trace_stages.py $OUT/tmp/l_kernel_trace.csv --detail > $OUT/stages_detail.txt 2>&1
rm -rf $OUT/tmp; tail -1 $OUT/run.json
