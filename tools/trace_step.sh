#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the replayed bench step -> per-stage tables.  tools/trace_step.sh OUTNAME [ENV=VAL ...]
set -u
N=$1; shift
OUT=gpurun_out/$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp -o b -- python bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-roofline --no-stock-loop > $OUT/bench.json 2> $OUT/rocprof.err
python tools/trace_stages.py $OUT/tmp/b_kernel_trace.csv > $OUT/stages.txt
python tools/trace_stages.py $OUT/tmp/b_kernel_trace.csv --detail > $OUT/stages_detail.txt
cp $OUT/tmp/b_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/tmp
