#!/bin/bash
# kernel trace of the k-steps-per-graph layout: stage table of an inner step (carry in + carry out)
set -u
OUT=gpurun_out/spg; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MICF_STEPS_PER_GRAPH=${1:-4} rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp -o s -- python bench.py --steps 8 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/bench.json 2> $OUT/rocprof.err
python tools/trace_stages.py $OUT/tmp/s_kernel_trace.csv --detail > $OUT/stages_detail.txt
python tools/trace_stages.py $OUT/tmp/s_kernel_trace.csv > $OUT/stages.txt
python tools/trace_concurrency.py $OUT/tmp/s_kernel_trace.csv > $OUT/concurrency.txt
rm -rf $OUT/tmp
cat $OUT/stages.txt; head -c 200 $OUT/bench.json
