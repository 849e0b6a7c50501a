#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: regenerates the evidence kept under profiles/ into gpurun_out/r01/.
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'   then   cp gpurun_out/r01/* profiles/
set -u
OUT=gpurun_out/r01
mkdir -p $OUT
python bench.py > $OUT/r01_bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# kernel trace + stats of the bench command (graph replay); the trace itself is large: keep the derived tables only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp -o b -- python bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-roofline \
  > $OUT/r01_bench_under_rocprof.json 2> $OUT/rocprof.err
cp $OUT/tmp/b_kernel_stats.csv $OUT/r01_bench_kernel_stats.csv
python tools/trace_concurrency.py $OUT/tmp/b_kernel_trace.csv > $OUT/r01_concurrency.txt
# by (kernel, grid) from an eager run (graph replays keep the grid too, but eager separates the warm-up cleanly)
rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp -o e -- python bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-roofline --no-graph \
  > /dev/null 2>> $OUT/rocprof.err
python tools/trace_by_shape.py $OUT/tmp/e_kernel_trace.csv 4 > $OUT/r01_kernels_by_shape.txt
# HBM-side traffic: separate PMC passes (never combined with other trace domains)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -o $c -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-roofline --no-graph \
    > /dev/null 2>> $OUT/rocprof.err
done
KEY=$(python -c "import json;print(json.load(open('$OUT/r01_bench.json'))['roofline']['kernel'])")
echo "dominant kernel: $KEY" > $OUT/r01_pmc_summary.txt
python tools/pmc_summary.py $OUT/pmc $OUT/r01_pmc_hbm_by_kernel.csv --key "$KEY" --kernels "${PMC_KERNELS:-wgrad_grouped_kernel,wgrad_grouped_reduce_kernel}" --json $OUT/pmc_traffic.json >> $OUT/r01_pmc_summary.txt
rm -rf $OUT/tmp $OUT/pmc/*_kernel_trace.csv $OUT/pmc/*_counter_collection.csv $OUT/pmc/*agent_info.csv
rmdir $OUT/pmc 2>/dev/null
tail -3 $OUT/r01_pmc_summary.txt; head -c 300 $OUT/r01_bench.json
