#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: regenerates the evidence kept under profiles/ into gpurun_out/$R/.
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r03'   then   cp gpurun_out/r03/r03_* gpurun_out/r03/pmc_traffic.json profiles/
set -u
R=${1:-r04}
OUT=gpurun_out/$R
mkdir -p $OUT
python bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err
python bench.py --dtype fp32 --no-cpu-baseline --no-stock-loop > $OUT/${R}_bench_fp32.json 2>> $OUT/bench.err
# the step as a sequence of graphs on two streams (opt-in layout), its main chain alone and everything serial: how much of the
# parameter-gradient side work is hidden (functional.StepSegmenter; MICF_SEG_SKIP_SIDE computes WRONG updates: a timing probe only)
{
  echo "# ms per step, base / 128^3 / batch 2 / bf16, one MI355X (python bench.py --segmented --no-roofline --no-cpu-baseline --steps 30)"
  for v in "" "MICF_SEG_SERIAL=1" "MICF_SEG_SKIP_SIDE=1"; do
    ms=$(env $v python bench.py --segmented --no-cpu-baseline --no-roofline --no-stock-loop --steps 30 2>>$OUT/bench.err | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "segmented ${v:-(main chain + side batches on two streams)}: $ms"
  done
  ms=$(python bench.py --no-cpu-baseline --no-roofline --no-stock-loop --steps 30 2>>$OUT/bench.err | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "one graph (default): $ms"
} > $OUT/${R}_graph_layouts.txt
python tools/run_large.py 1 bf16 2>>$OUT/bench.err | tail -1 > $OUT/${R}_large160.jsonl
python tools/run_large.py 2 bf16 2>>$OUT/bench.err | tail -1 >> $OUT/${R}_large160.jsonl
python tools/bench_infer.py 2>>$OUT/bench.err | tail -1 > $OUT/${R}_infer512.jsonl
python tools/bench_infer.py --autocast 2>>$OUT/bench.err | tail -1 >> $OUT/${R}_infer512.jsonl
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# kernel trace + stats of the bench command (graph replay); the trace itself is large: keep the derived tables only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp -o b -- python bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-roofline --no-stock-loop \
  > $OUT/${R}_bench_under_rocprof.json 2> $OUT/rocprof.err
cp $OUT/tmp/b_kernel_stats.csv $OUT/${R}_bench_kernel_stats.csv
python tools/trace_concurrency.py $OUT/tmp/b_kernel_trace.csv > $OUT/${R}_concurrency.txt
python tools/trace_stages.py $OUT/tmp/b_kernel_trace.csv > $OUT/${R}_stages.txt
python tools/trace_stages.py $OUT/tmp/b_kernel_trace.csv --detail > $OUT/${R}_stages_detail.txt
# by (kernel, grid) from an eager run (graph replays keep the grid too, but eager separates the warm-up cleanly)
rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp -o e -- python bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-roofline --no-stock-loop --no-graph \
  > /dev/null 2>> $OUT/rocprof.err
python tools/trace_by_shape.py $OUT/tmp/e_kernel_trace.csv 4 > $OUT/${R}_kernels_by_shape.txt
# HBM-side traffic: separate PMC passes (never combined with other trace domains)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -o $c -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-roofline --no-stock-loop --no-graph \
    > /dev/null 2>> $OUT/rocprof.err
done
# PMC traffic per C-ABI key: the launches of the dominant SURVEY 8(d) unit (cross-block backward at 32^3: block_bwd + offset_head_bwd)
# and the other large keys; bench.py sums a unit's rows (profiles/pmc_traffic.json)
echo "dominant unit: $(python -c "import json;r=json.load(open('$OUT/${R}_bench.json'))['roofline'];print(r['kernel'], r['frac'], [l['kernel'] for l in r['launches']])")" > $OUT/${R}_pmc_summary.txt
python tools/pmc_summary.py $OUT/pmc $OUT/${R}_pmc_hbm_by_kernel.csv --json $OUT/pmc_traffic.json \
  --key "micf_block_bwd|2x65536x48" --kernels "block_bwd_kernel<48,block_bwd_wave48_kernel" --calls-per-step 8 \
  --key "micf_block_fwd|2x65536x48" --kernels "block_fwd_kernel<48,block_fwd_wave48_kernel" --calls-per-step 8 \
  --key "micf_block_bwd|2x1024x192" --kernels "block_bwd_kernel<192" --calls-per-step 24 \
  --key "micf_block_fwd|2x1024x192" --kernels "block_fwd_kernel<192" --calls-per-step 24 \
  --key "micf_offset_head_bwd|65536.48x65536.48" --kernels "${PMC_HEAD_BWD_KERNELS:-offset_sample_bwd4_kernel<2>@1048576;sample_gather_tile_kernel@262144;conv3_bwdx_kernel<16, 6, true>@262144}" --calls-per-step 4 \
  --key "micf_offset_head_fwd|65536.48x65536.48" --kernels "conv3_fwdx_kernel<16, true>@262144;" --calls-per-step 4 \
  --key "micf_layernorm_fwd_pair|65536.48x65536.48" --kernels "ln_fwd_v2<16, 1>@1048576;" --calls-per-step 4 >> $OUT/${R}_pmc_summary.txt
# matrix-core utilisation per kernel (its own pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o m -- \
  python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-roofline --no-stock-loop > /dev/null 2>> $OUT/rocprof.err
python tools/mfma_summary.py $OUT/mfma 30 > $OUT/${R}_mfma_util.txt
# where the waves of the block kernels spend their cycles (two SQ passes of 8 counters; quad-cycle units, see tools/pmc_sq_summary.py)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $OUT/sq -o s -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-roofline --no-stock-loop --no-graph > /dev/null 2>> $OUT/rocprof.err
python tools/pmc_sq_summary.py $OUT/sq > $OUT/${R}_sq_wave_time.txt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM \
  --kernel-trace --output-format csv -d $OUT/sq2 -o s -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-roofline --no-stock-loop --no-graph > /dev/null 2>> $OUT/rocprof.err
python tools/pmc_sq_summary.py $OUT/sq2 >> $OUT/${R}_sq_wave_time.txt
rm -rf $OUT/tmp $OUT/pmc $OUT/mfma $OUT/sq $OUT/sq2
tail -3 $OUT/${R}_pmc_summary.txt; head -12 $OUT/${R}_mfma_util.txt; head -c 400 $OUT/${R}_bench.json
