mkdir -p gpurun_out/ab5
run() { # name lib envs...
  n=$1; L=$2; shift 2
  ms=$(env MICF_LIB=$L "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 2>>gpurun_out/ab5/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])")
  echo "$n: $ms" | tee -a gpurun_out/ab5/step.log
}
V=$PWD/ab/libmicformer_hc4tm16.so
python -m pytest tests/test_gpu_block_fused.py -x -q 2>&1 | tail -2
MICF_LIB=$V python -m pytest tests/test_gpu_block_fused.py -x -q 2>&1 | tail -2
MICF_LIB=$V MICF_BLOCK_TJ=1 MICF_BLOCK_TJ_BWD=1 python -m pytest tests/test_gpu_block_fused.py -x -q 2>&1 | tail -2
for rep in 1 2; do
run main "" A=1
run var_default $V A=1
run var_fwdTJ1 $V MICF_BLOCK_TJ=1
run var_bwdTJ1 $V MICF_BLOCK_TJ_BWD=1
run var_bothTJ1 $V MICF_BLOCK_TJ=1 MICF_BLOCK_TJ_BWD=1
run main_bothTJ1 "" MICF_BLOCK_TJ=1 MICF_BLOCK_TJ_BWD=1
done
