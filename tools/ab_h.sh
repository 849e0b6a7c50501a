set -u
mkdir -p gpurun_out/abh
python -m pytest tests/test_gpu_block_fused.py -x -q 2>&1 | tail -8 > gpurun_out/abh/tests.log
for rep in 1 2; do
for v in 1 0; do
  echo "MICF_BLOCK_SAVE_H=$v bench_block:" >> gpurun_out/abh/block.log
  MICF_BLOCK_SAVE_H=$v python tools/bench_block.py --dtype bf16 --fused-only >> gpurun_out/abh/block.log 2>&1
  MICF_BLOCK_SAVE_H=$v python tools/bench_block.py --dtype bf16 --fused-only --cross >> gpurun_out/abh/block.log 2>&1
  ms=$(MICF_BLOCK_SAVE_H=$v python bench.py --no-cpu-baseline --no-roofline --steps 30 2>>gpurun_out/abh/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])")
  echo "SAVE_H=$v step: $ms" >> gpurun_out/abh/step.log
done
done
cat gpurun_out/abh/tests.log gpurun_out/abh/step.log; grep -v "^$" gpurun_out/abh/block.log | cut -c1-200
