#!/bin/bash
# A/B of library variants on ONE box: tools/ab_libs.sh OUTDIR "name1 name2 ..."   ("main" = the in-tree library)
# per variant: the block microbenchmark (self + cross) and the benched step (30 replays), two rounds interleaved.
set -u
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for rep in 1 2; do
for v in $1; do
  if [ $v = main ]; then L=""; else L="$PWD/ab/libmicformer_$v.so"; fi
  echo "== $v" >> $OUT/block.log
  MICF_LIB=$L python tools/bench_block.py --dtype bf16 --fused-only 2>/dev/null >> $OUT/block.log
  MICF_LIB=$L python tools/bench_block.py --dtype bf16 --fused-only --cross 2>/dev/null >> $OUT/block.log
  ms=$(MICF_LIB=$L python bench.py --no-cpu-baseline --no-roofline --steps 30 2>>$OUT/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])")
  echo "$v step: $ms" >> $OUT/step.log
done
done
cat $OUT/step.log; grep -v "^$" $OUT/block.log | cut -c1-200
