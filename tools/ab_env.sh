#!/bin/bash
# A/B of an environment switch on ONE box: tools/ab_env.sh OUTDIR VAR "v1 v2 ..."  -> the benched step (30 replays), two rounds interleaved
set -u
OUT=gpurun_out/$1; VAR=$2; shift 2
mkdir -p $OUT
for rep in 1 2; do
for v in $1; do
  ms=$(env $VAR=$v python bench.py --no-cpu-baseline --no-roofline --steps 30 2>>$OUT/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])")
  echo "$VAR=$v step: $ms" | tee -a $OUT/step.log
done
done
