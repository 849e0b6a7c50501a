#!/bin/bash
# A/B of environment SETS on ONE box: tools/ab_envs.sh OUTDIR "A=1 B=2" "A=0" ...  ("-" = no variables) -> the benched step (30 replays), two rounds interleaved
set -u
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  ms=$(env $e python bench.py --no-cpu-baseline --no-roofline --steps 30 2>>$OUT/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])")
  echo "[$v] step: $ms" | tee -a $OUT/step.log
done
done
