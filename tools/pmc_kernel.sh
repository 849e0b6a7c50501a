#!/bin/bash
# Run ON THE GPU BOX: SQ / TCC counters of the kernels matching $1 while running "$2..." ; prints per-kernel averages.
# usage: bash tools/pmc_kernel.sh <kernel-substring> <outdir> -- <command...>
set -u
PAT=$1; OUT=$2; shift 3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o c -- "$@" > /dev/null 2>> $OUT/err.log
done
python - "$PAT" $OUT <<'PY'
import csv, glob, sys, collections
pat, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat in k:
            acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} avg {sum(v)/len(v):14.1f}  n={len(v)}")
PY
