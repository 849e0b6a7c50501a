cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/tr -o $c -- python tools/bench_block.py --dtype bf16 --stage 0 --fused-only > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/tr/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "block_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][11:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024 * 2 / 1e6
    w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024 / 1e6
    print(k, "fetch(2x) MB", round(f, 1), "write MB", round(w, 1))
PY
rm -rf gpurun_out/tr
