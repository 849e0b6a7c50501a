#!/usr/bin/env python3
"""Summarise the HBM-side traffic counters of two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -o FETCH_SIZE -- python bench.py --steps 2 --warmup 0 --no-graph ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR -o WRITE_SIZE -- (same command)
  tools/pmc_summary.py DIR profiles/r01_pmc_hbm_by_kernel.csv [--key 'entry|shape' --kernels k1,k2 --json profiles/pmc_traffic.json]

Units / corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE is
TCC_EA0_RDREQ x 64 B, i.e. HALF the bytes of wide coalesced reads -> doubled here; WRITE_SIZE is taken as reported
(uncalibrated).  Infinity-Cache hits are counted, so "traffic" is an upper bound of true HBM bytes.  Per-step values divide by
the number of adam_tick_kernel launches (one per step) seen in the pass.
DIR may also be a by-kernel csv this script wrote earlier (the raw counter files stay on the GPU box): then only the --key / --json
part runs.  --key / --kernels / --calls-per-step may be repeated (one table row per key); a kernel list is separated by ';' (or ',' when no name in it holds one); a kernel matches by SUBSTRING of
"name@grid size" (rows are per kernel AND grid: `offset_sample_bwd4_kernel<2>@1048576` is the 32^3 launch of that kernel)."""
import argparse, collections, csv, json, os, re

ap = argparse.ArgumentParser()
ap.add_argument("dir"); ap.add_argument("out_csv")
ap.add_argument("--key", action="append"); ap.add_argument("--kernels", action="append"); ap.add_argument("--json")
ap.add_argument("--calls-per-step", type=float, action="append", help="C-ABI calls of the --key entry point per step (bytes are divided by it)")
args = ap.parse_args()


def load(counter):
    tot, calls, steps = collections.Counter(), collections.Counter(), 0
    with open(os.path.join(args.dir, f"{counter}_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("micf::", "")
            n = re.sub(r"^at::native::", "aten::", n)[:90]
            gs = r.get("Grid_Size")                                    # total work-items of the dispatch: separates the stages a kernel runs at
            if not gs and r.get("Grid_Size_X"):
                gs = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
            if gs:
                n += "@" + str(gs)
            tot[n] += float(r["Counter_Value"]); calls[n] += 1
            steps += n.startswith("adam_tick_kernel")
    return tot, calls, max(steps, 1)


if os.path.isfile(args.dir):
    rows = []
    for r in csv.DictReader(open(args.dir)):
        vals = list(r.values())
        rows.append((float(vals[4]), vals[0], float(vals[1]), float(vals[2]), float(vals[3])))
else:
    fetch, calls, steps = load("FETCH_SIZE")
    write, _, steps_w = load("WRITE_SIZE")
    rows = []
    for k in set(fetch) | set(write):
        fb = 2.0 * fetch[k] * 1024 / steps          # gfx950 correction: x2
        wb = write[k] * 1024 / steps_w
        rows.append((fb + wb, k, calls[k] / steps, fb, wb))
    rows.sort(reverse=True)
    with open(args.out_csv, "w") as f:
        f.write("kernel,launches_per_step,fetch_bytes_per_step(2x FETCH_SIZE KiB),write_bytes_per_step,hbm_side_bytes_per_step\n")
        for t, k, c, fb, wb in rows:
            f.write(f"\"{k}\",{c:.1f},{fb:.0f},{wb:.0f},{t:.0f}\n")
    print(f"steps: {steps}; total HBM-side bytes per step: {sum(r[0] for r in rows)/1e9:.2f} GB")
    for t, k, c, fb, wb in rows[:12]:
        print(f"{t/1e6:9.1f} MB/step  fetch {fb/1e6:8.1f}  write {wb/1e6:8.1f}  x{c:6.1f}  {k}")
if args.key and args.json:
    table = {}
    if os.path.exists(args.json):
        table = json.load(open(args.json))
    for key, kernels, n in zip(args.key, args.kernels, args.calls_per_step or [1.0] * len(args.key)):
        ks = [k for k in (kernels.split(";") if ";" in kernels else kernels.split(",")) if k.strip()]      # ("name<a, b>;": ONE name holding a comma)
        sel = [r for r in rows if any(k in r[1] for k in ks)]
        table[key] = {"kernels": ks, "launches_per_call": round(sum(r[2] for r in sel) / n, 2),
                      "fetch_bytes_per_call": round(sum(r[3] for r in sel) / n), "write_bytes_per_call": round(sum(r[4] for r in sel) / n),
                      "hbm_bytes_per_launch": round(sum(r[0] for r in sel) / n),
                      "note": f"per C-ABI call ({n:g} per step); 2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes"}
        print("wrote", args.json, key, table[key])
    json.dump(table, open(args.json, "w"), indent=1)
