#!/usr/bin/env python3
"""bench.py -- training throughput of the MicFormer hot path on MI355X (BASELINE.json metric).

One "step" = the reference's training iteration (train_mmwhs_noPad.py:183-207) on a batch of synthetic 128^3 CT+MR
pairs: zero_grad -> Head forward -> MDiceLoss -> backward -> [RCCL grad all-reduce] -> Adam + cosine LR.
Workload = BASELINE.json configs[1]: MicFormer base (embed 48, depths 2-2-6-2, heads 3-6-12-24, window 2^3, 8 classes),
128^3 volumes, LOCAL batch 2 per GPU (weak scaling: configs[2] is 8 x 2 = global 16).  Inputs are generated on the device
before the timed region.  Arithmetic is fp32 end to end (the reference trains in fp32; bf16 is a later round).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields, incl. `roofline` and `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable copy rate
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_*_f32, dense (MI355X_MICROARCH.md)


def synthetic_batch(B, vol, num_classes, device, seed):
    """image ~ N(0,1) (NormalizeIntensityd contract), label = one-hot of a random integer class map (train.py:177)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn((B, 2) + vol, generator=g, device=device, dtype=torch.float32)
    # blocky label map (8^3 blocks) so classes form regions, as segmentation masks do
    coarse = torch.randint(0, num_classes, (B,) + tuple(v // 8 for v in vol), generator=g, device=device)
    lab = coarse.repeat_interleave(8, 1).repeat_interleave(8, 2).repeat_interleave(8, 3)
    tgt = torch.nn.functional.one_hot(lab, num_classes).permute(0, 4, 1, 2, 3).float().contiguous()
    return x, tgt


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written
    by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE passes with the gfx950 corrections of MI355X_MICROARCH.md);
    None when no PMC summary covers this kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    rec = table.get(kernel_key)
    if rec is None:          # same entry point, different shape tag (e.g. the number of queued layers changed): still the same kernels
        base = kernel_key.split("|")[0]
        rec = next((v for k, v in table.items() if k.split("|")[0] == base), None)
    return int(rec["hbm_bytes_per_launch"]) if rec else None


def cpu_baseline(vol, threads, budget_s=25.0):
    """The CPU restatement (oracle/, 'port') timed on the host cores on a BOUNDED sample of the same workload: full fp32
    train steps (fwd + MDiceLoss + bwd + Adam) of the base model on one (vol/2)^3 CT+MR pair -- 1/8 of the voxels of a
    128^3 pair (every stage keeps a token grid >= 2, so it is the same op mix) -- scaled by 1/8 to pairs/s of the full size."""
    import torch
    from oracle import micformer_ref as R
    from oracle.shapes import filled_params
    threads = max(1, min(threads, 64))          # torch CPU ops stop scaling (and oversubscribe SMT siblings) beyond this
    torch.set_num_threads(threads)
    sub = tuple(max(v // 2, 32) for v in vol)
    frac = (sub[0] * sub[1] * sub[2]) / float(vol[0] * vol[1] * vol[2])
    cfg = R.Cfg()
    P = filled_params(cfg)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((1, 2) + sub, generator=g)
    lab = torch.randint(0, 8, (1,) + sub, generator=g)
    tgt = torch.nn.functional.one_hot(lab, 8).permute(0, 4, 1, 2, 3).float().contiguous()
    st, times = {}, []
    t_start = time.perf_counter()
    step = 0
    while True:
        step += 1
        t0 = time.perf_counter()
        R.train_step(P, st, x, tgt, cfg, step=step)
        times.append(time.perf_counter() - t0)
        if step >= 4 or time.perf_counter() - t_start + times[-1] > budget_s:
            break
    best = min(times)
    return {"value": round(frac / best, 4), "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} full fp32 train step(s) (fwd+loss+bwd+Adam) of MicFormer base on one {sub[0]}^3 CT+MR pair "
                      f"(= {frac:.3f} of a {vol[0]}^3 pair; value scaled by that), oracle/ torch CPU ops, {threads} threads, "
                      f"best of {len(times)}: {best:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="local batch (pairs per GPU)")
    ap.add_argument("--vol", type=int, default=128)
    ap.add_argument("--embed-dim", type=int, default=48)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--serial-modalities", action="store_true", help="do not overlap the CT / MR branches on two streams")
    ap.add_argument("--detail", action="store_true", help="roofline leg: key kernels by shape too (diagnostic)")
    ap.add_argument("--eval-mode", action="store_true", help="DropPath off (default: train mode, DropPath active)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)          # backend "nccl" is RCCL on ROCm

    from micformer_amd import _lib
    from micformer_amd.engine import TrainEngine
    from micformer_amd.models.MICFormer_self import Head

    torch.manual_seed(1234)                                     # rank-identical initial weights (also broadcast by the engine)
    model = Head(embed_dim=args.embed_dim, num_classes=8).to(dev)
    model.train(not args.eval_mode)
    torch.manual_seed(1234 + rank)                              # rank-distinct DropPath stream and data
    vol = (args.vol,) * 3
    x, tgt = synthetic_batch(args.batch, vol, 8, dev, 1234 + rank)
    eng = TrainEngine(model, base_lr=1e-4, t_max=150, use_graph=not args.no_graph,
                      parallel_modalities=not args.serial_modalities)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step(x, tgt)
    if not args.no_graph and eng._graph is None:                # warmup 0: still capture outside the timed region
        eng._capture(x, tgt)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = eng.step(x, tgt)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss)
    assert loss_val == loss_val, "loss is NaN"

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        "metric": "train volumes/sec (128^3 CT+MRI pair)", "value": round(world * args.batch * args.steps / dt, 4),
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
        "config": {"workload": f"MicFormer base Head(embed_dim={args.embed_dim}, depths 2-2-6-2, heads 3-6-12-24, window 2^3, "
                               f"8 classes) full train step (fwd + MDiceLoss + bwd + Adam/cosine) on {args.vol}^3 CT+MR pairs, "
                               f"{'DropPath on' if not args.eval_mode else 'eval mode'}",
                   "local_batch": args.batch, "global_batch": world * args.batch, "parallelism": f"dp{world}",
                   "launch": "eager" if args.no_graph else "hipGraph replay"},
        "final_loss": round(loss_val, 6),
    }

    # ---- roofline of the dominant kernel: HIP events around every C-ABI launch of 2 eager steps on the launch stream(s),
    # keyed by (entry point, shape).  The dominant kernel is the (entry point, shape) with the largest total time.
    if not args.no_roofline:
        from micformer_amd import ops as _ops
        eng_graph = eng.use_graph
        eng.use_graph = False
        eng.step(x, tgt)
        torch.cuda.synchronize()
        _ops.DETAIL = True
        _lib.profile_start()
        nprof = 2
        for _ in range(nprof):
            eng.step(x, tgt)
        prof = _lib.profile_stop()
        _ops.DETAIL = False
        eng.use_graph = eng_graph
        total_ms = sum(v["ms"] for v in prof.values())
        name, top = max(prof.items(), key=lambda kv: kv[1]["ms"])
        per = top["calls"]
        sec = top["ms"] / 1e3
        gbs = top["bytes"] / sec / 1e9
        tfl = top["flops"] / sec / 1e12
        frac_hbm, frac_mfma = gbs / HBM_PEAK_GBS, tfl / MFMA_F32_PEAK_TFLOPS
        # the bound that applies is the one the kernel's arithmetic intensity puts it under
        ai = top["flops"] / max(top["bytes"], 1)
        ridge = MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        if ai > ridge:
            roof = {"bound": "mfma", "achieved": round(tfl, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(frac_mfma, 4)}
        else:
            roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(frac_hbm, 4)}
        roof.update({"traffic": pmc_traffic(name), "kernel": name, "launches_per_step": per // nprof,
                     "avg_launch_us": round(1e3 * top["ms"] / per, 2),
                     "share_of_kernel_time": round(top["ms"] / total_ms, 4),
                     "algorithmic_bytes_per_launch": top["bytes"] // per, "flops_per_launch": top["flops"] // per,
                     "hbm_frac": round(frac_hbm, 4), "mfma_f32_frac": round(frac_mfma, 4)})
        out["roofline"] = roof
        # whole-step view: entry points (all shapes merged), algorithmic bytes / flops over the sum of their event times
        merged = {}
        for k, v in prof.items():
            m = merged.setdefault(k.split("|")[0], dict(calls=0, ms=0.0, bytes=0, flops=0))
            for f in ("calls", "ms", "bytes", "flops"):
                m[f] += v[f]
        tot_b = sum(v["bytes"] for v in prof.values())
        tot_f = sum(v["flops"] for v in prof.values())
        out["kernels"] = {k: {"ms_per_step": round(v["ms"] / nprof, 3), "calls_per_step": v["calls"] // nprof,
                              "GB/s": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                              "TFLOP/s": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)}
                          for k, v in sorted(merged.items(), key=lambda kv: -kv[1]["ms"])}
        if args.detail:
            out["kernels_by_shape"] = {k: {"ms_per_step": round(v["ms"] / nprof, 3), "calls_per_step": v["calls"] // nprof,
                                           "GB/s": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                                           "TFLOP/s": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)}
                                       for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:60]}
        out["step_summary"] = {"kernel_ms_per_step": round(total_ms / nprof, 3), "launches_per_step": sum(v["calls"] for v in prof.values()) // nprof,
                               "algorithmic_GB_per_step": round(tot_b / nprof / 1e9, 3), "GFLOP_per_step": round(tot_f / nprof / 1e9, 1)}

    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(vol, os.cpu_count() or 1)
    elif not args.no_cpu_baseline:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
