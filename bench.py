#!/usr/bin/env python3
"""bench.py -- training throughput of the MicFormer hot path on MI355X (BASELINE.json metric).

One "step" = the reference's training iteration (train_mmwhs_noPad.py:183-207) on a batch of synthetic 128^3 CT+MR
pairs: zero_grad -> Head forward -> MDiceLoss -> backward -> [RCCL grad all-reduce] -> Adam + cosine LR.
Workload = BASELINE.json configs[1]: MicFormer base (embed 48, depths 2-2-6-2, heads 3-6-12-24, window 2^3, 8 classes),
128^3 volumes, LOCAL batch 2 per GPU (weak scaling: configs[2] is 8 x 2 = global 16).  Inputs are generated on the device
before the timed region.  --dtype selects the arithmetic of the matrix-core products: fp32 (exact, the parity mode) or bf16
(bf16 MFMA operands AND bf16 storage of what the fused block kernels save for their backward / the weight gradients; fp32
accumulation / residual stream / LayerNorm / softmax / loss / master weights).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  ... bench.py --gpus 2 --dist-backend gloo    (plumbing check on a ONE-GPU box: both ranks share cuda:0, collectives over gloo on
                                                device tensors -- the engine path is the RCCL one; the JSON says so)

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields, incl. `roofline` and `cpu_baseline`).
Every rank runs every leg (timed region, roofline leg) so no rank leaves while another still has a collective to issue.
"""
import argparse
import json
import os
import sys
import time

# --segmented: the step as a sequence of HIP graphs on two streams.  It needs the HIP runtime's graph packet capture off, and
# the runtime reads that flag when it initialises (micformer_amd/_lib.py explains): set before anything imports torch.
if "--segmented" in sys.argv:
    os.environ["MICF_SEGMENTED"] = "1"
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable copy rate
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_*_f32, dense (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_*_bf16, dense

BLOCK_MARKERS = (".blocks1.", ".blocks2.", ".self_blocks1.", ".self_blocks2.")


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
    base = kernel_key.split("|")[0]
    if rec is None and base == "micf_linear_bwd_weight_grouped":
        # the tag of the grouped weight gradient is the number of queued layers: another count is still the same kernels
        # (any other entry point's tag is a shape -- a different shape is a different kernel and has no row here)
        rec = next((v for k, v in table.items() if k.split("|")[0] == base), None)
    return int(rec["hbm_bytes_per_launch"]) if rec else None


def path_bytes(embed_dim, depths, vol, batch, elem_bytes, block_param_count):
    """algorithmic bytes of the attention / transformer path per step with ideal whole-block fusion,
    26 * sum_s d_s T_s C_s * e * B_local + 2 * W_block * e   (d_s = encoder + decoder depth units of stage s, T_s tokens per
    modality per sample, C_s channels; forward 10, backward 16 activation passes per unit; block weights read fwd + bwd)."""
    sigma = 0
    for s, d in enumerate(depths):
        tok = 1
        for v in vol:
            tok *= max(-(-v // 4) >> s, 1)
        sigma += 2 * d * tok * embed_dim * (1 << s)
    return 26 * sigma * elem_bytes * batch + 2 * block_param_count * elem_bytes, sigma


def host_cpu():
    """(model name, physical cores, logical cpus) of the host from /proc/cpuinfo."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "processor":
                    logical += 1
                elif k == "model name":
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
            if phys is not None:
                cores.add((phys, core))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    return model, (len(cores) or logical), logical


def cpu_baseline(vol, timed=5, budget_s=420.0, batches=(2, 1), warmups=2):
    """the CPU restatement (oracle/, kind 'port') timed on ALL PHYSICAL host cores: the identical train
    step (fwd + MDiceLoss + bwd + Adam, fp32) of the base model on full-size CT+MR pairs.  Protocol = BASELINE.md's: `warmups`
    (2) untimed full-size steps, then `timed` (5) timed steps, median -- at B = 2, the batch the GPU value is quoted on (a step is
    ~40 s on 128 cores, so that alone is ~4.5 min).  B = 1 follows with whatever is left of `budget_s` (up to 3 timed steps, the
    pools are warm by then; it was the slower one in pairs/s on every box so far).  `value` is the better batch size in pairs/s (the
    unit the GPU value counts).  Bounded: timing stops early (and says so) once `budget_s` of CPU work is spent."""
    import statistics
    import torch
    from oracle import micformer_ref as R
    from oracle.shapes import filled_params
    model, phys, logical = host_cpu()
    threads = max(1, phys)
    torch.set_num_threads(threads)
    cfg = R.Cfg()
    g = torch.Generator().manual_seed(1234)

    def batch(b, v):
        x = torch.randn((b, 2) + v, generator=g)
        lab = torch.randint(0, 8, (b,) + v, generator=g)
        return x, torch.nn.functional.one_hot(lab, 8).permute(0, 4, 1, 2, 3).float().contiguous()

    t_start = time.perf_counter()
    per_batch, notes = {}, []
    for i, b in enumerate(batches):
        x, tgt = batch(b, vol)
        if i == 0:
            P = filled_params(cfg)
            for k in range(warmups):                                   # untimed, full size, same batch (thread pools, allocator)
                R.train_step(P, {}, x, tgt, cfg, step=k + 1)
        elif time.perf_counter() - t_start > budget_s:
            notes.append(f"B={b}: skipped (CPU budget {budget_s:.0f} s spent)")
            continue
        P, state, times = filled_params(cfg), {}, []
        for k in range(timed if i == 0 else min(timed, 3)):
            if len(times) >= (3 if i == 0 else 1) and time.perf_counter() - t_start > budget_s:
                notes.append(f"B={b}: stopped after {len(times)} timed steps (CPU budget {budget_s:.0f} s)")
                break
            t0 = time.perf_counter()
            R.train_step(P, state, x, tgt, cfg, step=k + 1)
            times.append(time.perf_counter() - t0)
        per_batch[b] = {"median_s_per_step": round(statistics.median(times), 3), "timed_steps": len(times),
                        "pairs_per_s": round(b / statistics.median(times), 4)}
    best = max(v["pairs_per_s"] for v in per_batch.values())
    return {"value": best, "unit": "pairs/s", "cores": threads, "kind": "port", "cpu_model": model,
            "physical_cores": phys, "logical_cpus": logical, "by_batch": {f"B={b}": v for b, v in per_batch.items()},
            "sample": f"full fp32 train steps (fwd+loss+bwd+Adam, lr 1e-4, cosine per iteration) of MicFormer base on {vol[0]}^3 CT+MR "
                      f"pairs, oracle/ torch CPU ops on {threads} threads (= physical cores of {model}); {warmups} untimed full-size "
                      f"warm-up steps, then median of {timed} timed steps at B={batches[0]} (BASELINE.md section 3), then up to 3 timed "
                      f"steps at the other batch size within the {budget_s:.0f} s budget; value = the better pairs/s"
                      + ("; " + "; ".join(notes) if notes else "")}


def stock_loop(embed_dim, depths, train_mode, x, tgt, steps, warmup):
    """The module-level drop-in as a user of INTEGRATION.md section 1 gets it: the reference's LITERAL loop body
    (train_mmwhs_noPad.py:183-207) with stock torch.optim.Adam (:114) and CosineAnnealingLR (:148) on the HIP modules imported through the
    reference's own import lines (one sys.path entry) -- eager, un-graphed, per-tensor autograd accumulation, ATen's multi-tensor
    Adam over the 1626 parameters, and the loop's own loss_.item() host sync.  Same workload and arithmetic mode as the headline."""
    import torch
    dropin = os.path.join(ROOT, "micformer_amd", "dropin")
    if dropin not in sys.path:
        sys.path.insert(0, dropin)
    from models.MICFormer_self import Head          # MicFormer/test.ipynb:11
    from loss import MDiceLoss                      # train_mmwhs_noPad.py:19
    torch.manual_seed(1234)
    model_1 = Head(embed_dim=embed_dim, num_classes=8, depths=depths).to(x.device)
    model_1.train(train_mode)
    criterion = MDiceLoss().to(x.device)
    optimizer = torch.optim.Adam(model_1.parameters(), lr=1e-4, weight_decay=0)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, 150)

    def iteration():
        optimizer.zero_grad()
        segs_S1 = model_1(x)
        loss_ = criterion(segs_S1, tgt)
        v = loss_.item()
        loss_.backward()
        optimizer.step()
        scheduler.step()
        return v

    for _ in range(warmup):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        v = iteration()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert v == v, "stock loop: loss is NaN"
    del optimizer, model_1
    return {"pairs_per_s": round(x.shape[0] * steps / dt, 3), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": warmup,
            "final_loss": round(v, 6),
            "loop": "optimizer.zero_grad(); segs = model(x); loss = MDiceLoss()(segs, y); loss.item(); loss.backward(); optimizer.step(); "
                    "scheduler.step() -- torch.optim.Adam(lr=1e-4) + CosineAnnealingLR, eager launches, no TrainEngine, no HIP graph"}


class _StubEngine:
    """CPU stand-in used by tests/test_bench_flow.py (--cpu-stub): issues the collectives of a data-parallel step on gloo so the
    rank control flow of this script (who is still inside which leg when a collective is issued) is exercised without a GPU."""

    def __init__(self, world):
        import torch
        self.world, self.use_graph, self._graph = world, True, None
        self.t = torch.zeros(8)

    def _capture(self, x, tgt):
        self._graph = object()

    def step(self, x, tgt):
        import torch.distributed as dist
        if self._graph is None and self.use_graph:
            self._capture(x, tgt)
        if self.world > 1:
            dist.all_reduce(self.t)
        return self.t[0] + 0.5


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(nproc, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU, the env contract of
    torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT), pass rank 0's stdout
    through (its ONE JSON line), prefix the other ranks' output onto stderr, and return non-zero if any rank fails -- the first
    failure terminates the remaining ranks (exact PIDs) instead of leaving them parked in a collective."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    base = dict(os.environ, MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, WORLD_SIZE=str(nproc),
                LOCAL_WORLD_SIZE=str(nproc), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                MICF_BENCH_SELF_LAUNCHED="1")
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or nproc) // nproc)))
    import tempfile
    procs, logs = [], []
    for r in range(nproc):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        logs.append(None if r == 0 else tempfile.TemporaryFile())      # (a file, not a pipe: nobody drains it while the rank runs)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=logs[r], stderr=None if r == 0 else subprocess.STDOUT))
    rc, alive = 0, set(range(nproc))
    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if r != 0:
                    logs[r].seek(0)
                    tail = logs[r].read().decode(errors="replace")
                    if code != 0 or os.environ.get("MICF_BENCH_VERBOSE"):
                        sys.stderr.write("".join(f"[rank {r}] {l}\n" for l in tail.splitlines()[-40:]))
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write(f"bench.py: rank {r} exited with {code}; stopping the other ranks\n")
                    for o in alive:
                        procs[o].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="local batch (pairs per GPU)")
    ap.add_argument("--vol", type=int, default=128)
    ap.add_argument("--embed-dim", type=int, default=48)
    ap.add_argument("--dtype", choices=("fp32", "bf16"), default="bf16",
                    help="matrix-core arithmetic: bf16 (BASELINE config 2; bf16 MFMA operands and bf16 storage of the fused blocks' "
                         "saves, fp32 accumulate / residual stream / master weights; passes the SURVEY 8(c) gates of "
                         "tests/test_gpu_bf16.py) or fp32 (exact, the parity mode)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default=os.environ.get("MICF_DIST_BACKEND", "nccl"),
                    help="N > 1: nccl (= RCCL over xGMI, one GPU per rank) or gloo with ALL ranks on the visible GPUs round-robin "
                         "(a plumbing check of the real engine on a one-GPU box; not a throughput number)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--segmented", action="store_true", help="capture the step as a sequence of HIP graphs replayed on two streams "
                    "with explicit events (main chain / parameter-gradient batches) instead of ONE graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--stock-loop", dest="stock_loop", action="store_true", default=None,
                    help="also time the reference's literal loop body (stock torch.optim.Adam + CosineAnnealingLR, eager) on the drop-in "
                         "modules: extra keys stock_loop_pairs_per_s / stock_loop (default: on for --gpus 1)")
    ap.add_argument("--no-stock-loop", dest="stock_loop", action="store_false")
    ap.add_argument("--stock-steps", type=int, default=5)
    ap.add_argument("--no-profile-gate", action="store_true",
                    help="roofline leg: do not hold the launch stream while a profiled eager step is enqueued (event pairs then include the "
                         "start-up latency of launches on an idle queue)")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed CPU-baseline steps at B = 2 (median is reported)")
    ap.add_argument("--cpu-budget-s", type=float, default=420.0, help="stop timing further CPU steps once this much CPU time is spent")
    ap.add_argument("--serial-modalities", action="store_true", help="do not overlap the CT / MR branches on two streams")
    ap.add_argument("--detail", action="store_true", help="roofline leg: key kernels by shape too (diagnostic)")
    ap.add_argument("--eval-mode", action="store_true", help="DropPath off (default: train mode, DropPath active)")
    ap.add_argument("--no-flush-points", action="store_true", help="launch all queued weight gradients after backward")
    ap.add_argument("--split-step", action="store_true", help="single-GPU probe of the DATA-PARALLEL step layout (graph = forward + "
                    "backward, grouped weight gradients + Adam outside it, no collective at world 1)")
    ap.add_argument("--steps-per-graph", type=int, default=int(os.environ.get("MICF_STEPS_PER_GRAPH", "1")),
                    help="single GPU: k consecutive steps per captured graph (TrainEngine.step_many: step i + 1's encoder forward "
                         "runs beside the decoder-side parameter-gradient work + Adam of step i); --steps must be a multiple of k")
    ap.add_argument("--cpu-stub", action="store_true", help="control-flow test on CPU/gloo with a stub engine (no kernels)")
    ap.add_argument("--force-dist", action="store_true", help="--gpus 1 through the N > 1 code path: a ONE-rank process group of "
                    "--dist-backend is initialised (nccl = a live RCCL communicator), the engine takes the data-parallel step layout "
                    "with its per-slice all-reduces issued for real (always_collective) and the bf16 gradient wire -- everything "
                    "the 8-GPU run does except the second device")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: start the ranks ourselves (the torch.distributed.run form keeps working: it sets WORLD_SIZE)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    stub = args.cpu_stub
    forced = args.force_dist and world == 1 and not stub
    depths = (2, 2, 6, 2)
    vol = (args.vol,) * 3
    if stub:
        dev = torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        if os.environ.get("MICF_BENCH_FAIL_RANK") == str(rank):          # test hook: a rank that dies while its peers wait
            raise SystemExit(7)
        eng, x, tgt, dtype_name, nblock = _StubEngine(world), None, None, "fp32", 0
        _lib = _ops = None
    else:
        if args.dist_backend == "gloo":
            local_rank %= max(torch.cuda.device_count(), 1)        # ranks share the visible GPU(s)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1 or forced:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if forced:
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            if args.dist_backend == "gloo":
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)      # backend "nccl" is RCCL on ROCm

        from micformer_amd import _lib
        from micformer_amd import ops as _ops
        from micformer_amd.engine import TrainEngine
        from micformer_amd.models.MICFormer_self import Head

        if args.dtype is not None:
            _ops.set_compute_dtype(args.dtype)
        dtype_name = _ops.compute_dtype()
        torch.manual_seed(1234)                                     # rank-identical initial weights (also broadcast by the engine)
        model = Head(embed_dim=args.embed_dim, num_classes=8, depths=depths).to(dev)
        model.train(not args.eval_mode)
        nblock = sum(p.numel() for n, p in model.named_parameters() if any(m in n for m in BLOCK_MARKERS))
        torch.manual_seed(1234 + rank)                              # rank-distinct DropPath stream and data
        x, tgt = synthetic_batch(args.batch, vol, 8, dev, 1234 + rank)
        eng = TrainEngine(model, base_lr=1e-4, t_max=150, use_graph=not args.no_graph,
                          parallel_modalities=not args.serial_modalities, flush_points=not args.no_flush_points,
                          segmented=args.segmented, uniform_batches=True,     # (the synthetic batch has one shape on every rank)
                          **({"split_step": True} if args.split_step else {}),
                          **({"split_step": True, "always_collective": True,
                              "grad_bf16": _ops.compute_dtype() == "bf16" and os.environ.get("MICF_GRAD_WIRE", "bf16") != "fp32"}
                             if forced else {}))

    live_group = world > 1 or forced

    def barrier():
        if live_group:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    spg = args.steps_per_graph if (world == 1 and not stub and not args.no_graph and not args.split_step) else 1
    if spg > 1 and args.steps % spg:
        raise SystemExit("--steps must be a multiple of --steps-per-graph")
    inputs_note = None
    if spg > 1:
        xs, tgts = [x] * spg, [tgt] * spg
        for _ in range(max(1, -(-args.warmup // spg))):         # (at least one call: the capture stays outside the timed region)
            eng.step_many(xs, tgts)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps // spg):
            loss = eng.step_many(xs, tgts)[-1]
        barrier()
    else:
        for _ in range(args.warmup):
            eng.step(x, tgt)
        if not args.no_graph and eng._graph is None:            # warmup 0: still capture outside the timed region
            eng._capture(x, tgt)
        if not stub and not args.no_graph and eng.input_buffers() is not None:
            # the synthetic batch lives in the buffers the captured step reads (where a loader's H2D copies would land): the timed
            # region holds no device-to-device staging of data that is already resident
            bx, bt = eng.input_buffers()
            bx.copy_(x)
            bt.copy_(tgt)
            x_t, tgt_t = bx, bt
            inputs_note = ("resident in the captured step's input buffers (TrainEngine.input_buffers(): no device-to-device "
                           "staging copy in the timed region)")
        else:
            x_t, tgt_t = x, tgt
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = eng.step(x_t, tgt_t)
        barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss)
    probe = os.environ.get("MICF_SEG_SKIP_SIDE") == "1"             # timing probe: main chain alone, parameter gradients skipped
    assert probe or loss_val == loss_val, "loss is NaN"
    ms_step = 1000.0 * dt / args.steps

    out = {
        "metric": "train volumes/sec (128^3 CT+MRI pair)", "value": round(world * args.batch * args.steps / dt, 4),
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": f"MicFormer base Head(embed_dim={args.embed_dim}, depths 2-2-6-2, heads 3-6-12-24, window 2^3, "
                               f"8 classes) full train step (fwd + MDiceLoss + bwd + Adam/cosine) on {args.vol}^3 CT+MR pairs, "
                               f"{'DropPath on' if not args.eval_mode else 'eval mode'}",
                   "local_batch": args.batch, "global_batch": world * args.batch, "parallelism": f"dp{world}",
                   "launch": "eager" if args.no_graph else (f"hipGraph replay ({spg} consecutive steps per capture"
                                                            + (f" as a sequence of {len(eng._many['graph'].segments)} graphs on two streams"
                                                               if (eng._many is not None and eng.segmented) else " in one graph") +
                                                            ": the decoder-side parameter gradients + Adam of step i run beside step i + 1's "
                                                            "encoder forward; every step is a full update on its own batch)") if spg > 1 else
                             ("hipGraph replay (one graph)" if (stub or not eng.segmented) else
                                                            f"hipGraph replay ({len((eng._graph or eng._many['graph']).segments)} segments: main chain / "
                                                            "parameter-gradient batches as separate graphs on two streams)")},
        "final_loss": round(loss_val, 6),
    }
    if inputs_note:
        out["config"]["inputs"] = inputs_note
    if live_group:
        # what the collective backend itself saw: its rank count, its name, and how many distinct devices the ranks ran on
        ids = torch.zeros(world, device=dev, dtype=torch.int64)         # (an all-reduce: the one collective every backend has)
        ids[rank] = (local_rank if not stub else 0) + 1
        dist.all_reduce(ids)
        gathered = list(ids)
        out["rccl_ranks"] = dist.get_world_size()
        out["dist_backend"] = dist.get_backend() + (" (= RCCL over xGMI)" if dist.get_backend() == "nccl" else
                                                    " (plumbing check: NOT the RCCL / xGMI path, ranks may share a GPU)")
        out["distinct_local_devices"] = len({int(g.item()) for g in gathered})
        if not stub:
            # wire format of the gradient exchange: fp32 all-reduce | bf16 on the links with fp32 sums
            # (all-to-all + all-gather) | bf16-ring (the backend's all-reduce in bf16)
            out["grad_wire"] = eng.grad_wire or "fp32"
    else:
        out["rccl_ranks"] = 1
    if probe:
        out["probe"] = "MICF_SEG_SKIP_SIDE=1: the main (data-gradient) chain alone, parameter-gradient batches skipped -- a timing probe, NOT a training step"

    # ---- roofline leg: HIP events around every C-ABI launch of 2 eager steps on the launch stream(s), keyed by (entry point,
    # shape).  EVERY rank runs it (the eager steps contain the gradient all-reduce); rank 0 reports its own numbers.
    if not args.no_roofline:
        nprof = 2
        gate_cycles = 0
        eng_graph = eng.use_graph
        eng.use_graph = False
        eng.step(x, tgt)
        if stub:
            for _ in range(nprof):
                eng.step(x, tgt)
            prof = {"stub|x": dict(calls=2, ms=1.0, bytes=1, flops=1, block_ms=1.0)}
        else:
            torch.cuda.synchronize()
            # The eager host loop (~30 us per C-ABI call) is slower than the small kernels it launches: measured on an idle queue, an
            # event pair around a 29 us launch of the 8^3 stage reads 40 us (start-up latency of a launch on an idle GPU), which both
            # misprices the launch and biases the choice of the dominant key towards many small launches.  So every profiled step is
            # enqueued behind a GATE: a spin kernel (torch.cuda._sleep) holds the launch stream for a little longer than the host needs
            # to enqueue the step, the launches then run back to back and the event pairs read kernel time (rocprofv3's durations of
            # the replayed graph agree: profiles/).  --no-profile-gate restores the un-gated measurement.
            if not args.no_profile_gate:
                t0 = time.perf_counter()
                eng.step(x, tgt)
                host_ms = (time.perf_counter() - t0) * 1e3           # enqueue time of one eager step (the queue was empty: no back-pressure)
                torch.cuda.synchronize()
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(); torch.cuda._sleep(20_000_000); eb.record()
                torch.cuda.synchronize()
                per_ms = 20_000_000 / max(ea.elapsed_time(eb), 1e-3)  # spin cycles per millisecond on this device
                gate_cycles = int(min(1.3 * host_ms + 2.0, 60.0) * per_ms)
            _ops.DETAIL = True
            _lib.profile_start()
            for _ in range(nprof):
                if gate_cycles:
                    torch.cuda._sleep(gate_cycles)
                eng.step(x, tgt)
                torch.cuda.synchronize()
            prof = _lib.profile_stop()
            _ops.DETAIL = False
            if gate_cycles:
                # ... and what an event pair with NOTHING between its events reads behind the same gate (the cost of the second event
                # itself, ~2-5 us) is taken off every pair: without it a key of 24 launches of 29 us still read 34 us per launch
                torch.cuda._sleep(int(2.0 * per_ms))
                pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
                for a_, b_ in pairs:
                    a_.record(); b_.record()
                torch.cuda.synchronize()
                empty = sorted(a_.elapsed_time(b_) for a_, b_ in pairs)
                pair_ms = empty[len(empty) // 2]
                for v in prof.values():
                    v["ms_raw"] = v["ms"]                            # (the uncorrected sum stays in the record: frac_raw / *_us_raw)
                    cut = min(pair_ms * v["calls"], 0.5 * v["ms"])
                    if v.get("block_ms"):
                        v["block_ms"] = max(v["block_ms"] - cut * v["block_ms"] / max(v["ms"], 1e-9), 0.0)
                    v["ms"] -= cut
        eng.use_graph = eng_graph
        barrier()
        e = 2 if dtype_name == "bf16" else 4
        mfma_peak = MFMA_BF16_PEAK_TFLOPS if dtype_name == "bf16" else MFMA_F32_PEAK_TFLOPS
        total_ms = sum(v["ms"] for v in prof.values())
        # ---- SURVEY.md 8(d) UNITS.  The algorithmic bytes of 8(d) are defined per block of a depth slot: a self block moves its
        # activation 2 times forward / 3 times backward, a cross block 3 / 5 times (T*C*e per pass per modality) plus the block's
        # weights once.  A unit here = the pair of blocks of one kind of one depth slot in one direction (both modalities), i.e.
        # EVERYTHING the data path launches for it: the fused block launch and, for a cross pair, LayerNorm-1, the offset conv, the
        # sampler and their adjoints (entry points that have no 8(d) bytes of their own -- they are part of the cross block).
        # frac = unit's 8(d) bytes / the summed event time of all its launches / 8 TB/s.  The dominant unit is the one with the
        # largest total time per step (a stable rule: a unit's total does not depend on how its work is cut into launches).
        units = {}
        for k, v in prof.items():
            for uname, (ucalls, ums, u8d) in v.get("by_unit", {}).items():
                # (the event-pair correction applied to the key is applied to its share in the unit in the same proportion)
                scale = v["ms"] / max(v.get("ms_raw", v["ms"]), 1e-9)
                u = units.setdefault(uname, {"ms": 0.0, "ms_raw": 0.0, "s8d": 0, "own": 0, "flops": 0, "launches": {}, "n": 0})
                u["ms"] += ums * scale
                u["ms_raw"] += ums
                u["s8d"] += u8d
                u["own"] += v["bytes"] * ucalls // max(v["calls"], 1)
                u["flops"] += v["flops"] * ucalls // max(v["calls"], 1)
                u["launches"][k] = (ucalls, ums * scale)
                if u8d:
                    u["n"] += ucalls                         # occurrences of the unit = its block launches
        if not units:                                        # (stub engine / a model without fused pairs: fall back to the largest key)
            k, v = max(prof.items(), key=lambda kv: kv[1]["ms"])
            units = {k: {"ms": v["ms"], "ms_raw": v.get("ms_raw", v["ms"]), "s8d": v.get("s8d_bytes", 0) or v["bytes"], "own": v["bytes"],
                         "flops": v["flops"], "launches": {k: (v["calls"], v["ms"])}, "n": v["calls"]}}

        def unit_row(uname, u):
            n = max(u["n"], 1)
            sec, sec_raw = u["ms"] / 1e3, u["ms_raw"] / 1e3
            tr, complete = 0, True
            for k, (c, _) in u["launches"].items():
                t = pmc_traffic(k)
                if t is None:
                    complete = False
                else:
                    tr += t * c
            return {"unit": uname, "units_per_step": n // nprof, "avg_unit_us": round(1e3 * u["ms"] / n, 2),
                    "avg_unit_us_raw": round(1e3 * u["ms_raw"] / n, 2), "ms_per_step": round(u["ms"] / nprof, 3),
                    "algorithmic_bytes_per_unit": u["s8d"] // n,
                    "frac": round(u["s8d"] / max(sec, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_raw": round(u["s8d"] / max(sec_raw, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
                    "hbm_util": round(u["own"] / max(sec, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
                    "mfma_frac": round(u["flops"] / max(sec, 1e-12) / 1e12 / mfma_peak, 4),
                    "traffic": (tr // n) if (complete and tr) else None,
                    "launches": [{"kernel": k, "per_unit": round(c / n, 2), "avg_launch_us": round(1e3 * m / max(c, 1), 2)}
                                 for k, (c, m) in sorted(u["launches"].items(), key=lambda kv: -kv[1][1])]}

        name, top = max(units.items(), key=lambda kv: kv[1]["ms"])
        row = unit_row(name, top)
        ridge = mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        ai = top["flops"] / max(top["s8d"], 1)
        roof = {"bound": "hbm", "achieved": round(row["frac"] * HBM_PEAK_GBS, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": row["frac"],
                "traffic": row["traffic"], "kernel": name,
                "rule": "SURVEY 8(d) unit = every data-path launch of one depth slot's self / cross block pair in one direction; "
                        "bytes = (self 2 fwd / 3 bwd, cross 3 fwd / 5 bwd) passes of T*C*e per modality + the block weights once, "
                        "e = %d B; time = sum of the unit's launch durations (HIP events on the launch stream); dominant = the unit "
                        "with the largest total time per step; hbm_util = the launches' OWN bytes over the same time" % e,
                "frac_raw": row["frac_raw"], "units_per_step": row["units_per_step"], "avg_unit_us": row["avg_unit_us"],
                "avg_unit_us_raw": row["avg_unit_us_raw"], "ms_per_step": row["ms_per_step"],
                "share_of_kernel_time": round(top["ms"] / total_ms, 4),
                "algorithmic_bytes_per_unit": row["algorithmic_bytes_per_unit"], "hbm_util": row["hbm_util"],
                "mfma_frac": row["mfma_frac"], "arithmetic_intensity_vs_ridge": [round(ai, 1), round(ridge, 1)],
                "traffic_over_algorithmic": round(row["traffic"] / max(row["algorithmic_bytes_per_unit"], 1), 2) if row["traffic"] else None,
                "launches": row["launches"]}
        if gate_cycles:
            roof["event_pair_us_subtracted"] = round(1e3 * pair_ms, 2)
        roof["units"] = [unit_row(k, u) for k, u in sorted(units.items(), key=lambda kv: -kv[1]["ms"])[:8]]
        # whole-step view: entry points (all shapes merged), algorithmic bytes / flops over the sum of their event times
        merged = {}
        for k, v in prof.items():
            m = merged.setdefault(k.split("|")[0], dict(calls=0, ms=0.0, bytes=0, flops=0, s8d_bytes=0))
            for f in ("calls", "ms", "bytes", "flops", "s8d_bytes"):
                m[f] += v.get(f, 0)
        fam_name, fam = max(merged.items(), key=lambda kv: kv[1]["ms"])
        roof["family"] = {"kernel": fam_name, "ms_per_step": round(fam["ms"] / nprof, 3), "calls_per_step": fam["calls"] // nprof,
                          "GB/s": round(fam["bytes"] / max(fam["ms"], 1e-9) / 1e6, 1),
                          "TFLOP/s": round(fam["flops"] / max(fam["ms"], 1e-9) / 1e9, 2),
                          "hbm_util": round(fam["bytes"] / max(fam["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                          "frac_8d": round(fam.get("s8d_bytes", 0) / max(fam["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                          "share_of_kernel_time": round(fam["ms"] / total_ms, 4)}
        # SURVEY.md 8(d) "attention path's HBM roofline": ideal-fusion bytes of the transformer blocks over the time the step
        # spends in them = replayed step wall minus the event time of everything that is NOT a block kernel (patch embed /
        # merging / expand, concat linears, final norms, head, loss, Adam); the serial sum of the block launches is given too.
        pbytes, sigma = path_bytes(args.embed_dim, depths, vol, args.batch, e, nblock)
        block_sum_ms = sum(v.get("block_ms", 0.0) for v in prof.values()) / nprof
        other_ms = total_ms / nprof - block_sum_ms
        t_block_ms = max(ms_step - other_ms, 1e-6)
        roof["path"] = {"bound": "hbm", "algorithmic_bytes_per_step": int(pbytes), "sum_dTC": int(sigma), "elem_bytes": e,
                        "t_block_kernels_ms": round(t_block_ms, 3), "block_launch_sum_ms": round(block_sum_ms, 3),
                        "non_block_ms": round(other_ms, 3),
                        "achieved": round(pbytes / (t_block_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(pbytes / (t_block_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        roof["path_frac"] = roof["path"]["frac"]          # (north_star's "attention path" number, also at the top level of the object)
        # the largest C-ABI keys (single entry points; a key that is only PART of an 8(d) unit has no 8(d) bytes of its own: its
        # frac_hbm_algorithmic is null and hbm_util is its own traffic)
        top_keys = []
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:4]:
            a_b = v.get("s8d_bytes", 0)
            top_keys.append({"kernel": k, "launches_per_step": v["calls"] // nprof, "avg_launch_us": round(1e3 * v["ms"] / max(v["calls"], 1), 2),
                             "avg_launch_us_raw": round(1e3 * v.get("ms_raw", v["ms"]) / max(v["calls"], 1), 2),
                             "ms_per_step": round(v["ms"] / nprof, 3),
                             "frac_hbm_algorithmic": round(a_b / max(v["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4) if a_b else None,
                             "hbm_util": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                             "mfma_frac": round(v["flops"] / max(v["ms"], 1e-9) / 1e9 / mfma_peak, 4),
                             "traffic": pmc_traffic(k)})
        roof["top_keys"] = top_keys
        out["roofline"] = roof
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
                               "algorithmic_GB_per_step": round(tot_b / nprof / 1e9, 3), "GFLOP_per_step": round(tot_f / nprof / 1e9, 1),
                               "whole_step_TFLOP/s": round(tot_f / nprof / (ms_step * 1e-3) / 1e12, 2),
                               "whole_step_GB/s": round(tot_b / nprof / (ms_step * 1e-3) / 1e9, 1)}

    if (args.stock_loop if args.stock_loop is not None else (world == 1 and not forced)) and not stub and not live_group:
        # (one GPU only: the stock loop has no gradient exchange; the engine's flat buffers stay alive beside it -- ~6 GB + ~6 GB)
        sl = stock_loop(args.embed_dim, depths, not args.eval_mode, x, tgt, args.stock_steps, 2)
        out["stock_loop_pairs_per_s"] = sl["pairs_per_s"]
        out["stock_loop"] = sl
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and not stub:
            out["cpu_baseline"] = cpu_baseline(vol, timed=args.cpu_steps, budget_s=args.cpu_budget_s)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if live_group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
