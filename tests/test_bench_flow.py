"""bench.py at N > 1: every rank must stay in every leg (timed region, roofline leg, final barrier) so that no rank issues a
collective its peers never join (the round-1 script returned on ranks != 0 before rank 0's eager roofline steps all-reduced).
Runs the REAL bench.py control flow on CPU with 2 gloo ranks and a stub engine (--cpu-stub: same collectives, no kernels)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--cpu-stub", *extra]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)


@pytest.mark.timeout(300)
def _is_unit_roofline(roof):
    """The roofline object is keyed by a SURVEY 8(d) unit (self / cross block pair of a depth slot, one direction) whose launches
    are C-ABI entry points; the path fraction is also at the top level of the object."""
    return (roof["kernel"].split("|")[0] in ("self_fwd", "self_bwd", "cross_fwd", "cross_bwd")
            and all(l["kernel"].startswith("micf_") for l in roof["launches"]) and roof["path_frac"] == roof["path"]["frac"]
            and 0.0 < roof["frac"] < 1.0 and roof["frac_raw"] <= roof["frac"] + 1e-9)


def test_bench_two_ranks_complete_with_roofline_leg():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert "roofline" in out and "path" in out["roofline"] and "family" in out["roofline"]
    assert out["cpu_baseline"] is None                     # N > 1: the CPU leg is a rank-0, N = 1 item
    assert out["rccl_ranks"] == 2 and out["dist_backend"].startswith("gloo")   # what the collective backend itself saw


@pytest.mark.timeout(300)
def test_bench_single_process_stub_and_no_roofline():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--cpu-stub", "--no-roofline",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and "roofline" not in out and "cpu_baseline" not in out


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with NO launcher in the command (the form the driver types for N = 1) starts
    its N ranks itself: one JSON line from rank 0, rc 0."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-stub"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(env, OMP_NUM_THREADS="1"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["parallelism"] == "dp2"
    assert "roofline" in out and out["cpu_baseline"] is None


@pytest.mark.timeout(300)
def test_self_launched_bench_fails_when_a_rank_dies():
    """... and a non-zero exit code when any rank fails, with the surviving ranks stopped (not left parked in a collective)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "0", "--cpu-stub"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT,
                       env=dict(env, OMP_NUM_THREADS="1", MICF_BENCH_FAIL_RANK="1"))
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "rank 1 exited with 7" in r.stderr


def test_gpus_must_match_the_launchers_world_size():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--cpu-stub"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT, env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_self_launched_bench_two_ranks_on_one_gpu():
    """The self-launched form with the REAL engine: `python bench.py --gpus 2 --dist-backend gloo` (no torchrun), both ranks on the
    box's one MI355X: base / 128^3 / local batch 2 / bf16, capture under a live group, replays with the per-slice all-reduces."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo",
           "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["distinct_local_devices"] == 1 and out["grad_wire"] == "bf16"
    assert out["config"]["global_batch"] == 4 and out["dtype"] == "bf16" and 0.0 < out["final_loss"] < 2.0
    assert "roofline" in out and _is_unit_roofline(out["roofline"])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_nccl_branch_with_one_rank():
    """The `nccl` (= RCCL) branch of bench.py on the one device a test box has (`--force-dist`): init_process_group("nccl",
    device_id=...), the step captured under the live RCCL communicator in the data-parallel layout (split_step), every gradient
    slice all-reduced for real over the bf16 wire, the collectives of the timing / identification legs, destroy_process_group --
    everything the 8-GPU run does except the second device."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--force-dist",
           "--dist-backend", "nccl", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["dist_backend"].startswith("nccl") and out["grad_wire"] == "bf16"
    assert 0.0 < out["final_loss"] < 2.0 and out["value"] > 0 and "roofline" in out


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_real_engine_two_ranks_on_one_gpu():
    """bench.py's REAL N > 1 path before the driver's 8-GPU run -- torchrun with 2 ranks, both on the box's one
    MI355X, collectives over gloo on device tensors (`--dist-backend gloo`): process-group init, weight broadcast, the capture
    under a live group, timed replays with the post-replay weight-gradient groups + per-slice all-reduces + Adam, the eager
    roofline leg with its collectives, the MAX-over-ranks timing and rank 0's single JSON line.  BASELINE config 3's per-rank
    workload (base, 128^3, local batch 2, bf16), 3 steps."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--dist-backend", "gloo", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["distinct_local_devices"] == 1
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2" and out["dtype"] == "bf16"
    assert out["final_loss"] == out["final_loss"] and 0.0 < out["final_loss"] < 2.0
    assert out["value"] > 0 and "roofline" in out and _is_unit_roofline(out["roofline"])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_real_engine_four_ranks_on_one_gpu():
    """More than two ranks: 4 processes on the box's one MI355X (gloo on device tensors), base model on 64^3 pairs, fp32 parity mode
    (the exact fp32 gradient wire): every rank must reach every collective of the step and of the roofline leg, rank 0 prints the
    one JSON line, and the loss is the 4-rank data-parallel step's (finite, below the initial 0.75)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
           "--vol", "64", "--dtype", "fp32", "--dist-backend", "gloo", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["rccl_ranks"] == 4 and out["distinct_local_devices"] == 1 and out["grad_wire"] == "fp32"
    assert out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp4" and out["dtype"] == "fp32"
    assert 0.0 < out["final_loss"] < 0.8 and "roofline" in out


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_eight_ranks_on_one_gpu():
    """BASELINE config 3's rank count before the driver's 8-GPU run: `python bench.py --gpus 8` self-launched, all 8 ranks on the
    box's one MI355X (gloo on device tensors), base model on 64^3 pairs in the bf16 mode -- i.e. the DEFAULT gradient wire of the
    8-GPU run (bf16 on the links, fp32 in the sums: 8-way all-to-all + all-gather per slice through the HIP pack / sum / unpack
    kernels), the 8-rank bucket plan, capture under a live group, post-replay groups, MAX-over-ranks timing, `rccl_ranks: 8`."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--vol", "64",
           "--dist-backend", "gloo", "--no-cpu-baseline", "--no-roofline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["distinct_local_devices"] == 1 and out["grad_wire"] == "bf16"
    assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp8" and out["dtype"] == "bf16"
    assert 0.0 < out["final_loss"] < 0.8 and out["value"] > 0 and "roofline" not in out
