"""SURVEY.md 8(e) on hardware: the REAL TrainEngine with world = 2 -- two processes sharing ONE MI355X, torch.distributed `gloo`
on device tensors (the 1-GPU box has no second device for RCCL; the engine code path is identical: FlatGradSync.broadcast_params on
the flat buffer, the captured forward + backward, _plan_split, in-graph flushes, per-slice async all-reduces interleaved with the
grouped weight-gradient launches, grad_scale = 1/world inside micf_adam_step).

Checked against a 1-process run of the same engine under the per-rank-loss definition (each rank's loss is over ITS batch; the
update uses the mean of the rank gradients): parameters after 2 steps, rank-identical weights after the broadcast and after the
steps, exact zeros in the never-used concat_back_dim.0 slice, rank-distinct DropPath draws in train mode, and the optional bf16
gradient all-reduce within bf16 rounding of the fp32 one."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_engine_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RUNS = {}          # mode -> results of the two ranks (a mode's run is deterministic: tests share it)


def _run_two_ranks(tmp_path, mode):
    if mode not in _RUNS:
        _RUNS[mode] = _run_two_ranks_now(tmp_path, mode)
    return _RUNS[mode]


def _run_two_ranks_now(tmp_path, mode):
    port = _free_port()
    procs, outs = [], []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = str(tmp_path / f"rank{rank}_{mode}.pt")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, WORKER, out, mode], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for p, o in zip(procs, logs):
        assert p.returncode == 0, f"worker failed:\n{o[-3000:]}"
    return [torch.load(o) for o in outs]


def _single_process_reference(steps=2):
    """The same two batches through ONE engine: per rank batch forward + backward (eager), gradients summed, Adam with
    grad_scale = 1/2 -- the per-rank-loss definition of the 2-rank step."""
    from oracle import fill
    import micformer_amd.models.MICFormer_self as M
    from micformer_amd import ops
    from micformer_amd.engine import TrainEngine
    h = M.Head(embed_dim=48, num_classes=8, depths=(1, 1, 1, 1))
    with torch.no_grad():
        for name, t in h.state_dict().items():
            t.copy_(fill.fill_tensor(name, t))
    h = h.cuda().eval()
    xs = fill.make_volume(4, 64, 64, 64)
    ts = fill.one_hot(fill.make_label_map(4, 64, 64, 64))
    eng = TrainEngine(h, base_lr=1e-4, t_max=150, use_graph=False, early_adam=False)
    losses = []
    for _ in range(steps):
        gsum, ls = None, []
        for r in range(2):
            ls.append(float(eng._fwd_bwd(xs[2 * r:2 * r + 2].cuda(), ts[2 * r:2 * r + 2].cuda())))
            gsum = eng.flat_g.clone() if gsum is None else gsum + eng.flat_g
        eng.flat_g.copy_(gsum)
        eng._adam(0.5)
        losses.append(ls)
    torch.cuda.synchronize()
    return eng, losses


def test_two_rank_engine_matches_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r0, r1 = _run_two_ranks(tmp_path, "eval")
    # rank 1 started from different weights: the broadcast of the flat buffer made them rank-identical, and they stay so
    assert torch.equal(r0["p0"], r1["p0"])
    assert torch.equal(r0["p"], r1["p"]), float((r0["p"] - r1["p"]).abs().max())
    assert torch.equal(r0["g"], r1["g"])                         # the reduced gradient SUM is the same bits on both ranks
    assert r0["wplan_n"] > 10                                    # the encoder's weight gradients were launched after the replay
    for o, n in r0["dead"]:
        assert n > 0 and float(r0["g"][o:o + n].abs().max()) == 0.0, "concat_back_dim.0 must stay an exact zero slice"
    assert r0["losses"] != r1["losses"]                          # (each rank's loss is over its own batch)
    eng, ref_losses = _single_process_reference()
    for step in range(2):
        assert abs(ref_losses[step][0] - r0["losses"][step]) <= 1e-5
        assert abs(ref_losses[step][1] - r1["losses"][step]) <= 1e-5
    ref_p = eng.flat_p.cpu()
    moved = float((ref_p - r0["p0"]).abs().max())
    assert moved > 1e-5                                          # two Adam steps at lr 1e-4 moved the weights
    # Adam normalises: an element whose gradient is at rounding level can move by up to lr per step either way; compare where
    # the gradient is well above the accumulation noise, and bound the rest by 2 steps x lr
    diff = (ref_p - r0["p"]).abs()
    assert float(diff.max()) <= 2.2e-4
    g = eng.flat_g.cpu().abs()                                   # (the summed gradient of the last step)
    solid = g > 1e-3 * float(g.max())
    assert int(solid.sum()) > 1000
    assert float(diff[solid].max()) <= 2e-6, float(diff[solid].max())


def test_two_rank_engine_train_mode_and_bf16_gradient_exchange(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r0, r1 = _run_two_ranks(tmp_path, "train")
    assert torch.equal(r0["p"], r1["p"])
    assert r0["scales"] is not None and r0["scales"].shape == r1["scales"].shape
    assert not torch.equal(r0["scales"], r1["scales"]), "DropPath draws must be rank-distinct"
    b0, b1 = _run_two_ranks(tmp_path, "bf16grad")
    e0, _ = _run_two_ranks(tmp_path, "eval")
    assert torch.equal(b0["p"], b1["p"])
    # bf16 gradient exchange: the reduced gradient is within bf16 rounding (2^-8 relative per element after the sum of two
    # rounded terms) of the fp32 exchange, relative to each parameter tensor's scale
    ge, gb = e0["g_first"], b0["g_first"]                      # (same weights on both sides: the first step's reduced gradient)
    assert float((ge - gb).abs().max()) <= 2 ** -7 * float(ge.abs().max())
    assert float((ge - gb).abs().max()) > 0


def test_two_rank_engine_default_wire_in_the_bench_configuration(tmp_path):
    """the configuration the 8-GPU bench runs -- bf16 arithmetic, `grad_bf16=None` (auto => bf16 wire) -- against the
    same arithmetic with the exact fp32 exchange: rank-identical parameters, reduced gradient within 2^-7 of the exact one per
    element (relative to the buffer's largest entry), losses of the second step within 1e-4; and the auto wire follows the
    arithmetic mode in force (checked inside the worker)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a0, a1 = _run_two_ranks(tmp_path, "bf16auto")
    e0, _ = _run_two_ranks(tmp_path, "bf16exact")
    assert a0["auto_wire"] is True and e0["auto_wire"] is False
    assert torch.equal(a0["p"], a1["p"]) and torch.equal(a0["g"], a1["g"])
    ga, ge = a0["g_first"], e0["g_first"]
    assert float((ga - ge).abs().max()) <= 2 ** -7 * float(ge.abs().max())
    assert float((ga - ge).abs().max()) > 0
    assert abs(a0["losses"][1] - e0["losses"][1]) <= 1e-4, (a0["losses"], e0["losses"])


def test_hip_wire_kernels_bit_exact_against_torch():
    """micf_grad_wire_pack / _sum / _unpack (the three kernels of the default 8-GPU gradient wire) against dist.TorchWireOps, the
    form the CPU gloo tests drive: RNE rounding to bf16 incl. ties / subnormal-range values / signed zeros, the zero-padded tail,
    the fp32 sum of 8 shards in rank order rounded ONCE, the widening -- bit for bit, at sizes with ragged tails."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micformer_amd import ops
    from micformer_amd.dist import TorchWireOps
    g = torch.Generator().manual_seed(11)
    for n, ranks in ((8, 1), (1003, 8), (4096 * 9 + 5, 8), (200_003, 4)):
        src = torch.randn(n, generator=g) * torch.logspace(-12, 3, n)
        src[: min(n, 6)] = torch.tensor([0.0, -0.0, 1.00390625, -1.01171875, 3.0e-39, 65504.0])[: min(n, 6)]   # ties, a tiny value
        shard = (-(-n // ranks) + 7) // 8 * 8
        a = torch.empty(ranks * shard, dtype=torch.bfloat16)
        TorchWireOps.pack(src, a)
        b = torch.empty(ranks * shard, dtype=torch.bfloat16, device="cuda")
        ops.HipWireOps.pack(src.cuda(), b)
        assert torch.equal(a.view(torch.int16), b.cpu().view(torch.int16)), f"pack n={n}"
        recv = (torch.randn(ranks * shard, generator=g) * 3).bfloat16()
        own_t = torch.empty(shard, dtype=torch.bfloat16)
        TorchWireOps.sum_shards(recv, ranks, own_t)
        own_h = torch.empty(shard, dtype=torch.bfloat16, device="cuda")
        ops.HipWireOps.sum_shards(recv.cuda(), ranks, own_h)
        assert torch.equal(own_t.view(torch.int16), own_h.cpu().view(torch.int16)), f"sum n={n}"
        out_t, out_h = torch.empty(n), torch.empty(n, device="cuda")
        TorchWireOps.unpack(a, out_t)
        ops.HipWireOps.unpack(b, out_h)
        assert torch.equal(out_t, out_h.cpu()), f"unpack n={n}"
