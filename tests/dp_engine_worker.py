"""Worker of tests/test_gpu_dist_engine.py: ONE rank of a 2-process data-parallel TrainEngine run on ONE MI355X
(torch.distributed `gloo` on device tensors; RANK / WORLD_SIZE / MASTER_* from the environment).  Writes what the test compares
to the .pt file named in argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(out_path, mode):
    from oracle import fill
    import micformer_amd.models.MICFormer_self as M
    from micformer_amd.engine import TrainEngine
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # (the wire-format modes use the narrow model: a quarter of the gradient bytes through gloo's host copies -- the exchange logic
    #  under test does not depend on the kernels' widths)
    h = M.Head(embed_dim=24 if mode in ("bf16auto", "bf16exact") else 48, num_classes=8, depths=(1, 1, 1, 1))
    with torch.no_grad():
        for name, t in h.state_dict().items():
            t.copy_(fill.fill_tensor(name, t))
        if rank == 1:                                            # rank-distinct weights BEFORE the engine: broadcast_params must fix them
            for p in h.parameters():
                p.mul_(1.5)
    h = h.cuda()
    h.train(mode == "train")
    torch.manual_seed(1234 + rank)                               # rank-distinct DropPath stream (bench.py does the same)
    xs = fill.make_volume(2 * world, 64, 64, 64)
    ts = fill.one_hot(fill.make_label_map(2 * world, 64, 64, 64))
    x, t = xs[2 * rank:2 * rank + 2].cuda(), ts[2 * rank:2 * rank + 2].cuda()
    auto_wire = None
    if mode in ("bf16auto", "bf16exact"):
        # the bench's configuration: bf16 arithmetic; wire format left to the engine (auto = bf16) or forced to the exact fp32 one
        from micformer_amd import ops
        ops.set_compute_dtype("bf16")
        eng = TrainEngine(h, base_lr=1e-4, t_max=150, use_graph=True, grad_bf16=None if mode == "bf16auto" else False)
        auto_wire = bool(eng.grad_bf16)
        if mode == "bf16auto":
            ops.set_compute_dtype("fp32")
            assert not eng.grad_bf16, "auto wire must follow the arithmetic mode in force (fp32 parity mode = exact exchange)"
            ops.set_compute_dtype("bf16")
            assert eng.grad_bf16
    else:
        eng = TrainEngine(h, base_lr=1e-4, t_max=150, use_graph=True, grad_bf16=(mode == "bf16grad"))
    assert eng.world == world and eng.split_step
    p_after_bcast = eng.flat_p.clone()
    losses = [float(eng.step(x, t))]
    torch.cuda.synchronize()
    g_first = eng.flat_g.cpu()
    losses.append(float(eng.step(x, t)))
    torch.cuda.synchronize()
    scales = None
    if mode == "train":
        h.swin._predraw_drop_path(2, x.device)                   # one more draw from this rank's device-side DropPath stream
        scales = torch.stack([torch.stack(b.__dict__.pop("_pending_scales")) for b in h.modules() if "_pending_scales" in b.__dict__]).cpu()
    names = [n for n, _ in h.named_parameters()]
    dead = [i for i, n in enumerate(names) if ".concat_back_dim.0." in n]
    torch.save({"rank": rank, "losses": losses, "p0": p_after_bcast.cpu(), "p": eng.flat_p.cpu(), "g": eng.flat_g.cpu(), "g_first": g_first,
                "dead": [(eng.offsets[i], eng.sizes[i]) for i in dead], "scales": scales, "auto_wire": auto_wire,
                "wplan_n": eng._wplan.n if eng._wplan is not None else -1}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "eval")
