#!/usr/bin/env python3
"""step_many as a SEQUENCE of graphs (TrainEngine(segmented=True)): losses and Adam first moments of 2 x 4 steps of the base-width
model at 64^3 (bf16, train mode) against 8 replays of the one-step graph.  Own process: the runtime flag below must be set before
torch initialises the HIP runtime (micformer_amd/_lib.py)."""
import os, sys
os.environ["MICF_SEGMENTED"] = "1"
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import fill
from micformer_amd import ops
from micformer_amd.engine import TrainEngine
import micformer_amd.models.MICFormer_self as MM
ops.set_compute_dtype(os.environ.get("DT", "bf16"))
x = fill.make_volume(2, 64, 64, 64).cuda(); t = fill.one_hot(fill.make_label_map(2, 64, 64, 64)).cuda()
def engine(seg):
    torch.manual_seed(11)
    h = MM.Head(embed_dim=48, num_classes=8).cuda().train()
    torch.manual_seed(12)
    return TrainEngine(h, base_lr=1e-4, t_max=50, use_graph=True, segmented=seg)
one = engine(False)
l1 = [float(one.step(x, t)) for _ in range(8)]
many = engine(True)
assert many.segmented, "segmented capture not available in this process"
lm = [float(l) for _ in range(2) for l in many.step_many([x] * 4, [t] * 4)]
torch.cuda.synchronize()
print("one ", l1); print("many", lm)
print("segments", len(many._many["graph"].segments), {k: sum(1 for s in many._many["graph"].segments if s[0] == k) for k in ("main", "side", "side_ev", "wait", "join")})
ok = all(abs(a - b) <= 2e-3 for a, b in zip(l1, lm))
fin = torch.isfinite(one.flat_m) & torch.isfinite(many.flat_m)
dm = float((one.flat_m - many.flat_m)[fin].abs().max()) / float(one.flat_m[fin].abs().max())
print("rel dm", dm, "steps", int(many.adam_state[0].item()))
print("OK" if ok and dm <= 3e-2 and int(many.adam_state[0].item()) == 8 else "MISMATCH")
sys.exit(0 if ok and dm <= 3e-2 else 1)
