"""Worker of tests/test_gpu_segmented.py (own process: the segmented capture needs the runtime flag set before HIP starts): small / mid / base configurations, segmented vs eager."""
import os, sys, math
os.environ["MICF_SEGMENTED"] = "1"          # (before micformer_amd / the HIP runtime: _lib.py turns the graph packet capture off)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import fill
from micformer_amd import ops
from micformer_amd.engine import TrainEngine
import micformer_amd.models.MICFormer_self as M

def head(E, depths):
    h = M.Head(embed_dim=E, num_classes=8, depths=depths)
    with torch.no_grad():
        for n, t in h.state_dict().items():
            t.copy_(fill.fill_tensor(n, t))
    return h.cuda().eval()

which = sys.argv[1]
E, depths, n, B = {"small": (24, (1, 1, 1, 1), 64, 2), "mid": (48, (1, 1, 1, 1), 64, 2), "base": (48, (2, 2, 6, 2), 128, 2)}[which]
kw = eval(sys.argv[2]) if len(sys.argv) > 2 else {}
ops.set_compute_dtype(os.environ.get("DT", "fp32"))
x = fill.make_volume(B, n, n, n).cuda(); t = fill.one_hot(fill.make_label_map(B, n, n, n)).cuda()
ref = TrainEngine(head(E, depths), base_lr=1e-9, t_max=9, use_graph=False)
eng = TrainEngine(head(E, depths), base_lr=1e-9, t_max=9, use_graph=True, segmented=True, **kw)
rs = torch.cuda.Stream() if os.environ.get("SEG_STREAM", "0") == "1" else torch.cuda.current_stream()
for i in range(int(os.environ.get('NSTEPS', '3'))):
    l0 = float(ref.step(x, t)) if os.environ.get("NOREF", "0") != "1" else 0.0
    torch.cuda.synchronize()
    with torch.cuda.stream(rs):
        l1 = eng.step(x, t)
    torch.cuda.synchronize()
    l1 = float(l1)
    torch.cuda.synchronize()
    print(which, kw, "step", i, l0, l1, "segments", [k for k, _ in eng._graph.segments].count("main"), [k for k, _ in eng._graph.segments].count("side"), flush=True)
print("finite: g", bool(torch.isfinite(eng.flat_g).all()), "p", bool(torch.isfinite(eng.flat_p).all()), "m", bool(torch.isfinite(eng.flat_m).all()),
      "ref g", bool(torch.isfinite(ref.flat_g).all()), flush=True)
g0, g1 = ref.flat_g, eng.flat_g
sc = float(g0.abs().max())
bad = 0
names = [n for n, _ in ref.model.named_parameters()]
for nm, o, m in zip(names, ref.offsets, ref.sizes):
    a, b = g0[o:o + m], g1[o:o + m]
    # (bf16: tensors whose whole gradient is rounding-level small are run-to-run noise -- the gates of test_step_layouts_agree)
    rel_tol, floor = (0.06, 3e-4) if os.environ.get("DT", "fp32") == "bf16" else (0.02, 1e-6)
    if float((a - b).abs().max()) > rel_tol * float(a.abs().max()) + floor * sc:
        bad += 1
        print("   off:", nm, float(a.abs().max()), float(b.abs().max()), float((a - b).abs().max()))
print("gradient slices off:", bad, "of", len(ref.sizes), flush=True)
