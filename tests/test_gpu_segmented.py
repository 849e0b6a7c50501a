"""The opt-in segmented step capture (functional.StepSegmenter: the step as a sequence of HIP graphs on two streams, explicit
events; TrainEngine(segmented=True) / MICF_SEGMENTED=1) in its own process -- it needs the HIP runtime's graph packet capture
switched off before the runtime initialises (micformer_amd/_lib.py): three replayed steps of the 48-wide model must reproduce
the eager engine's loss and every parameter-gradient slice."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_segmented_capture_matches_eager(dtype):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, DT=dtype)
    env.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "segmented_worker.py"), "mid"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "segments 10 9" in out or "segments" in out, out[-2000:]
    lines = [l for l in out.splitlines() if l.startswith("mid ")]
    assert len(lines) == 3
    for l in lines:
        f = l.split()
        i = f.index("step")
        assert abs(float(f[i + 2]) - float(f[i + 3])) <= (1e-5 if dtype == "fp32" else 2e-3), l
        assert int(f[-1]) >= 1                                  # side segments exist: the step really was cut
    off = [l for l in out.splitlines() if l.startswith("gradient slices off:")]
    assert off and int(off[0].split()[3]) == 0, off


def test_step_many_as_a_sequence_of_graphs():
    """TrainEngine.step_many with segmented=True (tests/segmented_many_worker.py, own process): 2 x 4 steps of the base-width model -- every
    stage group of the carried parameter work its own graph on the side stream with an event behind it, the main chain cut where it
    waits for one -- against 8 replays of the one-step graph: losses, Adam first moments, step counter."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    env.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "segmented_many_worker.py")], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "\nOK" in out, out[-3000:]
    assert "'side_ev': 18" in out and "'wait': 18" in out, out[-1500:]        # (3 carrying steps x (5 stage groups + backward preparation))
