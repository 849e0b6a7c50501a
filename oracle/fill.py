"""Closed-form, RNG-free tensor fills shared by the golden generator and the tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Weights and inputs are never
stored in fixtures: both the reference (when the goldens are generated) and the
build (when they are checked) fill their ``state_dict`` with ``fill_state_dict``,
keyed by parameter NAME, so construction order / RNG state do not matter.
"""
import math
import zlib

import torch


def _phase(name: str) -> float:
    return (zlib.crc32(name.encode()) % 10007) / 10007.0 * 2.0 * math.pi


def lattice(shape, name: str, amp: float = 1.0, freq: float = 0.37, dtype=torch.float32):
    """amp * sin(freq*k + phase(name)) + 0.31*amp*sin(0.0173*k*k_mod + ...) over the flat index k."""
    n = 1
    for s in shape:
        n *= int(s)
    k = torch.arange(n, dtype=torch.float64)
    ph = _phase(name)
    v = torch.sin(freq * k + ph) + 0.31 * torch.sin(1.618 * k + 2.0 * ph + 0.5)
    return (amp * v).to(dtype).reshape(shape)


def fill_tensor(name: str, t: torch.Tensor) -> torch.Tensor:
    """Deterministic value for the state_dict entry `name` with the shape of `t`."""
    shape = tuple(t.shape)
    leaf = name.split(".")[-1]
    is_norm = ".norm" in name or name.endswith("norm.weight") or name.endswith("norm.bias") \
        or ".norm1." in name or ".norm2." in name
    if is_norm and leaf == "weight":
        return 1.0 + lattice(shape, name, 0.10)
    if is_norm and leaf == "bias":
        return lattice(shape, name, 0.05)
    if leaf == "bias":
        return lattice(shape, name, 0.05)
    # weights: amplitude ~ 1/sqrt(fan_in) so activations stay O(1)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= int(s)
    if "reverse_patch_embedding" in name or "up_conv" in name:
        # ConvTranspose weight is [Cin, Cout, k, k, k]: fan_in = Cin
        fan_in = int(shape[0])
    amp = 1.2 / math.sqrt(max(fan_in, 1))
    if "conv_offset.3" in name:
        amp = 2.5 / math.sqrt(max(fan_in, 1))   # offsets of O(1) voxel: exercises the sampler
    return lattice(shape, name, amp)


def stage_amplitude(name: str) -> float:
    """Round-5 amplitude schedule (fixture f10): with the plain fill the loss gradient decays ~30x per decoder level towards the
    deep stages -- every PatchExpand ends in a LayerNorm whose input is the un-normalised residual stream of the stage below
    (sigma ~ 30), so d/dx of that LayerNorm divides by 30 -- and the 8^3 / 4^3 stages' gradient norms sit 1e-4 ... 1e-6 below the
    head's (f7).  Multiplying the gain of exactly those LayerNorms by 30 keeps every stage group within 1e-2 ... 2e-1 of the
    largest gradient norm, so that per-tensor gates see the deep stages."""
    for j in range(3):
        if name == f"swin.up_layers.{j}.downsample.norm.weight":
            return 30.0
    return 1.0


@torch.no_grad()
def fill_state_dict(module: torch.nn.Module, amplitude=None) -> None:
    """amplitude: optional name -> multiplier (e.g. stage_amplitude)."""
    sd = module.state_dict()
    for name, t in sd.items():
        v = fill_tensor(name, t)
        t.copy_(v * amplitude(name) if amplitude is not None else v)


def make_volume(B, D, H, W, name="image"):
    """Two-modality input (B,2,D,H,W), O(1) values, smooth + high-frequency parts."""
    return lattice((B, 2, D, H, W), name, 0.9, freq=0.113)


def make_label_map(B, D, H, W, num_classes=8):
    """Integer class map from nested shells around the volume centre (deterministic)."""
    z = torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)
    y = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1)
    x = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W)
    b = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1)
    r = torch.sqrt(((z - D / 2 + 0.5 + b) / D) ** 2 + ((y - H / 2 + 0.5) / H) ** 2
                   + ((x - W / 2 + 0.5 - b) / W) ** 2)
    ang = torch.atan2(y - H / 2 + 0.5, x - W / 2 + 0.5) + 0 * z + 0 * b
    cls = torch.clamp(((0.5 - r) * 2.0 * (num_classes - 1)).floor() + (ang > 0).float(), 0, num_classes - 1)
    return cls.long()


def one_hot(label_map, num_classes=8):
    return torch.nn.functional.one_hot(label_map, num_classes).permute(0, 4, 1, 2, 3).float().contiguous()
