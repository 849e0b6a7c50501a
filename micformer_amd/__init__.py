"""micformer_amd -- MI355X-native (gfx950) MicFormer training hot path behind the reference's nn.Module API.

Importing this package loads libmicformer_hip.so (hand-written HIP kernels, plain C-ABI).  There is no CPU or
PyTorch-eager fallback: a missing library raises ImportError, CPU tensors raise at call time.
"""
from . import _lib  # noqa: F401  (fail loudly if the HIP library is missing)
from .models.MICFormer_self import Head, MicFormer  # noqa: F401
from .loss.dice import MDiceLoss, MDiceLoss_Val  # noqa: F401

__version__ = "0.1.0"
