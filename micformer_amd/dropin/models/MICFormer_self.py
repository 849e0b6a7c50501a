"""`from models.MICFormer_self import Head` (MicFormer/test.ipynb:11; train_mmwhs_noPad.py:26) -> micformer_amd's HIP modules.
Every public name of the reference file is re-exported; the classes ARE micformer_amd.models.MICFormer_self's (same objects)."""
import importlib.util as _u
import os as _os

_spec = _u.spec_from_file_location("_micf_dropin_locate", _os.path.join(_os.path.dirname(_os.path.dirname(__file__)), "_locate.py"))
_loc = _u.module_from_spec(_spec)
_spec.loader.exec_module(_loc)
_loc.package()

from micformer_amd.models import MICFormer_self as _impl  # noqa: E402
from micformer_amd.models.MICFormer_self import *  # noqa: E402,F401,F403

__all__ = list(_impl.__all__)
