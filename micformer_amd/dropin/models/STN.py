"""`from models.STN import SpatialTransformer` (MicFormer/models/MICFormer_self.py:10 imports it this way) -> the HIP module."""
import importlib.util as _u
import os as _os

_spec = _u.spec_from_file_location("_micf_dropin_locate", _os.path.join(_os.path.dirname(_os.path.dirname(__file__)), "_locate.py"))
_loc = _u.module_from_spec(_spec)
_spec.loader.exec_module(_loc)
_loc.package()

from micformer_amd.models.STN import Re_SpatialTransformer, SpatialTransformer  # noqa: E402,F401

__all__ = ["SpatialTransformer", "Re_SpatialTransformer"]
