"""Makes `micformer_amd` importable from the drop-in shim packages: the shim directory is the ONE sys.path entry a caller adds
(INTEGRATION.md section 1); the package itself lives two levels up and need not be installed."""
import importlib.util
import os
import sys


def package():
    if importlib.util.find_spec("micformer_amd") is None:
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        sys.path.append(root)
    import micformer_amd
    return micformer_amd
