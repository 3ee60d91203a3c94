"""Makes `agl` importable when only this directory's parent is on sys.path (drop-in use:
put attribute-guided-image-generation-from-layout_amd/ on PYTHONPATH in place of the reference root)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
