"""Drop-in for the reference's utils/model_saver_iter.py (same import path, same behaviour)."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)
from agl.checkpoint import load_model, save_model  # noqa: E402,F401
