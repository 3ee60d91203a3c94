"""Drop-in for the part of the reference's data/utils.py the train/test loops use on the hot path's outputs
(`imagenet_deprocess_batch`, data/utils.py:47-66); computed on device, returns the same CPU ByteTensor."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)
from agl.hostlogic import IMAGENET_MEAN, IMAGENET_STD, imagenet_deprocess_batch  # noqa: E402,F401

INV_IMAGENET_MEAN = [-m for m in IMAGENET_MEAN]
INV_IMAGENET_STD = [1.0 / s for s in IMAGENET_STD]
