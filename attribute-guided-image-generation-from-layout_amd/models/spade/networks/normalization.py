"""Drop-in for the SPADE layer of the reference's models/spade/networks/normalization.py:66-108."""
from .. import _bootstrap_up  # noqa: F401
from agl.generator import SPADE  # noqa: F401
