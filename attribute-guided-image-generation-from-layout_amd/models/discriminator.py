"""Drop-in for the reference's models/discriminator.py."""
from . import _bootstrap  # noqa: F401
from agl.discriminator import (AttributeDiscriminator, AttributeDiscriminator128, ImageDiscriminator,  # noqa: F401
                               ObjectDiscriminator, OptimizedBlock, ResidualBlock, add_sn)
