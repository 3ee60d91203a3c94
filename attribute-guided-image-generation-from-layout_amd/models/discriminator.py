"""Drop-in for the reference's models/discriminator.py."""
from . import _bootstrap  # noqa: F401
from agl.discriminator import (AttributeDiscriminator, AttributeDiscriminator128, ImageDiscriminator,  # noqa: F401
                               ObjectDiscriminator, OptimizedBlock, ResidualBlock, add_sn)
from agl.losses import loss_hinge_dis, loss_hinge_gen  # noqa: F401  (models/spade/networks/loss.py:65-76; off the reference's train path)
