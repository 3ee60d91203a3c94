"""Drop-in for the hinge branch of the reference's models/spade/networks/loss.py:65-76 (GANLoss, gan_mode='hinge').
The reference never calls it (train64.py uses BCE-with-logits; the file's own import at :9 is broken), so these helpers
are OFF the train path; BASELINE.json's north_star names them as part of the module surface."""
from .. import _bootstrap_up  # noqa: F401
from agl.losses import loss_hinge_dis, loss_hinge_gen  # noqa: F401
