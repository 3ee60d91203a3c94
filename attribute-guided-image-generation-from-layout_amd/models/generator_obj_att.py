"""Drop-in for the reference's models/generator_obj_att.py (64x64 generator): same public names,
constructor / forward signatures and state_dict keys; the arithmetic runs on libagl.so (MI355X)."""
from . import _bootstrap  # noqa: F401
from agl.generator import (AttributeEncoder, ConditionalBatchNorm2d, ConvLSTMCell, CropEncoder, Decoder,  # noqa: F401
                           GlobalEncoder, LayoutConvLSTM, LayoutEncoder, ResidualBlock, get_z_random)
from agl.generator import Generator as _Generator
from .bilinear import crop_bbox_batch  # noqa: F401
from .spade.networks.normalization import SPADE  # noqa: F401


class Generator(_Generator):
    def __init__(self, num_embeddings, obj_att_dim=64, z_dim=8, obj_size=64, clstm_layers=3, attribute_dim=128):
        super().__init__(num_embeddings, obj_att_dim, z_dim, obj_size, clstm_layers, attribute_dim, res128=False)


Generator64 = Generator
