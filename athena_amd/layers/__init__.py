"""Layers of the hybrid attention/CTC model (mirror of the reference's athena/layers)."""
from .commons import ACTIVATIONS, BatchNormParams, Dense, LayerNormParams, PositionalEncoding
from .attention import AttnMask, MultiHeadAttention
from .transformer import (Transformer, TransformerDecoder, TransformerDecoderLayer, TransformerEncoder,
                          TransformerEncoderLayer)
